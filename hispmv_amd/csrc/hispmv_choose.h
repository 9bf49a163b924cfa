// hispmv_choose.h -- per-matrix FORMAT AND TILING CHOICE: slice stream or transposed tile stream, column tiles, band tiles.
// The MI355X analogue of the reference's per-matrix configuration search (automation_tool/src/dse.py:23-95 picks the
// hardware configuration per matrix from its row statistics; tileAndPad spmv-helper.cpp:242-263 decides the tiling).
// Host-only code (no HIP): the loader (hispmv_abi.cpp) calls choose_format() for every sparse handle, the host-only entry
// hispmv_prep_choose_format() exposes the same decision to tests on a CPU-only box, tools/sanitize_host.sh runs it under
// ASan/UBSan.  Until round 3 this logic lived inside hispmv_abi.cpp (compiled by hipcc, reachable only with a device).
#pragma once
#include <cstdint>
#include <functional>
#include <utility>
#include <vector>

#include "hispmv_plan.h"
#include "hispmv_prep.h"
#include "hispmv_tts.h"

namespace hispmv {

// The host side of one part of a sparse matrix: a slice stream with its launch plan and device layout, or a tile stream.
struct HostPart {
    SliceStream st;
    std::vector<FixEntry> fix_short, fix_long;   // rows cut by slice boundaries: chains of <= / > kFixShortMax slices
    LaunchPlan plan;
    DeviceStream dstream;                        // the planned stream in its device layout (compact / wide groups)
    bool is_tts = false;
    TtsStream tts;
    // BATCH LAYOUT (round 4): a second plan and device layout of the same slices with longer groups (four times, else three times,
    // else twice as long: batch_group_div), for batch calls that share the chip between many matrices.  Since the step kernel an item of
    // the queue pays 3 - 4 us of start-up (descriptor chain, first slice from HBM, window staging) whatever its length: resident groups
    // below 80 slices (TSOPF, Si41Ge41H72, crankseg_2, nd6k, thread) take it: the set 0.2749 -> 0.2617 ms, the pessimistic family 0.2772 ->
    // 0.2678 (profiles/r4_experiments/step_kernel/group_len.json).  First built for the two-lane grids:  A resident plan gives a matrix one workgroup per CU; when its groups come out short
    // (< kBatchGroupBelow slices: nd6k 27, thread 17) a workgroup spends a large part of its life staging its x window, which nothing
    // hides with one 1024-thread workgroup per CU.  Half as many workgroups of twice the length take 1.2 % off the step of the
    // benchmark set -- and 20 - 36 % longer when such a matrix runs alone on half the chip, which is why the single launches keep
    // the first plan (profiles/r4_experiments/group_len.json).  Same slices, same carries, same arithmetic per slice: same bits.
    bool has_batch_layout = false;
    LaunchPlan batch_plan;
    DeviceStream batch_dstream;
    WordVec batch_words;                         // (device_layout only: the planned host words of the batch layout)
};
constexpr int kBatchGroupBelow = 80;

struct FormatOptions {
    int format_mode = 2;          // HISPMV_FORMAT: 0 slices always, 1 tile stream whenever the plan has no window, 2 auto
    int tts_geometry = 0;         // HISPMV_TTS_GEOMETRY: 0 standard, 1 tall, 2 auto, 3 paired, 4 zerofill
    bool band_tiles = true;       // HISPMV_BAND_TILES
    bool stray_split = true;      // HISPMV_STRAY_SPLIT: the few elements outside the groups' x windows become a second part (tile_kind 3)
    int64_t col_tile_bytes = 4 << 20;   // HISPMV_COL_TILE_BYTES: x bytes per L2-sized column tile (0 = no tiling)
    int64_t tts_min_nnz = 1 << 20;      // HISPMV_TTS_MIN_NNZ
    bool tts_small = false;             // HISPMV_TTS_SMALL (experiment)
    double tts_max_lines = 48.0;        // HISPMV_TTS_MAX_LINES: a tile stream is taken when a gather of 64 elements touches at most this many lines of x
                                        //   (32 until round 4: ASIC_680k, 34 lines, costs the step of the set 4 us less as 84 tiles than as 645 L2-gather
                                        //   workgroups although it is no faster alone -- the step is the sum of its kernels' CU-time; r4_small_class.sh)
    int tall_rows = 0, tall_slots = 0, tall_tiles = 0, tall_zero_fill = -1, tall_parts = 0;   // HISPMV_TTS_TALL_SHAPE=rows,slots,tiles per part,zero fill[,parts] (experiments; 0 / -1 = the geometry's own)
    bool no_stream_skip = false;  // HISPMV_NO_STREAM_SKIP: always build the whole-matrix slice stream before deciding (round 3's order)
    bool device_layout = false;   // HISPMV_LAYOUT=device: the slice streams keep their 8-byte host words and the loader lays them out on the
                                  //   device (layout_on_device, byte-identical).  Off by default: the upload of the larger words from pageable
                                  //   memory costs more than the host packer saves (set of 20: prep + upload 1.38 s against 1.22 s)
    bool batch_layout = true;     // HISPMV_BATCH_LAYOUT=0: no second (long-group) layout for batch calls
    int batch_group_div = 4;      // HISPMV_BATCH_GROUP_DIV: the batch layout is planned for n_cus / this many workgroups (then / 3, / 2 when that plan does not keep its kind)
    bool batch_plan_search = false;   // HISPMV_BATCH_PLAN_SEARCH=1: the batch layout's plan from all six planner configurations (as until the end of round 4) instead of the resident one alone
    int64_t batch_min_slices = 512;   // HISPMV_BATCH_MIN_SLICES: matrices with fewer slices keep their first layout (many short items: the filler at the end of the step kernel's queue)
    int batch_group_below = kBatchGroupBelow;     // HISPMV_BATCH_GROUP_BELOW (experiments): resident groups shorter than this get the batch layout
    bool decide_only = false;     // skip the device layouts the decision does not need (tests: the choice, not the bytes)
    static FormatOptions from_env();
};

// The queue of the step kernel (hispmv_kernels.h: launch_spmv_step), host-only: items of two classes -- slice items (a 1024-thread group
// or four 256-thread groups) and tiles -- with a cost each in microseconds of a CU, `n_wg` workgroups drawing them in queue order.
//   mode 0 (default): the LONG tiles (cost > a quarter of the step, total cost / n_wg) alternate with the longest slice items, so that
//          cache-bound tiles and HBM-bound groups start side by side and every long tile has started early; behind them longest first;
//   mode 1: longest first (both classes merged by cost);  mode 2: all tiles, then all slice items, each in the order given;
//   modes 3, 4, 16*t + s (experiments): two / three slice items per long tile; cycles of t long tiles and s slice items -- all measured
//   worse than one to one (profiles/r4_experiments/step_kernel/summary.json: head_of_the_queue).
// -> for every queue position {class (0 slice item, 1 tile), index into that class's list}.
std::vector<std::pair<int, int>> order_step_queue(const std::vector<double>& slice_costs, const std::vector<double>& tile_costs, int n_wg, int mode);

struct FormatChoice {
    int format = 0;               // 0 slice stream(s), 1 transposed tile stream(s)
    int tile_kind = 0;            // parts.size() > 1: 1 column ranges, 2 ranges of the offset from the scaled diagonal (band tiles),
                                  //   3 stray split (part 0: the elements inside their group's x window, part 1: the others)
    int col_tile_width = 0, col_tile_base = 0;
    bool l2_tiles = false;        // the column tiles gather x through L2: pinned to XCD subsets in a batch call
    double tts_lines_per_gather = 0;
    std::vector<HostPart> parts;
};

// For every CSR entry: 1 when the x window of ITS workgroup in `plan` (the launch plan of the matrix's whole slice stream) holds
// the entry's 64-byte block of x, 0 when the entry gathers through L2 -- the criterion of the stray split.
std::vector<uint8_t> window_membership(const Csr& csr, const LaunchPlan& plan);

// Decides the device format of `csr` for a device with n_cus compute units and builds its parts.  `prebuilt` (may be NULL):
// the whole-matrix slice stream when the device preprocessor has already made it.  `lap` (may be empty) is called with the
// name of every finished phase (HISPMV_PREP_TRACE).  csr is consumed.
FormatChoice choose_format(Csr&& csr, SliceStream* prebuilt, int n_cus, const FormatOptions& opt,
                           const std::function<void(const char*)>& lap = {});

}  // namespace hispmv
