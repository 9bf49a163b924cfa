// hispmv_ctx.h -- the context and matrix objects behind the C ABI (include/hispmv.h), shared by the translation units of the
// library's host side: hispmv_abi.cpp (context, handles, upload, single launches), hispmv_batch.cpp (hispmv_spmv_device_batch:
// launch tables, lanes, HIP-graph replay) and hispmv_prep_abi.cpp (the host-only hispmv_prep_* entries).  MI355X counterpart of
// the reference's FpgaHandle (pyhispmv/include/fpga_handle.h:9-74), which owns the XRT device, the per-channel matrix arena and
// the kernel run object.  Internal header: nothing here is part of the ABI.
#pragma once
#include "../../include/hispmv.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "hispmv_choose.h"
#include "hispmv_kernels.h"
#include "hispmv_plan.h"
#include "hispmv_prep.h"
#include "hispmv_prep_device.h"
#include "hispmv_tts.h"

#define HISPMV_API extern "C" __attribute__((visibility("default")))


namespace hispmv {

struct Matrix {
    bool dense = false;
    int32_t rows = 0, cols = 0;
    int64_t nnz = 0;
    double prep_seconds = 0;
    int64_t device_bytes = 0;
    bool loaded = false;
    // A sparse matrix is one slice stream, or -- when x is too large for an XCD's L2 and the columns
    // are scattered -- one stream per COLUMN TILE (the reference's column tiling, tileAndPad
    // spmv-helper.cpp:242-263, with L2 in the role of the BRAM x window): part 0 computes
    // y = alpha*A0*x + beta*bias, part t > 0 accumulates y = alpha*At*x + 1*y.
    struct Part : HostPart {                       // host side (hispmv_choose.h; released after upload) + device side
        TtsDeviceMatrix tdev;                      //   of a tile stream
        hispmv::SpmvDeviceMatrix dev;                      //   of a slice stream
        hispmv::SpmvDeviceMatrix batch_dev;                //   ... in its batch layout (HostPart::has_batch_layout): same slices, headers' rows,
        bool has_batch_dev = false;                        //       carries and fix lists; its own groups, fragments, slice bytes and spill flags
    };
    std::vector<Part> parts;
    std::vector<float> dense_host;
    int64_t n_slices = 0, n_elems = 0, n_split = 0, compact_slices = 0;
    int plan_threads = 0, plan_group = 0, plan_lds = 0, col_tile_width = 0, col_tile_base = 0;
    int tile_kind = 0;          // parts.size() > 1: 1 column ranges, 2 ranges of the offset from the (scaled) diagonal (band tiles)
    int format = 0;             // 0 slice stream, 1 transposed tile stream
    int index = -1;             // position in the context's handle list
    double tts_lines_per_gather = 0;
    bool l2_tiles = false;      // the column tiles gather x through L2 (L2-sized tiles): pinned to XCD subsets in a batch call
    float* d_dense = nullptr;
    // column tiles t > 0 write alpha*A_t*x here (tile t, vector v of a batched pass: d_ypart + ((t-1)*kMaxBatch + v)*rows);
    // a merge pass adds them to y after the cut rows of every tile are fixed up
    float* d_ypart = nullptr;
    // column parts: for every part the fix-list index of each row (or -1), parts x rows, so that the merge of the partial
    // vectors can apply the fix-ups of its rows itself (spmv_tail_multi_kernel); nullptr when a part has a long chain
    int32_t* d_fix_of_row = nullptr;
    std::vector<void*> allocs;
};

}  // namespace hispmv

struct hispmv_ctx {
    int device = 0;
    int num_ch_A = 0, num_ch_B = 0, num_ch_C = 0, urams_per_pe = 0, fp_acc_latency = 0;
    bool dense_overlay = false, pre_accumulator = false, row_dist_net = false;
    hipStream_t stream = nullptr;
    hipStream_t user_stream = nullptr;   // last caller-supplied stream a launch went to (hispmv_synchronize waits for it too)
    // hispmv_spmv_device_batch: the independent main launches of a call (tile streams, slice classes) go to the caller's
    // stream and to these side streams, forked from / joined to it with events, so that the tail of one grid overlaps the
    // head of the next (HISPMV_BATCH_STREAMS=1 keeps everything on one stream)
    hipStream_t side[2] = {nullptr, nullptr};
    bool cu_split = false;       // HISPMV_CU_SPLIT (experiment): the side streams carry CU masks; slice / dense grids -> side[0], tile streams -> side[1]
    hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
    int batch_streams = 2;
    int batch_order = 0;         // HISPMV_BATCH_ORDER: 0 tile streams first (default), 1 small slice grids first
    bool batch_lanes_heavy_first = true;   // HISPMV_BATCH_LANES=rr: plain round-robin lanes
    // The step kernel (hispmv_kernels.h: launch_spmv_step): a call that shares the chip between its matrices runs all its slice groups
    // and tiles as items of ONE queue drawn by one persistent workgroup per CU.  HISPMV_STEP_KERNEL=0: the grids of rounds 1 - 4 on two lanes.
    bool step_kernel = true;
    int step_order = 0;          // HISPMV_STEP_ORDER: 0 = kinds mixed in proportion, long items first (default); 1 = longest first (lpt); 2 = grid order
    bool batch_graphs = false;   // HISPMV_BATCH_GRAPH=1: two-stream batch calls captured into a HIP graph and replayed (the default until round 4;
                                 //   plain launches measure 1 - 1.5 % faster on the benchmark set: profiles/r4_experiments/graph_vs_plain.json)
    // ... for calls that stream at least this much: forking to and joining from a side stream costs ~13 us (measured on
    // the three model_test layers: 49.9 us on one stream, 63.1 on two; the 20-matrix set: 353 -> 344 us with two)
    int64_t batch_streams_min_bytes = 256ll << 20;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ms = -1.0f;
    std::mutex mu;
    std::string err;
    std::vector<std::unique_ptr<hispmv::Matrix>> mats;
    int selected = -1;
    int64_t arena_budget = 0, arena_used = 0;
    float *d_x = nullptr, *d_y = nullptr;     // device vectors of run_kernel / linear: [x | bias] and y
    int64_t cap_x = 0, cap_y = 0;
    int* h_err = nullptr;        // pinned, device-mapped word set by a kernel whose bounded carry wait expired
    int* d_err = nullptr;        //   (its device address): read on the host after a stream sync, no copy
    // run_kernel / linear with host vectors: x and bias are gathered in one pinned block and go up in ONE copy, y comes
    // back through pinned memory too (pageable hipMemcpyAsync stages and synchronises per call: 3 copies + the error word
    // cost ~65 us around a 20 us kernel)
    float* h_stage = nullptr;
    float* d_stage = nullptr;            // the device address of the same pinned block (run_kernel / linear write y straight into it)
    int64_t cap_stage = 0;
    // hispmv_spmv_device_batch: the launches of one call signature (handles, vectors, beta == 0 or not) with their device
    // tables, built on the first call and replayed afterwards
    struct BatchLaunch {
        int kind = 0;                                   // 0 slice kernels of one workgroup size, 1 fix-up of cut rows, 2 merge of column-tile
                                                        // partial vectors, 3 transposed tile streams, 4 dense overlay (GeMV), 5 tail, 6 step kernel
        std::vector<hispmv::TtsEntry> tts;                      // kind 3
        std::vector<hispmv::GemvEntry> gemv;                    // kind 4: the dense overlay handles of the call in one grid
        std::vector<const hispmv::SpmvDeviceMatrix*> parts;     // kinds 0, 1
        std::vector<float*> ys;                         // kind 1: where each part's cut rows live (y or a partial vector)
        std::vector<int32_t> rows;                      // kind 2
        std::vector<uint8_t> item_tiles;                // kind 0: parts per item (> 1: the XCD-pinned column tiles of one matrix)
        std::vector<int32_t> fix_counts;                // kind 5 (fix-up + merge in one launch): short fix entries per part; rows = merged matrices
        std::vector<hispmv::MultiEntry> multi;                  // kind 0: host copy of the table (the step kernel's table is the concatenation)
        void* d_table = nullptr;
        void* d_table2 = nullptr;                       // kind 5: the TailMergeEntry table; kind 6: the TtsEntry table
        // kind 6 (the step kernel, hispmv_kernels.h: launch_spmv_step): every slice group and every tile of the call as items of one queue
        void* d_items = nullptr;                        //   n_items x {kind | entry << 8, index}, in queue order
        unsigned* d_sync = nullptr;                     //   ticket + exit counter (zeroed once; the kernel rearms them)
        unsigned n_items = 0;
        int step_workgroups = 0;
        size_t step_lds = 0;
        bool step_strays = false;
        int lane = 0;                                   // main launches: 0 = the caller's stream, k > 0 = side stream k - 1
        bool in_lane = false;                           // kind 5 (HISPMV_LANE_TAILS): the tail of ONE lane's matrices, enqueued on that lane's stream before the join
        int64_t weight = 0;                             // main launches: device bytes of the matrices in the grid
    };
    struct BatchPlan {
        std::vector<uint64_t> key;
        std::vector<BatchLaunch> launches;
        int64_t stream_bytes = 0;     // 8 B per entry of the call's sparse matrices: decides whether side streams pay
        int lanes = 1;                // streams the main launches are spread over
        // The launches of a two-stream call captured once into a HIP graph and replayed.  TWO executables of the same captured
        // graph, each with the alpha it was last patched to and an event recorded behind its last launch: a call with another
        // alpha patches the executable that is NOT in flight (hipGraphExecKernelNodeSetParams rewrites the executable's kernel
        // arguments in place -- patching one whose earlier launch is still queued could run that launch with the new alpha),
        // and waits for that executable's own last launch -- two calls back -- before it touches it.
        struct GraphSlot {
            hipGraphExec_t exec = nullptr;
            float alpha = 0.0f;
            hipEvent_t done = nullptr;       // recorded on the launch stream behind the last launch of `exec`
            bool launched = false;
            uint64_t last_use = 0;
        } slot[2];
        uint64_t use_counter = 0;
        hipGraph_t graph_src = nullptr;      // the captured graph the executables were instantiated from (kept: its node handles patch alpha)
        int runs = 0;
    };
    std::vector<BatchPlan> batch_plans;
    int64_t graph_instantiations = 0, graph_alpha_updates = 0;     // hispmv_batch_graph_stats
    int64_t last_batch[4] = {-1, 0, 0, 0};                          // hispmv_batch_call_info
    // Rows shared between slices: "fixup" = second tiny launch, "lookback" = single launch with carry
    // granules, "auto" (default) = look-back without ticket when the whole grid is co-resident (small
    // matrices, where the extra launch costs as much as the kernel), fix-up otherwise.
    int carry_mode = 2;          // 0 fixup, 1 lookback, 2 auto (HISPMV_CARRY)
    // COO -> CSR -> slice stream: 0 on the host (OpenMP), 1 on the device (hispmv_prep_device.hip), 2 auto = device from
    // 2 M entries (HISPMV_PREP=host|device|auto); both give the same stream byte for byte
    int prep_mode = 2;
    hispmv::DevicePrepTimes last_prep_times;
    // device format of matrices whose plan gathers x through L2: 0 slice stream always, 1 transposed tile stream whenever
    // the plan has no window, 2 auto = transposed tile stream when its gathers touch <= 32 cache lines of x per wave
    // instruction (HISPMV_FORMAT=slices|tts|auto)
    hispmv::FormatOptions format_opts;   // HISPMV_FORMAT / _TTS_GEOMETRY / _BAND_TILES / _COL_TILE_BYTES / _TTS_MIN_NNZ (hispmv_choose.h)
    // geometry of a transposed tile stream: 0 the 8 K-row tiles always, 1 the tall geometry (two column parts of 16 K-row
    // tiles) for every tile stream, 2 auto (HISPMV_TTS_GEOMETRY=standard|tall|auto)
    int n_cus = 256;
};

struct hispmv_prep {
    hispmv::Csr csr;
    hispmv::SliceStream st;
    hispmv::LaunchPlan plan;
    hispmv::TtsStream tts;
    hispmv::DeviceStream dstream;     // hispmv_prep_device_stream: the planned stream in its device layout
};


namespace hispmv {

// error reporting (hispmv_abi.cpp): the message lands in the context, or -- without one -- in a thread-local that
// hispmv_last_error(NULL) reads; prep_error() is the one hispmv_prep_last_error() reads
int fail(hispmv_ctx* c, int code, const std::string& msg);
int hip_fail(hispmv_ctx* c, hipError_t e, const char* what);
std::string& prep_error();
#define HIP_TRY(c, call)                                                 \
    do {                                                                 \
        hipError_t e_ = (call);                                          \
        if (e_ != hipSuccess) return ::hispmv::hip_fail((c), e_, #call); \
    } while (0)

// Every device / pinned allocation of the library is released through these: a failing free (a pointer freed twice, a
// pointer the runtime does not know) is counted, and hispmv_free_failures() lets a test read the count.
extern std::atomic<int64_t> g_free_failures;
template <class T> void dev_free(T*& p) {
    if (p && hipFree((void*)p) != hipSuccess) { g_free_failures++; (void)hipGetLastError(); }
    p = nullptr;
}
template <class T> void host_free(T*& p) {
    if (p && hipHostFree((void*)p) != hipSuccess) { g_free_failures++; (void)hipGetLastError(); }
    p = nullptr;
}

int check_device_error(hispmv_ctx* c);      // hispmv_abi.cpp
void free_batch_plans(hispmv_ctx* c);       // hispmv_batch.cpp
int spmv_batch_locked(hispmv_ctx* c, int32_t n, const int32_t* idx, const float* const* d_x, const float* const* d_bias,
                      float* const* d_y, float alpha, float beta, hipStream_t s);

}  // namespace hispmv
