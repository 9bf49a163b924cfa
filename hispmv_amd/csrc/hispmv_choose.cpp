// hispmv_choose.cpp -- see hispmv_choose.h.  Host-only.
#include "hispmv_choose.h"

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace hispmv {

FormatOptions FormatOptions::from_env() {
    FormatOptions o;
    if (const char* e = std::getenv("HISPMV_FORMAT")) o.format_mode = !std::strcmp(e, "slices") ? 0 : !std::strcmp(e, "tts") ? 1 : 2;
    if (const char* e = std::getenv("HISPMV_BAND_TILES")) o.band_tiles = std::atoi(e) != 0;
    if (const char* e = std::getenv("HISPMV_STRAY_SPLIT")) o.stray_split = std::atoi(e) != 0;
    if (const char* e = std::getenv("HISPMV_TTS_GEOMETRY"))
        o.tts_geometry = !std::strcmp(e, "standard") ? 0 : !std::strcmp(e, "tall") ? 1 : !std::strcmp(e, "paired") ? 3 : !std::strcmp(e, "zerofill") ? 4 : !std::strcmp(e, "tallgap") ? 5 : 2;
    if (const char* e = std::getenv("HISPMV_COL_TILE_BYTES")) o.col_tile_bytes = std::atoll(e);
    if (const char* e = std::getenv("HISPMV_TTS_MIN_NNZ")) o.tts_min_nnz = std::atoll(e);
    o.tts_small = std::getenv("HISPMV_TTS_SMALL") != nullptr;
    o.no_stream_skip = std::getenv("HISPMV_NO_STREAM_SKIP") != nullptr;
    if (const char* e = std::getenv("HISPMV_BATCH_LAYOUT")) o.batch_layout = std::atoi(e) != 0;
    if (const char* e = std::getenv("HISPMV_BATCH_PLAN_SEARCH")) o.batch_plan_search = std::atoi(e) != 0;
    if (const char* e = std::getenv("HISPMV_BATCH_GROUP_DIV")) o.batch_group_div = std::max(2, std::atoi(e));
    if (const char* e = std::getenv("HISPMV_BATCH_MIN_SLICES")) o.batch_min_slices = std::max<int64_t>(1, std::atoll(e));
    if (const char* e = std::getenv("HISPMV_BATCH_GROUP_BELOW")) o.batch_group_below = std::max(1, std::atoi(e));
    if (const char* e = std::getenv("HISPMV_LAYOUT")) o.device_layout = !std::strcmp(e, "device");
    if (const char* e = std::getenv("HISPMV_TTS_MAX_LINES")) o.tts_max_lines = std::atof(e);
    if (const char* e = std::getenv("HISPMV_TTS_TALL_SHAPE")) std::sscanf(e, "%d,%d,%d,%d,%d", &o.tall_rows, &o.tall_slots, &o.tall_tiles, &o.tall_zero_fill, &o.tall_parts);
    return o;
}

namespace {

// Column range [c0, c1) of a CSR matrix as its own CSR (global column ids are kept: x is shared).
Csr column_tile(const Csr& m, int32_t c0, int32_t c1) {
    Csr t;
    t.rows = m.rows; t.cols = m.cols;
    t.row_ptr.assign((size_t)m.rows + 1, 0);
#pragma omp parallel for num_threads(host_threads()) schedule(static)
    for (int32_t i = 0; i < m.rows; ++i) {
        const int32_t* b = m.col.data() + m.row_ptr[i];
        const int32_t* e = m.col.data() + m.row_ptr[(size_t)i + 1];
        t.row_ptr[(size_t)i + 1] = std::lower_bound(b, e, c1) - std::lower_bound(b, e, c0);
    }
    for (int32_t i = 0; i < m.rows; ++i) t.row_ptr[(size_t)i + 1] += t.row_ptr[i];
    t.col.resize((size_t)t.row_ptr[m.rows]); t.val.resize((size_t)t.row_ptr[m.rows]);
#pragma omp parallel for num_threads(host_threads()) schedule(static)
    for (int32_t i = 0; i < m.rows; ++i) {
        const int32_t* b = m.col.data() + m.row_ptr[i];
        const int32_t* e = m.col.data() + m.row_ptr[(size_t)i + 1];
        const int64_t k0 = m.row_ptr[i] + (std::lower_bound(b, e, c0) - b);
        const int64_t n = t.row_ptr[(size_t)i + 1] - t.row_ptr[i];
        std::copy_n(m.col.data() + k0, n, t.col.data() + t.row_ptr[i]);
        std::copy_n(m.val.data() + k0, n, t.val.data() + t.row_ptr[i]);
    }
    return t;
}

// Offsets [o0, o1) from the (scaled) diagonal of a CSR matrix as its own CSR: row i keeps its columns in
// [i*cols/rows + o0, i*cols/rows + o1) (global column ids are kept: x is shared).  open_lo / open_hi: no lower / upper bound.
Csr band_tile(const Csr& m, int64_t o0, int64_t o1, bool open_lo, bool open_hi) {
    Csr t;
    t.rows = m.rows; t.cols = m.cols;
    t.row_ptr.assign((size_t)m.rows + 1, 0);
    auto bounds = [&](int32_t i, int32_t& c0, int32_t& c1) {
        const int64_t cen = (int64_t)i * m.cols / m.rows;
        c0 = open_lo ? 0 : (int32_t)std::max<int64_t>(0, std::min<int64_t>(m.cols, cen + o0));
        c1 = open_hi ? m.cols : (int32_t)std::max<int64_t>(0, std::min<int64_t>(m.cols, cen + o1));
    };
#pragma omp parallel for num_threads(host_threads()) schedule(static)
    for (int32_t i = 0; i < m.rows; ++i) {
        int32_t c0, c1; bounds(i, c0, c1);
        const int32_t* b = m.col.data() + m.row_ptr[i];
        const int32_t* e = m.col.data() + m.row_ptr[(size_t)i + 1];
        t.row_ptr[(size_t)i + 1] = c1 > c0 ? std::lower_bound(b, e, c1) - std::lower_bound(b, e, c0) : 0;
    }
    for (int32_t i = 0; i < m.rows; ++i) t.row_ptr[(size_t)i + 1] += t.row_ptr[i];
    t.col.resize((size_t)t.row_ptr[m.rows]); t.val.resize((size_t)t.row_ptr[m.rows]);
#pragma omp parallel for num_threads(host_threads()) schedule(static)
    for (int32_t i = 0; i < m.rows; ++i) {
        int32_t c0, c1; bounds(i, c0, c1);
        const int32_t* b = m.col.data() + m.row_ptr[i];
        const int32_t* e = m.col.data() + m.row_ptr[(size_t)i + 1];
        const int64_t k0 = m.row_ptr[i] + (std::lower_bound(b, e, c0) - b);
        const int64_t n = t.row_ptr[(size_t)i + 1] - t.row_ptr[i];
        std::copy_n(m.col.data() + k0, n, t.col.data() + t.row_ptr[i]);
        std::copy_n(m.val.data() + k0, n, t.val.data() + t.row_ptr[i]);
    }
    return t;
}

// Width (in columns) of a column tile; 0 = no tiling.  Tiling applies only when the plan of the whole
// matrix gathers x through L2 (no LDS window) and x exceeds `tile_bytes` (default 4 MiB = one XCD's
// L2, i.e. tiles of 2-4 MiB; HISPMV_COL_TILE_BYTES overrides, 0 disables).
int32_t column_tile_width(int32_t cols, int64_t tile_bytes) {
    if (tile_bytes <= 0 || (int64_t)cols * 4 <= tile_bytes + tile_bytes / 2) return 0;
    // 2, 4 or 8 tiles: in a batch call the tiles of a matrix run in the same round, each pinned to 8 / tiles of the
    // 8 XCDs, so that an XCD's L2 holds one tile's part of x (more than 8 tiles' worth of x: 8 larger tiles)
    int64_t tiles = ((int64_t)cols * 4 + tile_bytes - 1) / tile_bytes;
    tiles = tiles <= 2 ? 2 : tiles <= 4 ? 4 : 8;
    const int64_t w = (((int64_t)cols + tiles - 1) / tiles + 63) & ~63LL;     // equal tiles, 256-byte aligned
    return (int32_t)w;
}

// The launch plan of a part, then its device layout.  Two steps: a matrix that becomes a tile stream needs the plan (the
// decision reads it) but not the device layout of the slice stream it drops (0.14 s on soc-Pokec's shape).
void plan_part(HostPart& p, int n_cus) {
    for (const FixEntry& f : p.st.fix) (f.len <= kFixShortMax ? p.fix_short : p.fix_long).push_back(f);
    p.plan = make_plan(p.st, n_cus);     // also rewrites the column field of LDS-staged groups
}
void pack_part(HostPart& p, const FormatOptions& opt) {
    if (opt.decide_only) return;
    p.dstream = pack_device_stream(p.st, p.plan, !opt.device_layout);
    if (!opt.device_layout) p.st.words = WordVec();   // the device layout replaces the host words (else the loader uploads them and lays them out there)
}
void finish_part(HostPart& p, int n_cus, const FormatOptions& opt) { plan_part(p, n_cus); pack_part(p, opt); }

// The batch layout of a whole-matrix slice stream (HostPart::has_batch_layout): planned for half the CUs, kept when it is the same
// kind of plan with longer groups.  Call BEFORE pack_part (which may release the words).
void add_batch_layout(HostPart& p, int n_cus, const FormatOptions& opt) {
    if (!opt.batch_layout || opt.decide_only || n_cus < 2 || p.plan.lds_floats <= 0 || p.plan.group_slices >= opt.batch_group_below || p.st.n_slices < opt.batch_min_slices) return;
    if (p.plan.group_slices != (p.st.n_slices + (int64_t)n_cus * p.plan.per_cu - 1) / ((int64_t)n_cus * p.plan.per_cu)) return;      // (resident plans only)
    WordVec keep = std::move(p.st.words);         // (the copy below takes the headers, the fix list and the sizes, not 8 bytes per element)
    SliceStream alt = p.st;
    p.st.words = std::move(keep);
    alt.words = unplanned_words(p.st, p.plan);    // the words with their columns again
    // groups four times as long where such a plan is the same kind of plan (its windows still fit), else three times, else twice
    LaunchPlan q;
    bool found = false;
    for (int div = opt.batch_group_div; div >= 2 && !found; --div) {
        // (only the resident configuration of the first plan: the other five cannot be "the same kind of plan with longer groups")
        q = make_plan(alt, std::max(1, n_cus / div), opt.batch_plan_search ? -1 : p.plan.block_threads == 1024 ? kPlanCfgResident1024 : p.plan.block_threads == 512 ? kPlanCfgResident512 : -1);
        found = q.block_threads == p.plan.block_threads && q.group_slices > p.plan.group_slices && q.lds_floats > 0 && q.ytile_floats == p.plan.ytile_floats;
        if (!found) alt.words = unplanned_words(p.st, p.plan);      // (make_plan rewrote the column fields of the words it staged)
    }
    if (!found) return;
    p.batch_dstream = pack_device_stream(alt, q, !opt.device_layout);
    if (opt.device_layout) p.batch_words = std::move(alt.words);
    p.batch_plan = std::move(q);
    p.has_batch_layout = true;
}

// "No x window can pay", decided from 32 samples of the CSR instead of from the launch plan of a slice stream: in every sample --
// 8192 consecutive entries, the elements of the SMALLEST group any plan has (8 slices) -- at least 90 % of the entries sit in a
// 64-byte block of x of their own.  A window then holds one element per staged block whatever the group size (larger groups touch
// more blocks, not fewer per element), which is what make_plan prices as "gather through L2" (cost >= 0.8).  Lets the loader skip
// building a slice stream it would drop for a tile stream: 44 + 7 ms of 154 on soc-Pokec's shape.
bool sampled_scattered(const Csr& csr) {
    const int64_t nnz = csr.nnz();
    constexpr int64_t kSample = 8 * kSliceElems;
    constexpr int kSamples = 32;
    if (nnz < kSamples * kSample) return false;
    int bad = 0;
#pragma omp parallel for num_threads(host_threads()) schedule(static) reduction(+ : bad)
    for (int sidx = 0; sidx < kSamples; ++sidx) {
        const int64_t k0 = (nnz - kSample) * sidx / (kSamples - 1);
        std::vector<int32_t> b((size_t)kSample);
        for (int64_t k = 0; k < kSample; ++k) b[(size_t)k] = csr.col[(size_t)(k0 + k)] / kFragBlock;
        std::sort(b.begin(), b.end());
        const int64_t distinct = std::unique(b.begin(), b.end()) - b.begin();
        if (distinct * 10 < kSample * 9) ++bad;
    }
    return bad == 0;
}

}  // namespace

std::vector<uint8_t> window_membership(const Csr& csr, const LaunchPlan& plan) {
    const int64_t G = plan.group_slices;
    const std::vector<int64_t> eoff = stream_row_offsets(csr.rows, csr.row_ptr.data());
    std::vector<uint8_t> inside((size_t)csr.nnz());
#pragma omp parallel for num_threads(host_threads()) schedule(dynamic, 1024)
    for (int32_t i = 0; i < csr.rows; ++i) {
        for (int64_t k = csr.row_ptr[i]; k < csr.row_ptr[(size_t)i + 1]; ++k) {
            const int64_t g = ((eoff[i] + (k - csr.row_ptr[i])) / kSliceElems) / G;
            if ((size_t)g >= plan.groups.size()) { inside[(size_t)k] = 0; continue; }
            const GroupDesc& gd = plan.groups[(size_t)g];
            const Frag* f0 = plan.frags.data() + gd.frag_begin;
            const Frag* f1 = f0 + gd.frag_count;
            const int32_t c = csr.col[(size_t)k];
            const Frag* it = std::upper_bound(f0, f1, c, [](int32_t col, const Frag& f) { return col < f.col_start; });
            inside[(size_t)k] = it != f0 && c < (it - 1)->col_start + (it - 1)->len;
        }
    }
    return inside;
}

FormatChoice choose_format(Csr&& csr, SliceStream* prebuilt, int n_cus, const FormatOptions& opt,
                           const std::function<void(const char*)>& lap_fn) {
    auto lap = [&](const char* what) { if (lap_fn) lap_fn(what); };
    FormatChoice out;
    out.parts.emplace_back();
    // The whole-matrix slice stream and its launch plan -- unless the samples already say that no window pays and a tile stream is
    // on the cards (auto mode, at least tts_min_nnz entries): then the plan is the planner's "gather through L2" default, and the
    // stream is built only if the tile stream is turned down further below.
    bool have_stream = true;
    if (!prebuilt && opt.format_mode == 2 && !opt.no_stream_skip && csr.nnz() >= std::max<int64_t>(opt.tts_min_nnz, 1 << 20) && sampled_scattered(csr)) {
        have_stream = false;
        LaunchPlan& d = out.parts[0].plan;
        d.block_threads = 256; d.group_slices = 8; d.lds_floats = 0; d.per_cu = 4;
        lap("sampled: scattered (no slice stream)");
    } else {
        out.parts[0].st = prebuilt ? std::move(*prebuilt) : build_stream(csr);      // (the device preprocessor hands its stream over)
        lap("slice stream (host) / adopt");
        plan_part(out.parts[0], n_cus);            // (its device layout: once the format is decided, below)
        lap("launch plan");
    }
    auto ensure_stream = [&]() {
        if (have_stream) return;
        out.parts[0] = HostPart();
        out.parts[0].st = build_stream(csr);
        plan_part(out.parts[0], n_cus);
        have_stream = true;
        lap("slice stream + plan (tile stream turned down)");
    };
    // Column tiling when the whole-matrix plan has to gather x through L2:
    //  * x a little too large for one LDS window (<= 2 windows): two tiles, each with its x window in LDS;
    //  * x larger than an XCD's L2: L2-sized tiles.
    int32_t tw = 0, tbase = 0;
    const LaunchPlan& whole = out.parts[0].plan;
    // columns the matrix actually uses: a block of a larger matrix (a rank's shard: x is replicated at full length)
    // is tiled over ITS column range, not over the width of x; the 0.1 % of elements at either end do not count
    // (a shard also holds a few rows of the next block) -- they go to the first / last tile, which are open-ended
    int32_t cmin = INT32_MAX, cmax = -1;
    const int64_t nnz_all = csr.nnz();
#pragma omp parallel for num_threads(host_threads()) reduction(min : cmin) reduction(max : cmax) schedule(static)
    for (int64_t k = 0; k < nnz_all; ++k) { cmin = std::min(cmin, csr.col[(size_t)k]); cmax = std::max(cmax, csr.col[(size_t)k]); }
    if (cmax >= cmin) {
        // (bins of 2^shift columns, at most 1024 of them; every 4th entry of a large matrix: the cut is a 0.1 % quantile.
        // A 64-bit division per entry made this pass 44 ms on soc-Pokec's shape.)
        constexpr int kBins = 1024;
        int shift = 0;
        while ((((int64_t)cmax - cmin) >> shift) >= kBins) ++shift;
        const int64_t bin_w = 1ll << shift;
        const int64_t stride = nnz_all >= (4 << 20) ? 4 : 1;
        std::vector<int64_t> hist(kBins, 0);
#pragma omp parallel num_threads(host_threads())
        {
            std::vector<int64_t> local(kBins, 0);
#pragma omp for schedule(static) nowait
            for (int64_t k = 0; k < nnz_all; k += stride) local[(size_t)((uint32_t)(csr.col[(size_t)k] - cmin) >> shift)] += stride;
#pragma omp critical
            for (int b = 0; b < kBins; ++b) hist[(size_t)b] += local[(size_t)b];
        }
        const int64_t cut = nnz_all / 1000;
        int lo = 0, hi = kBins - 1;
        for (int64_t acc = 0; lo < hi && acc + hist[(size_t)lo] <= cut; ++lo) acc += hist[(size_t)lo];
        for (int64_t acc = 0; hi > lo && acc + hist[(size_t)hi] <= cut; --hi) acc += hist[(size_t)hi];
        const int64_t qlo = cmin + lo * bin_w, qhi = std::min<int64_t>(cmax, cmin + (hi + 1) * bin_w - 1);
        cmin = (int32_t)qlo; cmax = (int32_t)qhi;
    }
    const int32_t used = cmax >= cmin ? cmax - (cmin & ~63) + 1 : 0;
    // Scattered columns (no window pays): the transposed tile stream, when sorting a row tile's elements by column brings
    // several of them onto each cache line of x (hispmv_tts.h).  Takes the place of the L2-sized column tiles below.
    // (from 1 M entries: a tile is a long latency chain -- column-order pass, barrier, row-order pass per block --, the
    // smallest matrices of the benchmark set run faster as slice streams: three alternating rounds, step of the set with
    // two launch streams, threshold 64 K: 342-345 us, 1 M: 342-346 us, 4 M: 352-355 us; matrices alone: trans5 16.7 vs 10.1 us)
    const int64_t tts_min = opt.tts_min_nnz;
    // Candidates: plans without a window, and plans whose window leaves more than 5 % of the gathers to L2 (a wide band
    // without column reuse between rows: the pessimistic stand-ins of PFlow_742 / Si41Ge41H72, 86 and 69 us with a 128 KiB
    // window of the most used blocks) -- there the two formats are compared by their L2 requests per element.
    const int64_t all_elems = have_stream ? out.parts[0].st.n_slices * (int64_t)kSliceElems : nnz_all;
    const double slice_requests = whole.lds_floats == 0 ? 1.0
                                 : ((double)whole.global_elems + (double)whole.staged_floats / kFragBlock) / (double)std::max<int64_t>(all_elems, 1);
    // (x at most two windows wide is cut into two column tiles that each run from LDS: mouse_gene 56 us that way, 77 us as
    // a tile stream)
    const bool two_windows = used > 0 && used <= 2 * kMaxLdsFloats && opt.col_tile_bytes > 0 && (have_stream ? out.parts[0].st.n_slices : nnz_all / kSliceElems) >= 4096;
    const bool candidate = whole.lds_floats == 0 || (whole.global_elems * 20 > all_elems && !two_windows);
    // (x of at most 256 KiB stays in L1 / L2 whatever the order of the gathers: the slice stream's per-element gathers are cheap
    // there and a tile is a longer latency chain -- the 1024 x 8192 layer of apps/model_test.py: 10.0 us as a slice stream
    // against 15.0 us as a tile stream, 8 vectors through `linear` 33 against 58 us.  Not 1 MiB: Si41Ge41H72 as an
    // unstructured band, 742 KB of x, is 39.5 us as a tile stream and 45 as a slice stream; the pessimistic family's step
    // 0.329 -> 0.363 ms with that threshold.)
    const bool x_is_small = (int64_t)used * 4 <= (256 << 10);
    // STRAY SPLIT (round 4; tile_kind 3): the plan has a window, but a few per cent of the elements lie outside it -- long-range
    // couplings of an otherwise banded / clustered matrix.  One stray element makes its whole GROUP take 8-byte elements and the
    // two-way gather (window read + predicated gather through L2): with 2 % of the entries re-drawn at random columns EVERY group of
    // the PFlow_742 stand-in has strays and the matrix falls from 0.70 to 0.43 of the roofline (tools/standin_sweep.py,
    // profiles/r4_standin_sweep.json).  The matrix is split A = A_in + A_out by exactly that criterion -- an element is "in" when the
    // window of ITS group (the planner's fragment list) holds its 64-byte block of x: part 0 = A_in keeps the rows, gets clean
    // windows and 6-byte elements again, part 1 = A_out (the strays) is a small scattered matrix that gathers through L2 and writes
    // alpha * A_out * x into a partial vector the tail launch adds -- the reference's hybrid row distribution (dense part on the
    // PEs' own rows, the rest through the shared-row network, spmv-helper.cpp:265-347) in the coordinates of the x window.
    // (Strays that fit the kernel's stray slots -- at most 64 per slice, hispmv_plan.h -- are served there for the price of one
    // gather per slice: the split is for what the slots do not cover.)
    // (not for the small-matrix class -- 256-thread plans: crystk03 with 2 % strays 0.152 -> 0.109 of the roofline split, two grids
    // and a tail around a 10 us kernel)
    if (opt.stray_split && opt.format_mode != 1 && whole.lds_floats > 0 && nnz_all >= (1 << 20) && whole.block_threads >= 512 &&
        whole.global_elems * 1000 > all_elems && whole.global_elems * 100 <= 15 * all_elems &&
        stray_slot_coverage(out.parts[0].st, whole) < 0.9) {
        const std::vector<uint8_t> inside = window_membership(csr, out.parts[0].plan);
        auto take = [&](bool want) {
            Csr t;
            t.rows = csr.rows; t.cols = csr.cols;
            t.row_ptr.assign((size_t)csr.rows + 1, 0);
#pragma omp parallel for num_threads(host_threads()) schedule(static)
            for (int32_t i = 0; i < csr.rows; ++i) {
                int64_t n = 0;
                for (int64_t k = csr.row_ptr[i]; k < csr.row_ptr[(size_t)i + 1]; ++k) n += (inside[(size_t)k] != 0) == want;
                t.row_ptr[(size_t)i + 1] = n;
            }
            for (int32_t i = 0; i < csr.rows; ++i) t.row_ptr[(size_t)i + 1] += t.row_ptr[i];
            t.col.resize((size_t)t.row_ptr[csr.rows]); t.val.resize((size_t)t.row_ptr[csr.rows]);
#pragma omp parallel for num_threads(host_threads()) schedule(static)
            for (int32_t i = 0; i < csr.rows; ++i) {
                int64_t o = t.row_ptr[i];
                for (int64_t k = csr.row_ptr[i]; k < csr.row_ptr[(size_t)i + 1]; ++k)
                    if ((inside[(size_t)k] != 0) == want) { t.col[(size_t)o] = csr.col[(size_t)k]; t.val[(size_t)o] = csr.val[(size_t)k]; ++o; }
            }
            return t;
        };
        std::vector<HostPart> parts(2);
        {
            Csr in = take(true);
            parts[0].st = build_stream(in);
            plan_part(parts[0], n_cus);
        }
        const HostPart& q = parts[0];
        // accepted when the band part really comes out clean (<= 0.5 % of its elements outside the windows)
        if (q.plan.lds_floats > 0 && q.plan.global_elems * 200 <= q.st.n_slices * (int64_t)kSliceElems) {
            pack_part(parts[0], opt);
            Csr outp = take(false);
            parts[1].st = build_stream(outp);
            finish_part(parts[1], n_cus, opt);
            lap("stray split");
            out.parts = std::move(parts);
            out.tile_kind = 3;
            csr = Csr{};
            return out;
        }
        lap("stray split (rejected)");
    }
    // BAND TILES: a banded matrix whose band is wider than an LDS window -- every group of rows touches band + rows columns --
    // is cut along the DIAGONAL: part t holds the elements whose offset from the (scaled) diagonal lies in the t-th of P equal
    // ranges of the band.  A group of a part then touches (band / P + its rows) columns: a window that fits, every element in
    // it, 6-byte elements from LDS instead of per-element gathers through the cache -- the reference's column tiling
    // (tileAndPad spmv-helper.cpp:242-263) in the coordinates of a band.  Part 0 writes y, the others partial vectors that
    // the tail launch merges; all parts share one grid.  (PFlow_742 as an unstructured band +-20000: 82 us as a tile stream.)
    if (candidate && opt.band_tiles && opt.format_mode != 1 && nnz_all >= (4 << 20) && csr.rows > 1 && !x_is_small) {
        constexpr int kBins = 4096;
        const int64_t span = (int64_t)csr.cols + csr.rows;            // offsets lie in (-cols, cols)
        int shift = 0;
        while ((2 * span >> shift) >= kBins) ++shift;
        std::vector<int64_t> hist(kBins, 0);
#pragma omp parallel num_threads(host_threads())
        {
            std::vector<int64_t> local(kBins, 0);
#pragma omp for schedule(static) nowait
            for (int32_t i = 0; i < csr.rows; ++i) {
                const int64_t cen = (int64_t)i * csr.cols / csr.rows;
                for (int64_t k = csr.row_ptr[i]; k < csr.row_ptr[(size_t)i + 1]; ++k) local[(size_t)((csr.col[(size_t)k] - cen + span) >> shift)]++;
            }
#pragma omp critical
            for (int b = 0; b < kBins; ++b) hist[(size_t)b] += local[(size_t)b];
        }
        const int64_t cut = nnz_all / 1000;
        int lo = 0, hi = kBins - 1;
        for (int64_t acc = 0; lo < hi && acc + hist[(size_t)lo] <= cut; ++lo) acc += hist[(size_t)lo];
        for (int64_t acc = 0; hi > lo && acc + hist[(size_t)hi] <= cut; --hi) acc += hist[(size_t)hi];
        const int64_t dmin = ((int64_t)lo << shift) - span, dmax = (((int64_t)hi + 1) << shift) - span - 1;
        const int64_t W = dmax - dmin + 1;
        constexpr int64_t kTarget = 20 * 1024;                        // offsets per part: leaves ~12 K floats of window for the rows of a group
        lap("band histogram");
        if (W > kTarget && W <= 8 * kTarget && W < (int64_t)used) {
            const int P = (int)((W + kTarget - 1) / kTarget);
            const int64_t width = (((W + P - 1) / P) + 63) & ~63LL;
            std::vector<HostPart> parts;
            bool ok = true;
            for (int t = 0; t < P && ok; ++t) {
                Csr tile = band_tile(csr, dmin + t * width, dmin + (t + 1) * width, t == 0, t == P - 1);
                parts.emplace_back();
                parts.back().st = build_stream(tile);
                finish_part(parts.back(), std::max(1, n_cus / P), opt);
                const HostPart& q = parts.back();
                ok = q.plan.lds_floats > 0 && q.plan.global_elems * 50 <= q.st.n_slices * (int64_t)kSliceElems;
            }
            lap("band tiles");
            if (ok) {
                out.parts = std::move(parts);
                out.tile_kind = 2; out.col_tile_width = (int)width; out.col_tile_base = (int)dmin;
                csr = Csr{};
                return out;
            }
        }
    }
    if (candidate && opt.format_mode != 0 && ((nnz_all >= tts_min && !x_is_small) || (opt.format_mode == 1 && nnz_all >= 64 * 1024))) {
        lap("column range / histogram");
        TtsGeometry g0;
        g0.zero_fill = opt.tts_geometry == 4;        // HISPMV_TTS_GEOMETRY=zerofill: the standard sizes, no filler words (experiment)
        TtsStream ts = build_tts(csr, 0, g0);
        lap("tile stream packer");
        if (ts.lines_per_gather <= 8.0 && opt.tts_small) {
            // experiment (off by default: measured slower): cheap gathers -> the half-LDS geometry, two workgroups per CU
            TtsGeometry small;
            small.max_slots = kTtsSmallSlots; small.max_rows = kTtsSmallRows; small.tiles_wanted = 512;
            TtsStream t2 = build_tts(csr, 0, small);
            if (t2.lines_per_gather <= 16.0) ts = std::move(t2);
        }
        // (a tile is one workgroup's work; the packer cuts rows longer than a tile and a quarter into pieces and orders
        // the tiles longest first, so this only rejects what is left: tiles capped by their row count next to full ones)
        const bool balanced = ts.max_tile_slots <= 2 * (ts.total_slots / std::max<int64_t>(1, (int64_t)ts.tiles.size())) + 4096;
        const bool fewer_requests = ts.lines_per_gather / 64.0 + 0.1 < slice_requests;     // (+0.1: the two passes and barriers of a block)
        if (opt.format_mode == 1 || (ts.lines_per_gather <= opt.tts_max_lines && balanced && fewer_requests)) {
            // The TALL geometry (hispmv_tts.h): when a gather of the 8 K-row tiles still touches many lines of x and x is
            // larger than an XCD's L2, the matrix becomes two column parts of 16 K-row tiles -- the same number of tiles
            // and elements per tile over half the column range (soc-Pokec: 23 -> 13 lines per gather), and in a batch
            // call each part is pinned to four XCDs whose L2s then hold its half of x.  HISPMV_TTS_GEOMETRY=standard|tall|paired|auto
            // (standard is the default: the gathers are bound by the lines they pull through the cache, and the tall tiles
            // pay for fewer lines with 46 % more row-order slots -- DESIGN.md 2.2).
            std::vector<TtsStream> tall;
            const bool paired = opt.tts_geometry == 3;
            const bool tallgap = opt.tts_geometry == 5;
            const bool want_tall = opt.tts_geometry == 1 || paired || tallgap ||
                                   (opt.tts_geometry == 2 && ts.lines_per_gather > 16.0 && (int64_t)used * 4 > (4 << 20) && csr.rows >= 64 * kTtsTallRows);
            if (want_tall) {
                const int n_parts = opt.tall_parts > 0 ? opt.tall_parts : kTtsTallParts;
                const std::vector<int32_t> cuts = tts_column_cuts(csr, n_parts);
                double lines = 0; int64_t slices = 0;
                bool ok = true;
                for (int q = 0; q < n_parts && ok; ++q) {
                    Csr part = csr_column_range(csr, q == 0 ? 0 : cuts[(size_t)q - 1], q + 1 == n_parts ? csr.cols : cuts[(size_t)q]);
                    TtsGeometry geo = tallgap ? tts_tallgap_geometry(n_cus, n_parts) : paired ? tts_paired_geometry(n_cus) : tts_tall_geometry(n_cus, n_parts);
                    if (opt.tall_rows > 0) geo.max_rows = opt.tall_rows;
                    if (opt.tall_slots > 0) geo.max_slots = opt.tall_slots;
                    if (opt.tall_tiles > 0) geo.tiles_wanted = opt.tall_tiles;
                    if (opt.tall_zero_fill >= 0) geo.zero_fill = opt.tall_zero_fill != 0;
                    tall.push_back(build_tts(part, 0, geo));
                    const TtsStream& t = tall.back();
                    lines += t.lines_per_gather * (double)t.col_base.size(); slices += (int64_t)t.col_base.size();
                    ok = t.max_tile_slots <= 2 * (t.total_slots / std::max<int64_t>(1, (int64_t)t.tiles.size())) + 4096;
                }
                // (auto: only when it pays -- at least a quarter fewer lines per gather)
                if (!ok || (opt.tts_geometry == 2 && lines / (double)std::max<int64_t>(slices, 1) > 0.75 * ts.lines_per_gather)) tall.clear();
            }
            out.format = 1;
            if (!tall.empty()) {
                out.parts.clear();
                double lines = 0; int64_t slices = 0;
                for (TtsStream& t : tall) {
                    out.parts.emplace_back();
                    HostPart& p = out.parts.back();
                    p.is_tts = true;
                    lines += t.lines_per_gather * (double)t.col_base.size(); slices += (int64_t)t.col_base.size();
                    p.tts = std::move(t);
                }
                out.tts_lines_per_gather = lines / (double)std::max<int64_t>(slices, 1);
                out.tile_kind = 1;
                out.col_tile_width = tts_column_cuts(csr, (int)out.parts.size())[0];
            } else {
                HostPart& p = out.parts[0];
                p.is_tts = true;
                p.tts = std::move(ts);
                p.st = SliceStream{}; p.dstream = DeviceStream{}; p.fix_short = {}; p.fix_long = {};
                out.tts_lines_per_gather = p.tts.lines_per_gather;
            }
            csr = Csr{};
            return out;
        }
    }
    ensure_stream();
    // (a window that leaves more than a tenth of the gathers to L2 counts as "does not fit" here)
    const bool spilling = whole.lds_floats > 0 && whole.global_elems * 10 > out.parts[0].st.n_slices * (int64_t)kSliceElems;
    if (used > 0 && (whole.lds_floats == 0 || spilling)) {
        if (used <= 2 * kMaxLdsFloats && opt.col_tile_bytes > 0 && out.parts[0].st.n_slices >= 4096)
            tw = ((used + 1) / 2 + 63) & ~63;
        else if (whole.lds_floats == 0) tw = column_tile_width(used, opt.col_tile_bytes);
        tbase = cmin & ~63;
    }
    if (tw > 0) {
        out.col_tile_width = tw; out.col_tile_base = tbase; out.tile_kind = 1;
        out.parts.clear();
        for (int64_t c0 = tbase; c0 <= cmax; c0 += tw) {
            const bool first = c0 == tbase, last = c0 + tw > cmax;           // the end tiles are open-ended
            Csr tile = column_tile(csr, first ? 0 : (int32_t)c0, last ? csr.cols : (int32_t)(c0 + tw));
            out.parts.emplace_back();
            out.parts.back().st = build_stream(tile);
            // the tiles of a matrix run in one grid: a two-window tile (resident plan: one long chunk per workgroup, its x
            // window staged once) is planned for its share of the CUs
            const int n_tiles = (int)((cmax - tbase) / tw + 1);
            finish_part(out.parts.back(), used <= 2 * kMaxLdsFloats ? std::max(1, n_cus / std::max(1, n_tiles)) : n_cus, opt);
        }
        // two tiles were meant to bring the x window into LDS: if they still gather through L2, tiling only
        // costs a launch and a read-modify-write of y -- go back to the single stream
        bool lds_goal = used <= 2 * kMaxLdsFloats, all_lds = true;
        for (auto& p : out.parts)
            all_lds = all_lds && p.plan.lds_floats > 0 && p.plan.global_elems * 50 <= p.st.n_slices * (int64_t)kSliceElems;
        if ((lds_goal && !all_lds) || out.parts.size() < 2) {
            out.parts.clear();
            out.col_tile_width = 0; out.col_tile_base = 0; out.tile_kind = 0;
            out.parts.emplace_back();
            out.parts[0].st = build_stream(csr);
            finish_part(out.parts[0], n_cus, opt);
        } else if (!lds_goal) {
            const size_t np = out.parts.size();
            bool same = np == 2 || np == 4 || np == 8;
            for (auto& p : out.parts) same = same && p.plan.block_threads == out.parts[0].plan.block_threads && p.plan.lds_floats == 0;
            out.l2_tiles = same;
        }
    }
    csr = Csr{};
    lap("column tiles (if any)");
    if (tw == 0) { add_batch_layout(out.parts[0], n_cus, opt); pack_part(out.parts[0], opt); }       // the whole-matrix stream stays: its device layout(s) now
    lap("device layout");
    return out;
}

std::vector<std::pair<int, int>> order_step_queue(const std::vector<double>& slice_costs, const std::vector<double>& tile_costs, int n_wg, int mode) {
    const std::vector<double>* cost[2] = {&slice_costs, &tile_costs};
    std::vector<std::pair<int, int>> out;
    out.reserve(slice_costs.size() + tile_costs.size());
    if (mode == 2) {
        for (size_t i = 0; i < tile_costs.size(); ++i) out.emplace_back(1, (int)i);
        for (size_t i = 0; i < slice_costs.size(); ++i) out.emplace_back(0, (int)i);
        return out;
    }
    std::vector<int> idx[2];
    double total = 0;
    for (int k = 0; k < 2; ++k) {
        idx[k].resize(cost[k]->size());
        for (size_t i = 0; i < idx[k].size(); ++i) { idx[k][i] = (int)i; total += (*cost[k])[i]; }
        std::stable_sort(idx[k].begin(), idx[k].end(), [&](int a, int b) { return (*cost[k])[(size_t)a] > (*cost[k])[(size_t)b]; });
    }
    const double T = total / std::max(1, n_wg);          // the step if nothing idles
    size_t pos[2] = {0, 0};
    while (pos[0] < idx[0].size() || pos[1] < idx[1].size()) {
        int k;
        if (pos[0] >= idx[0].size()) k = 1;
        else if (pos[1] >= idx[1].size()) k = 0;
        else {
            const double c0 = (*cost[0])[(size_t)idx[0][pos[0]]], c1 = (*cost[1])[(size_t)idx[1][pos[1]]];
            // long tiles alternate with the longest slice items (mode 0: one to one; modes 3, 4: two / three slice items per long tile)
            // (mode >= 16, experiments: cycles of (mode >> 4) long tiles and (mode & 15) slice items)
            const int nt = mode >= 16 ? std::max(1, mode >> 4) : 1, ns = mode >= 16 ? std::max(1, mode & 15) : mode == 3 ? 2 : mode == 4 ? 3 : 1;
            if ((mode == 0 || mode == 3 || mode == 4 || mode >= 16) && c1 > 0.25 * T) k = (pos[0] + pos[1]) % (size_t)(nt + ns) < (size_t)nt ? 1 : 0;
            else k = c1 > c0 ? 1 : 0;                                                     // longest first
        }
        out.emplace_back(k, idx[k][pos[k]]);
        ++pos[k];
    }
    return out;
}

}  // namespace hispmv
