// hispmv_abi.cpp -- the C ABI of libhispmv.so (include/hispmv.h): context, matrix handles,
// arena accounting, HBM upload and launches.  MI355X counterpart of the reference's
// FpgaHandle (pyhispmv/src/fpga_handle.cpp:40-388), which owns the XRT device, the
// per-channel matrix arena and the kernel run object.  Host-side HIP runtime calls only;
// the kernels live in hispmv_kernels.hip, the preprocessor in hispmv_prep.cpp.
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

using namespace hispmv;

namespace {

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
        SpmvDeviceMatrix dev;                      //   of a slice stream
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

}  // namespace

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
    hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
    int batch_streams = 2;
    int batch_order = 0;         // HISPMV_BATCH_ORDER: 0 tile streams first (default), 1 small slice grids first
    bool batch_lanes_heavy_first = true;   // HISPMV_BATCH_LANES=rr: plain round-robin lanes
    bool batch_graphs = true;    // HISPMV_BATCH_GRAPH=0: no HIP graph replay of batch calls
    // ... for calls that stream at least this much: forking to and joining from a side stream costs ~13 us (measured on
    // the three model_test layers: 49.9 us on one stream, 63.1 on two; the 20-matrix set: 353 -> 344 us with two)
    int64_t batch_streams_min_bytes = 256ll << 20;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ms = -1.0f;
    std::mutex mu;
    std::string err;
    std::vector<std::unique_ptr<Matrix>> mats;
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
    int64_t cap_stage = 0;
    // hispmv_spmv_device_batch: the launches of one call signature (handles, vectors, beta == 0 or not) with their device
    // tables, built on the first call and replayed afterwards
    struct BatchLaunch {
        int kind = 0;                                   // 0 slice kernels of one workgroup size, 1 fix-up of cut rows, 2 merge of column-tile
                                                        // partial vectors, 3 transposed tile streams, 4 dense overlay (GeMV)
        std::vector<TtsEntry> tts;                      // kind 3
        std::vector<GemvEntry> gemv;                    // kind 4: the dense overlay handles of the call in one grid
        std::vector<const SpmvDeviceMatrix*> parts;     // kinds 0, 1
        std::vector<float*> ys;                         // kind 1: where each part's cut rows live (y or a partial vector)
        std::vector<int32_t> rows;                      // kind 2
        std::vector<uint8_t> item_tiles;                // kind 0: parts per item (> 1: the XCD-pinned column tiles of one matrix)
        std::vector<int32_t> fix_counts;                // kind 5 (fix-up + merge in one launch): short fix entries per part; rows = merged matrices
        void* d_table = nullptr;
        void* d_table2 = nullptr;                       // kind 5: the TailMergeEntry table
        int lane = 0;                                   // main launches: 0 = the caller's stream, k > 0 = side stream k - 1
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
    // Rows shared between slices: "fixup" = second tiny launch, "lookback" = single launch with carry
    // granules, "auto" (default) = look-back without ticket when the whole grid is co-resident (small
    // matrices, where the extra launch costs as much as the kernel), fix-up otherwise.
    int carry_mode = 2;          // 0 fixup, 1 lookback, 2 auto (HISPMV_CARRY)
    // COO -> CSR -> slice stream: 0 on the host (OpenMP), 1 on the device (hispmv_prep_device.hip), 2 auto = device from
    // 2 M entries (HISPMV_PREP=host|device|auto); both give the same stream byte for byte
    int prep_mode = 2;
    DevicePrepTimes last_prep_times;
    // device format of matrices whose plan gathers x through L2: 0 slice stream always, 1 transposed tile stream whenever
    // the plan has no window, 2 auto = transposed tile stream when its gathers touch <= 32 cache lines of x per wave
    // instruction (HISPMV_FORMAT=slices|tts|auto)
    FormatOptions format_opts;   // HISPMV_FORMAT / _TTS_GEOMETRY / _BAND_TILES / _COL_TILE_BYTES / _TTS_MIN_NNZ (hispmv_choose.h)
    // geometry of a transposed tile stream: 0 the 8 K-row tiles always, 1 the tall geometry (two column parts of 16 K-row
    // tiles) for every tile stream, 2 auto (HISPMV_TTS_GEOMETRY=standard|tall|auto)
    int n_cus = 256;
};

struct hispmv_prep {
    Csr csr;
    SliceStream st;
    LaunchPlan plan;
    TtsStream tts;
};

namespace {

thread_local std::string g_create_err;
thread_local std::string g_prep_err;

int fail(hispmv_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg; else g_create_err = msg;
    return code;
}
int hip_fail(hispmv_ctx* c, hipError_t e, const char* what) {
    return fail(c, HISPMV_EDEVICE, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(c, call)                                        \
    do {                                                        \
        hipError_t e_ = (call);                                 \
        if (e_ != hipSuccess) return hip_fail((c), e_, #call);  \
    } while (0)

// Every device / pinned allocation of the library is released through these: a failing free (a pointer freed twice, a
// pointer the runtime does not know) is counted, and hispmv_free_failures() lets a test read the count.
std::atomic<int64_t> g_free_failures{0};
template <class T> void dev_free(T*& p) {
    if (p && hipFree((void*)p) != hipSuccess) { g_free_failures++; (void)hipGetLastError(); }
    p = nullptr;
}
template <class T> void host_free(T*& p) {
    if (p && hipHostFree((void*)p) != hipSuccess) { g_free_failures++; (void)hipGetLastError(); }
    p = nullptr;
}

void free_batch_plans(hispmv_ctx* c) {
    for (auto& p : c->batch_plans) {
        for (auto& g : p.slot) {
            if (g.launched && g.done) (void)hipEventSynchronize(g.done);
            if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
            if (g.done) { (void)hipEventDestroy(g.done); g.done = nullptr; }
            g.launched = false;
        }
        if (p.graph_src) { (void)hipGraphDestroy(p.graph_src); p.graph_src = nullptr; }
        for (auto& l : p.launches) { dev_free(l.d_table); dev_free(l.d_table2); }
    }
    c->batch_plans.clear();
}

int64_t sparse_device_bytes(const SliceStream& st, const DeviceStream& ds) {
    return (int64_t)ds.bytes.size() + (int64_t)st.hdr.size() * 16 + (int64_t)st.fix.size() * 16 +
           (int64_t)st.n_slices * 12 + 8;
}

// A bounded in-kernel wait that expired leaves 1 in the context's error word.
int check_device_error(hispmv_ctx* c) {
    // callers have synchronised the stream the kernels ran on; the word lives in host memory
    const int flag = *(volatile int*)c->h_err;
    if (flag) {
        *(volatile int*)c->h_err = 0;
        return fail(c, HISPMV_EDEVICE, "carry hand-off between slices timed out (lost or overlapping launch on one handle)");
    }
    return HISPMV_OK;
}

void free_matrix_device(Matrix& m) {
    for (void*& p : m.allocs) dev_free(p);
    m.allocs.clear();
    for (auto& p : m.parts) p.dev = SpmvDeviceMatrix{};
    m.d_dense = nullptr; m.d_ypart = nullptr; m.d_fix_of_row = nullptr;
    m.loaded = false;
}

int ensure_vec(hispmv_ctx* c, float** p, int64_t* cap, int64_t n) {
    if (n <= *cap) return HISPMV_OK;
    dev_free(*p);
    *cap = 0;
    const int64_t want = std::max<int64_t>(n, 1024);
    HIP_TRY(c, hipMalloc((void**)p, (size_t)want * sizeof(float)));
    *cap = want;
    return HISPMV_OK;
}

// Registers a prepared sparse matrix with the context (capacity check = the reference's
// "offset + size > MAX_BUFFER_SIZE_BYTES -> return -1", fpga_handle.cpp:192-195).  The format and tiling decision itself is
// host-only code: choose_format (hispmv_choose.cpp).
int add_sparse(hispmv_ctx* c, Csr&& csr, double t_csr, SliceStream* prebuilt = nullptr) {
    auto t0 = std::chrono::steady_clock::now();
    // HISPMV_PREP_TRACE=1: the phases of the host side of preprocessing on stderr (diagnostics)
    static const bool trace = std::getenv("HISPMV_PREP_TRACE") != nullptr;
    auto lap = [&, last = t0](const char* what) mutable {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[hispmv prep] %-28s %7.1f ms\n", what, std::chrono::duration<double>(now - last).count() * 1e3);
        last = now;
    };
    if (trace) std::fprintf(stderr, "[hispmv prep] %-28s %7.1f ms (device: upload %.1f csr %.1f offsets %.1f stream %.1f download %.1f)\n", "COO -> CSR (-> stream)", t_csr * 1e3,
                            c->last_prep_times.upload * 1e3, c->last_prep_times.csr_device * 1e3, c->last_prep_times.offsets_host * 1e3,
                            c->last_prep_times.stream_device * 1e3, c->last_prep_times.download * 1e3);
    auto m = std::make_unique<Matrix>();
    m->rows = csr.rows; m->cols = csr.cols; m->nnz = csr.nnz();
    FormatChoice ch = choose_format(std::move(csr), prebuilt, c->n_cus, c->format_opts, lap);
    m->format = ch.format; m->tile_kind = ch.tile_kind; m->col_tile_width = ch.col_tile_width; m->col_tile_base = ch.col_tile_base;
    m->l2_tiles = ch.l2_tiles; m->tts_lines_per_gather = ch.tts_lines_per_gather;
    for (HostPart& hp : ch.parts) {
        m->parts.emplace_back();
        static_cast<HostPart&>(m->parts.back()) = std::move(hp);
    }
    for (auto& p : m->parts) {
        if (p.is_tts) {
            m->n_slices += (int64_t)p.tts.col_base.size(); m->n_elems += p.tts.nnz + p.tts.n_fillers; m->n_split += (int64_t)p.tts.fix.size() / 4;
            m->device_bytes += p.tts.bytes();
        } else {
            m->n_slices += p.st.n_slices; m->n_elems += p.st.n_elems; m->n_split += (int64_t)p.st.fix.size();
            m->device_bytes += sparse_device_bytes(p.st, p.dstream) + (int64_t)p.dstream.groups.size() * 4 + (int64_t)p.plan.frags.size() * 16;
            m->compact_slices += p.dstream.compact_slices;
        }
    }
    if (m->parts.size() > 1) m->device_bytes += (int64_t)(m->parts.size() - 1) * kMaxBatch * m->rows * 4;   // partial vectors of parts t > 0
    if (m->format == 1) {
        m->plan_threads = m->parts[0].tts.geometry.threads; m->plan_group = m->parts[0].tts.geometry.max_slots / kTtsChunk; m->plan_lds = 0;
    } else {
        m->plan_threads = m->parts[0].plan.block_threads; m->plan_group = m->parts[0].plan.group_slices; m->plan_lds = m->parts[0].plan.lds_floats;
    }
    m->prep_seconds = t_csr + std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (c->arena_used + m->device_bytes > c->arena_budget) return HISPMV_FULL;
    c->arena_used += m->device_bytes;
    m->index = (int)c->mats.size();
    c->mats.push_back(std::move(m));
    return (int)c->mats.size() - 1;
}

template <class T>
int upload(hispmv_ctx* c, Matrix& m, const T* host, size_t count, const T** dev_out) {
    *dev_out = nullptr;
    if (count == 0) return HISPMV_OK;
    void* d = nullptr;
    HIP_TRY(c, hipMalloc(&d, count * sizeof(T)));
    m.allocs.push_back(d);
    HIP_TRY(c, hipMemcpyAsync(d, host, count * sizeof(T), hipMemcpyHostToDevice, c->stream));
    *dev_out = (const T*)d;
    return HISPMV_OK;
}

int spmv_batch_locked(hispmv_ctx* c, int32_t n, const int32_t* idx, const float* const* d_x, const float* const* d_bias,
                      float* const* d_y, float alpha, float beta, hipStream_t s);

int launch_matrix(hispmv_ctx* c, Matrix& m, const float* d_x, const float* d_bias, float* d_y,
                  float alpha, float beta, hipStream_t s, bool fixup_only = false) {
    if (!m.dense && (m.parts.size() > 1 || (m.format == 1 && m.parts[0].tdev.zero_fill)) && m.index >= 0) {
        // column tiles: all of them in ONE grid (+ one fix-up, one merge launch) through the batch machinery -- launched
        // one after the other each tile had the chip to itself for half the work (mouse_gene 48 -> 40 us)
        const int32_t idx = m.index;
        return spmv_batch_locked(c, 1, &idx, &d_x, &d_bias, &d_y, alpha, beta, s);
    }
    if (m.dense) {
        hipError_t e = launch_gemv(m.d_dense, m.rows, m.cols, d_x, d_bias, d_y, alpha, beta, s);
        if (e != hipSuccess) return hip_fail(c, e, "launch_gemv");
        return HISPMV_OK;
    }
    if (m.format == 1) {
        hipError_t e = launch_tts(m.parts[0].tdev, d_x, d_bias, d_y, alpha, beta, s);
        if (e != hipSuccess) return hip_fail(c, e, "launch_tts");
        return HISPMV_OK;
    }
    for (size_t t = 0; t < m.parts.size(); ++t) {
        // column tile 0 computes alpha*A_0*x + beta*bias into y; tile t > 0 writes alpha*A_t*x into its partial vector
        hipError_t e = (t == 0) ? launch_spmv(m.parts[t].dev, d_x, d_bias, d_y, alpha, beta, s, fixup_only)
                                : launch_spmv(m.parts[t].dev, d_x, nullptr, m.d_ypart + (t - 1) * (size_t)kMaxBatch * m.rows, alpha, 0.0f, s, fixup_only);
        if (e != hipSuccess) return hip_fail(c, e, "launch_spmv");
    }
    if (m.parts.size() > 1) {
        hipError_t e = launch_merge_parts(d_y, m.d_ypart, (int)m.parts.size() - 1, (int64_t)kMaxBatch * m.rows, m.rows, 1, 0, 0, s);
        if (e != hipSuccess) return hip_fail(c, e, "launch_merge_parts");
    }
    return HISPMV_OK;
}

// `vecs` vectors with a shared bias (FpgaHandle::linear): the reference relaunches its kernel per vector
// (fpga_handle.cpp:366-379); here up to 8 (dense) / 4 (sparse) vectors share one pass over the matrix when the plan allows.
int launch_matrix_vectors(hispmv_ctx* c, Matrix& m, int64_t vecs, const float* d_x, const float* d_bias, float* d_y,
                          float alpha, float beta, hipStream_t s, bool fixup_only) {
    // (`linear` always takes the fix-up carry variant: one vector or many, every vector gets the same bits)
    if (vecs == 1) return launch_matrix(c, m, d_x, d_bias, d_y, alpha, beta, s, fixup_only);
    if (m.dense) {
        hipError_t e = launch_gemv_batched(m.d_dense, m.rows, m.cols, vecs, d_x, d_bias, d_y, alpha, beta, s);
        if (e != hipSuccess) return hip_fail(c, e, "launch_gemv_batched");
        return HISPMV_OK;
    }
    if (m.format == 1) {          // transposed tile stream: up to 8 vectors per launch (same bits as a single-vector call each)
        int64_t k = 0;
        while (k < vecs) {
            // 4 or 2 vectors through every pass over the words where the tiles are small enough; else up to 8 in one launch, one after the other
            int nv = 1;
            if (m.parts.size() == 1 && !m.parts[0].tdev.zero_fill) {
                nv = tts_batch_width(m.parts[0].tdev, vecs - k);
                if (nv < 2) nv = (int)std::min<int64_t>(vecs - k, kTtsMaxVectors);
            }
            if (nv >= 2) {
                hipError_t e = launch_tts_batched(m.parts[0].tdev, nv, d_x + k * m.cols, d_bias, d_y + k * m.rows, alpha, beta, s);
                if (e != hipSuccess) return hip_fail(c, e, "launch_tts_batched");
            } else {
                const int rc = launch_matrix(c, m, d_x + k * m.cols, d_bias, d_y + k * m.rows, alpha, beta, s);
                if (rc != HISPMV_OK) return rc;
            }
            k += nv;
        }
        return HISPMV_OK;
    }
    int64_t k = 0;
    while (k < vecs) {
        int nv = beta != 0.0f ? kMaxBatch : 1;       // (linear always has beta = 1; the batched kernel's tile 0 reads a bias)
        for (auto& p : m.parts) nv = std::min(nv, spmv_batch_width(p.dev, vecs - k));
        const float* xk = d_x + k * m.cols;
        float* yk = d_y + k * m.rows;
        if (nv < 2) {
            int rc = launch_matrix(c, m, xk, d_bias, yk, alpha, beta, s, true);
            if (rc != HISPMV_OK) return rc;
            k += 1;
            continue;
        }
        for (size_t t = 0; t < m.parts.size(); ++t) {
            // tile 0: shared bias; tile t > 0: the nv partial vectors of that tile, no bias
            hipError_t e = (t == 0) ? launch_spmv_batched(m.parts[t].dev, nv, xk, d_bias, 0, yk, alpha, beta, s)
                                    : launch_spmv_batched(m.parts[t].dev, nv, xk, nullptr, 0, m.d_ypart + (t - 1) * (size_t)kMaxBatch * m.rows, alpha, 0.0f, s);
            if (e != hipSuccess) return hip_fail(c, e, "launch_spmv_batched");
        }
        if (m.parts.size() > 1) {
            hipError_t e = launch_merge_parts(yk, m.d_ypart, (int)m.parts.size() - 1, (int64_t)kMaxBatch * m.rows, m.rows, nv, m.rows, m.rows, s);
            if (e != hipSuccess) return hip_fail(c, e, "launch_merge_parts");
        }
        k += nv;
    }
    return HISPMV_OK;
}


}  // namespace

// ------------------------------------------------------------------------------------------------
// `stream` NULL = the context's stream, exactly as in hispmv_spmv_device / hispmv_spmv_device_batch: a caller that passes NULL
// everywhere gets its SpMVs and its boundary kernels on ONE queue, in order.  (Until round 3 NULL meant HIP's null stream here:
// the boundary kernels then did not wait for SpMVs issued with NULL on the context's non-blocking stream.)
HISPMV_API int hispmv_boundary_pack(hispmv_ctx* c, const float* const* d_last, const float* d_mask, float* d_send, int32_t n, void* stream) {
    if (!c) return HISPMV_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (n < 0 || (n > 0 && (!d_last || !d_mask || !d_send))) return fail(c, HISPMV_EINVAL, "bad boundary_pack arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    if (stream) c->user_stream = (hipStream_t)stream;
    const hipError_t e = launch_boundary_pack(d_last, d_mask, d_send, n, stream ? (hipStream_t)stream : c->stream);
    return e == hipSuccess ? HISPMV_OK : hip_fail(c, e, "launch_boundary_pack");
}

HISPMV_API int hispmv_boundary_apply(hispmv_ctx* c, float* const* d_first, const float* d_recv, const float* d_weights, int32_t n,
                                     int32_t world, void* stream) {
    if (!c) return HISPMV_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (n < 0 || world < 1 || (n > 0 && (!d_first || !d_recv || !d_weights))) return fail(c, HISPMV_EINVAL, "bad boundary_apply arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    if (stream) c->user_stream = (hipStream_t)stream;
    const hipError_t e = launch_boundary_apply(d_first, d_recv, d_weights, n, world, stream ? (hipStream_t)stream : c->stream);
    return e == hipSuccess ? HISPMV_OK : hip_fail(c, e, "launch_boundary_apply");
}

HISPMV_API const char* hispmv_version(void) { return "hispmv-amd 0.2.0 gfx950"; }

HISPMV_API int64_t hispmv_free_failures(void) { return g_free_failures.load(); }

HISPMV_API const char* hispmv_last_error(const hispmv_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

HISPMV_API void hispmv_destroy(hispmv_ctx* c);

HISPMV_API int hispmv_create(hispmv_ctx** out, const char* xclbin_path, int device_id, int a, int b, int cc,
                             int urams, int fp_acc_latency, int dense, int pre_acc, int row_dist) {
    if (!out) return fail(nullptr, HISPMV_EINVAL, "out is NULL");
    *out = nullptr;
    host_threads();       // OpenMP threads of the preprocessor = the CPUs this process may use (cgroup quota)
    // same argument checks as fpga_handle.cpp:51-52,70-71
    if (device_id < 0) return fail(nullptr, HISPMV_EINVAL, "Device ID must be a non-negative integer.");
    if (!xclbin_path || !*xclbin_path) return fail(nullptr, HISPMV_EINVAL, "XCLBIN path is empty.");
    if (a <= 0 || b <= 0 || cc <= 0) return fail(nullptr, HISPMV_EINVAL, "channel counts must be positive");
    if ((a * 8) % (cc * 16) != 0)   // spmv-helper.cpp:15
        return fail(nullptr, HISPMV_EINVAL, "Number of PEs should be an integer multiple of Number of FP32 elements in output vector");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, HISPMV_EDEVICE, std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
    if (device_id >= ndev) return fail(nullptr, HISPMV_EDEVICE, "device_id out of range");
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) return hip_fail(nullptr, e, "hipGetDeviceProperties");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, HISPMV_EDEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    if ((e = hipSetDevice(device_id)) != hipSuccess) return hip_fail(nullptr, e, "hipSetDevice");
    if ((e = prepare_spmv_kernels()) != hipSuccess) return hip_fail(nullptr, e, "hipFuncSetAttribute(max dynamic LDS)");

    auto c = std::make_unique<hispmv_ctx>();
    c->device = device_id;
    c->num_ch_A = a; c->num_ch_B = b; c->num_ch_C = cc; c->urams_per_pe = urams; c->fp_acc_latency = fp_acc_latency;
    c->dense_overlay = dense != 0; c->pre_accumulator = pre_acc != 0; c->row_dist_net = row_dist != 0;
    c->arena_budget = (int64_t)a * 256 * 1024 * 1024;      // fpga_handle.h:12, one 256 MiB bank per A channel
    if (const char* env = std::getenv("HISPMV_ARENA_BYTES")) { long long v = std::atoll(env); if (v > 0) c->arena_budget = v; }
    // from here on the context owns HIP objects: a failure releases them through hispmv_destroy
    auto give_up = [&](hipError_t err, const char* what) { hispmv_destroy(c.release()); return hip_fail(nullptr, err, what); };
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return give_up(e, "hipStreamCreate");
    if ((e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev1)) != hipSuccess) return give_up(e, "hipEventCreate");
    if ((e = hipHostMalloc((void**)&c->h_err, sizeof(int), hipHostMallocMapped)) != hipSuccess) return give_up(e, "hipHostMalloc(err flag)");
    *c->h_err = 0;
    if ((e = hipHostGetDevicePointer((void**)&c->d_err, c->h_err, 0)) != hipSuccess) return give_up(e, "hipHostGetDevicePointer(err flag)");
    c->format_opts = FormatOptions::from_env();
    if (const char* env = std::getenv("HISPMV_CARRY"))
        c->carry_mode = !std::strcmp(env, "fixup") ? 0 : !std::strcmp(env, "lookback") ? 1 : !std::strcmp(env, "ticket") ? 3 : !std::strcmp(env, "resident") ? 5 : 2;
    if (const char* env = std::getenv("HISPMV_BATCH_GRAPH")) c->batch_graphs = std::atoi(env) != 0;
    if (const char* env = std::getenv("HISPMV_BATCH_ORDER")) c->batch_order = !std::strcmp(env, "small_first") ? 1 : 0;
    if (const char* env = std::getenv("HISPMV_BATCH_LANES")) c->batch_lanes_heavy_first = std::strcmp(env, "rr") != 0;
    if (const char* env = std::getenv("HISPMV_BATCH_STREAMS")) { c->batch_streams = std::max(1, std::min(3, std::atoi(env))); c->batch_streams_min_bytes = 0; }
    for (int i = 0; i < 2; ++i) {
        if ((e = hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking)) != hipSuccess) return give_up(e, "hipStreamCreate(side)");
        if ((e = hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming)) != hipSuccess) return give_up(e, "hipEventCreate(join)");
    }
    if ((e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming)) != hipSuccess) return give_up(e, "hipEventCreate(fork)");
    if (const char* env = std::getenv("HISPMV_PREP"))
        c->prep_mode = !std::strcmp(env, "host") ? 0 : !std::strcmp(env, "device") ? 1 : 2;
    c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char* env = std::getenv("HISPMV_PLAN_CUS")) { const int v = std::atoi(env); if (v > 0) c->n_cus = v; }   // experiments
    *out = c.release();
    return HISPMV_OK;
}

HISPMV_API void hispmv_destroy(hispmv_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& m : c->mats) free_matrix_device(*m);
    dev_free(c->d_x);
    dev_free(c->d_y);
    host_free(c->h_err);
    host_free(c->h_stage);
    free_batch_plans(c);
    for (int i = 0; i < 2; ++i) { if (c->side[i]) (void)hipStreamDestroy(c->side[i]); if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

HISPMV_API int hispmv_set_arena_bytes(hispmv_ctx* c, int64_t bytes) {
    if (!c || bytes < 0) return HISPMV_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    c->arena_budget = bytes;
    return HISPMV_OK;
}
HISPMV_API int64_t hispmv_arena_bytes_used(const hispmv_ctx* c) { return c ? c->arena_used : 0; }

// COO -> handle, on the device or on the host (hispmv_ctx::prep_mode)
static int add_from_coo(hispmv_ctx* c, int32_t rows, int32_t cols, int64_t nnz, const int32_t* r, const int32_t* cl, const float* v) {
    auto t0 = std::chrono::steady_clock::now();
    const bool on_device = c->prep_mode == 1 || (c->prep_mode == 2 && nnz >= (2 << 20));
    if (on_device) {
        HIP_TRY(c, hipSetDevice(c->device));
        Csr csr; SliceStream st; std::string err;
        if (!prep_on_device(rows, cols, nnz, r, cl, v, csr, st, c->last_prep_times, err))
            return fail(c, err.find("outside") != std::string::npos || err.find("dimension") != std::string::npos ? HISPMV_EINVAL : HISPMV_EDEVICE, err);
        double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return add_sparse(c, std::move(csr), t, &st);
    }
    Csr csr = coo_to_csr(rows, cols, nnz, r, cl, v);
    double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return add_sparse(c, std::move(csr), t);
}

HISPMV_API int hispmv_create_sparse_handle(hispmv_ctx* c, const int32_t* r, const int32_t* cl, const float* v,
                                           int64_t nnz, int32_t rows, int32_t cols) {
    if (!c) return HISPMV_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (rows <= 0 || cols <= 0 || nnz < 0 || (nnz > 0 && (!r || !cl || !v))) return fail(c, HISPMV_EINVAL, "bad sparse matrix arguments");
    try {
        return add_from_coo(c, rows, cols, nnz, r, cl, v);
    } catch (const std::out_of_range& ex) { return fail(c, HISPMV_EINVAL, ex.what());
    } catch (const std::bad_alloc&) { return fail(c, HISPMV_ENOMEM, "host out of memory");
    } catch (const std::exception& ex) { return fail(c, HISPMV_EINVAL, ex.what()); }
}

HISPMV_API int hispmv_create_sparse_handle_from_mtx(hispmv_ctx* c, const char* path, int flavor) {
    if (!c) return HISPMV_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (!path || (flavor != 0 && flavor != 1)) return fail(c, HISPMV_EINVAL, "bad arguments");
    try {
        auto t0 = std::chrono::steady_clock::now();
        Coo coo = read_mtx(path, (MtxFlavor)flavor);
        (void)t0;                                     // like the reference, file parsing is not "Pre-processing Time"
        return add_from_coo(c, coo.rows, coo.cols, (int64_t)coo.r.size(), coo.r.data(), coo.c.data(), coo.v.data());
    } catch (const std::runtime_error& ex) { return fail(c, HISPMV_EIO, ex.what());
    } catch (const std::bad_alloc&) { return fail(c, HISPMV_ENOMEM, "host out of memory");
    } catch (const std::exception& ex) { return fail(c, HISPMV_EINVAL, ex.what()); }
}

HISPMV_API int hispmv_create_sparse_handle_from_csr(hispmv_ctx* c, const int32_t* rp, const int32_t* ci, const float* va,
                                                    int32_t rows, int32_t cols) {
    if (!c) return HISPMV_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (rows <= 0 || cols <= 0 || !rp) return fail(c, HISPMV_EINVAL, "bad CSR arguments");
    if (rows >= (1 << 30) || cols >= (1 << 30)) return fail(c, HISPMV_EINVAL, "dimension >= 2^30 is not supported");
    try {
        auto t0 = std::chrono::steady_clock::now();
        Csr csr;
        csr.rows = rows; csr.cols = cols;
        csr.row_ptr.resize((size_t)rows + 1);
        for (int32_t i = 0; i <= rows; ++i) csr.row_ptr[i] = rp[i];
        const int64_t nnz = rp[rows];
        if (rp[0] != 0 || nnz < 0) return fail(c, HISPMV_EINVAL, "row_ptr must start at 0");
        for (int32_t i = 0; i < rows; ++i) if (rp[i + 1] < rp[i]) return fail(c, HISPMV_EINVAL, "row_ptr must be non-decreasing");
        if (nnz > 0 && (!ci || !va)) return fail(c, HISPMV_EINVAL, "col_idx / values are NULL");
        csr.col.assign(ci, ci + nnz); csr.val.assign(va, va + nnz);
        for (int64_t k = 0; k < nnz; ++k) if (ci[k] < 0 || ci[k] >= cols) return fail(c, HISPMV_EINVAL, "CSR column outside matrix");
        sort_rows_by_column(csr);     // rows with unsorted columns (scipy: has_sorted_indices == False) are sorted, stably
        double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return add_sparse(c, std::move(csr), t);
    } catch (const std::bad_alloc&) { return fail(c, HISPMV_ENOMEM, "host out of memory");
    } catch (const std::exception& ex) { return fail(c, HISPMV_EINVAL, ex.what()); }
}

HISPMV_API int hispmv_create_dense_handle(hispmv_ctx* c, const float* vals, int32_t rows, int32_t cols) {
    if (!c) return HISPMV_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->dense_overlay)   // assert at spmv-helper.cpp:718
        return fail(c, HISPMV_ENOTDENSE, "Hardware is not built with Dense Overlay, cannot support dense workload");
    if (rows <= 0 || cols <= 0 || !vals) return fail(c, HISPMV_EINVAL, "bad dense matrix arguments");
    try {
        auto t0 = std::chrono::steady_clock::now();
        auto m = std::make_unique<Matrix>();
        m->dense = true; m->rows = rows; m->cols = cols; m->nnz = (int64_t)rows * cols;   // spmv-helper.cpp:722
        m->device_bytes = m->nnz * 4;
        if (c->arena_used + m->device_bytes > c->arena_budget) return HISPMV_FULL;
        m->dense_host.assign(vals, vals + m->nnz);
        m->prep_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        c->arena_used += m->device_bytes;
        m->index = (int)c->mats.size();
    c->mats.push_back(std::move(m));
        return (int)c->mats.size() - 1;
    } catch (const std::bad_alloc&) { return fail(c, HISPMV_ENOMEM, "host out of memory"); }
}

HISPMV_API int hispmv_load_matrices(hispmv_ctx* c) {
    if (!c) return HISPMV_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(c, hipSetDevice(c->device));
    for (auto& mp : c->mats) {
        Matrix& m = *mp;
        if (m.loaded) continue;
        int rc;
        std::vector<std::vector<int32_t>> tts_fix_rows;       // tile streams: the rows cut into pieces, per part (fix list order)
        if (m.dense) {
            const float* d = nullptr;
            if ((rc = upload(c, m, m.dense_host.data(), m.dense_host.size(), &d)) != HISPMV_OK) return rc;
            m.d_dense = const_cast<float*>(d);
        } else if (m.format == 1) {
          for (Matrix::Part& p : m.parts) {
            TtsStream& ts = p.tts;
            tts_fix_rows.emplace_back();
            for (size_t k = 0; k + 3 < ts.fix.size(); k += 4) tts_fix_rows.back().push_back(ts.fix[k]);
            const uint8_t* dw = nullptr; const int32_t* dcb = nullptr; const uint16_t* dfl = nullptr; const int32_t* dci = nullptr;
            const TtsTile* dt = nullptr; const TtsBlock* db = nullptr;
            if ((rc = upload(c, m, ts.words.data(), ts.words.size(), &dw)) != HISPMV_OK) return rc;
            if ((rc = upload(c, m, ts.col_base.data(), ts.col_base.size(), &dcb)) != HISPMV_OK) return rc;
            if ((rc = upload(c, m, ts.flags.data(), ts.flags.size(), &dfl)) != HISPMV_OK) return rc;
            if ((rc = upload(c, m, ts.chunk_info.data(), ts.chunk_info.size(), &dci)) != HISPMV_OK) return rc;
            if ((rc = upload(c, m, ts.tiles.data(), ts.tiles.size(), &dt)) != HISPMV_OK) return rc;
            if ((rc = upload(c, m, ts.blocks.data(), ts.blocks.size(), &db)) != HISPMV_OK) return rc;
            TtsDeviceMatrix& d = p.tdev;
            d.words = dw; d.col_base = dcb; d.flags = dfl; d.chunk_info = (const int2*)dci; d.tiles = (const int4*)dt; d.blocks = (const int4*)db;
            d.n_tiles = (int32_t)ts.tiles.size(); d.rows = m.rows; d.cols = m.cols;
            if (!ts.fix.empty()) {       // rows cut into pieces: carry slots + the slice stream's fix-up entries
                const int32_t* dfix = nullptr;
                if ((rc = upload(c, m, ts.fix.data(), ts.fix.size(), &dfix)) != HISPMV_OK) return rc;
                void* carry = nullptr;
                HIP_TRY(c, hipMalloc(&carry, (size_t)std::max(ts.n_carry, 1) * kTtsMaxVectors * sizeof(float)));      // one set per vector of a batched launch
                m.allocs.push_back(carry);
                HIP_TRY(c, hipMemsetAsync(carry, 0, (size_t)std::max(ts.n_carry, 1) * kTtsMaxVectors * sizeof(float), c->stream));
                d.n_carry = ts.n_carry;
                d.fix = (const int4*)dfix; d.n_fix = (int32_t)(ts.fix.size() / 4); d.carry = (float*)carry;
                // (the same three fields where the multi-matrix fix-up launch looks for them)
                p.dev.fix_short = d.fix; p.dev.n_fix_short = d.n_fix; p.dev.carry = d.carry; p.dev.n_fix_long = 0;
            }
            d.acc_floats = (ts.max_rows + 63) & ~63; d.threads = ts.geometry.threads;
            d.zero_fill = ts.geometry.zero_fill ? 1 : 0;
            d.staging_floats = ts.geometry.max_slots + 64;        // (the dummy slot of padding words sits behind the last real one)
            d.batch_stage_floats = ((ts.max_slots + kTtsChunk - 1) / kTtsChunk) * kTtsChunk + 64;
            // x in the LDS for short x (HISPMV_TTS_XLDS=1; off by default -- measured slower on the 1024 x 8192 layer of
            // apps/model_test.py: 16.7 against 15.1 us alone, 8 vectors 74 against 58 us: that layer's tiles are latency chains
            // of 8 K elements, not gather-bound)
            d.xlds_floats = (m.cols <= kTtsXldsMax && std::getenv("HISPMV_TTS_XLDS")) ? ((m.cols + 63) & ~63) : 0;
            if (((size_t)d.acc_floats + (size_t)d.staging_floats + 64) * 4 > 160 * 1024 - 256) return fail(c, HISPMV_EINVAL, "internal: tile stream exceeds the LDS of a CU");
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            p.tts = TtsStream{};
          }
        } else {
            for (auto& p : m.parts) {
                const uint8_t* dw = nullptr; const SliceHdr* dh = nullptr; const FixEntry *fs = nullptr, *fl = nullptr;
                const int32_t* dg = nullptr; const Frag* dfr = nullptr;
                const int64_t ns = p.st.n_slices;
                if ((rc = upload(c, m, p.dstream.groups.data(), p.dstream.groups.size(), &dg)) != HISPMV_OK) return rc;
                if ((rc = upload(c, m, p.plan.frags.data(), p.plan.frags.size(), &dfr)) != HISPMV_OK) return rc;
                if ((rc = upload(c, m, p.dstream.bytes.data(), p.dstream.bytes.size(), &dw)) != HISPMV_OK) return rc;
                // device header: {row_base, chain_len, rows ending in the slice, 1 if some of its elements lie outside
                // the group's x window} (the column window of a slice is only needed by the planner)
                std::vector<SliceHdr>& hh = p.st.hdr;
                int max_rows = 1;
                for (int64_t sl = 0; sl < ns; ++sl) {
                    const int nr = (sl + 1 < ns ? hh[sl + 1].row_base : m.rows) - hh[sl].row_base;
                    hh[sl].x_base = nr;
                    hh[sl].x_span = (!p.plan.slice_spills.empty() && p.plan.slice_spills[(size_t)sl]) ? 1 : 0;
                    max_rows = std::max(max_rows, nr);
                }
                if ((rc = upload(c, m, hh.data(), hh.size(), &dh)) != HISPMV_OK) return rc;
                if ((rc = upload(c, m, p.fix_short.data(), p.fix_short.size(), &fs)) != HISPMV_OK) return rc;
                if ((rc = upload(c, m, p.fix_long.data(), p.fix_long.size(), &fl)) != HISPMV_OK) return rc;
                // carry per slice; {carry, launch tag} granules and the group ticket of the look-back variant
                void *carry = nullptr, *gran = nullptr, *ticket = nullptr;
                const size_t n1 = (size_t)std::max<int64_t>(ns, 1);
                HIP_TRY(c, hipMalloc(&carry, n1 * kMaxBatch * sizeof(float)));      // one set per vector of a batched pass
                m.allocs.push_back(carry);
                HIP_TRY(c, hipMemsetAsync(carry, 0, n1 * kMaxBatch * sizeof(float), c->stream));
                HIP_TRY(c, hipMalloc(&gran, n1 * sizeof(unsigned long long)));
                m.allocs.push_back(gran);
                HIP_TRY(c, hipMemsetAsync(gran, 0, n1 * sizeof(unsigned long long), c->stream));
                HIP_TRY(c, hipMalloc(&ticket, sizeof(unsigned long long)));
                m.allocs.push_back(ticket);
                HIP_TRY(c, hipMemsetAsync(ticket, 0, sizeof(unsigned long long), c->stream));
                SpmvDeviceMatrix& d = p.dev;
                d.words = dw; d.hdr = (const int4*)dh; d.groups = (const int4*)dg; d.frags = (const int4*)dfr;
                d.fix_short = (const int4*)fs; d.fix_long = (const int4*)fl;
                d.carry = (float*)carry; d.gran = (unsigned long long*)gran; d.ticket = (unsigned long long*)ticket;
                d.err = c->d_err; d.launches = 0; d.ticket_launches = 0;
                d.n_slices = ns; d.n_groups = (ns + p.plan.group_slices - 1) / p.plan.group_slices;
                d.group_slices = p.plan.group_slices; d.block_threads = p.plan.block_threads; d.lds_floats = p.plan.lds_floats;
                d.ytile_floats = std::min(kSliceElems, (max_rows + 63) & ~63);
                const size_t lds_plain = (size_t)(d.lds_floats + d.ytile_floats * (d.block_threads / 64)) * 4;
                if (lds_plain > 160 * 1024 - 256) return fail(c, HISPMV_EINVAL, "internal: launch plan exceeds the LDS of a CU");
                const bool mailbox_fits = lds_plain + (size_t)d.group_slices * 8 <= 160 * 1024 - 256;   // look-back: 8 B per slice of a group
                d.n_fix_short = (int32_t)p.fix_short.size(); d.n_fix_long = (int32_t)p.fix_long.size();
                d.rows = m.rows; d.cols = m.cols;
                // co-residency of the whole grid: workgroups per CU by LDS and waves (conservative: <= 4 blocks,
                // <= 16 waves per CU; MI355X_MICROARCH.md "Residency")
                const int lds_b = std::max(1, (int)lds_plain + 64);
                const int per_cu = std::max(1, std::min({4, (160 * 1024) / lds_b, 16 / (d.block_threads / 64)}));
                const bool resident = d.n_groups <= (int64_t)c->n_cus * per_cu;
                const bool one_round = d.group_slices <= d.block_threads / 64;
                // carry_mode: 0 fix-up launch; 1 look-back for every plan, workgroups in blockIdx order (relies on the
                // dispatcher starting workgroups in increasing id order -- observed, not contractual; the wait is
                // bounded and reports instead of hanging); 3 the same with start-order tickets (contract-safe);
                // 2 (auto) look-back when the whole grid is co-resident (every workgroup is running, so waiting for an
                // earlier slice cannot deadlock) AND every wavefront has one slice (small matrices, where the second
                // launch costs as much as the kernel), fix-up otherwise; 5 ("resident") look-back for every co-resident
                // grid: correct, but measured slower than main kernel + fix-up launch on the large matrices
                // (PFlow_742 71.6 vs 65.2 us, TSOPF 35.5 vs 33.6: wavefronts that run ahead wait for slower neighbours)
                d.lookback = (c->carry_mode == 1 || c->carry_mode == 3 || (c->carry_mode == 2 && resident && one_round) ||
                              (c->carry_mode == 5 && resident)) && mailbox_fits;
                d.use_ticket = c->carry_mode == 3;
                if (m.parts.size() > 1) { d.lookback = false; d.use_ticket = false; }      // column tiles share one grid: fix-up launch
            }
        }
        if (!m.dense && m.parts.size() > 1) {
            void* yp = nullptr;
            HIP_TRY(c, hipMalloc(&yp, (m.parts.size() - 1) * (size_t)kMaxBatch * m.rows * sizeof(float)));
            m.allocs.push_back(yp);
            m.d_ypart = (float*)yp;
            // row -> fix entry of every part (short chains only: a part with a long chain keeps the two-launch tail)
            bool fusable = m.parts.size() <= (size_t)kTailMaxParts;
            for (auto& p : m.parts) fusable = fusable && (p.is_tts || p.fix_long.empty());
            if (fusable) {
                std::vector<int32_t> of((size_t)m.parts.size() * m.rows, -1);
                for (size_t t = 0; t < m.parts.size(); ++t) {
                    int32_t* o = of.data() + t * (size_t)m.rows;
                    if (m.parts[t].is_tts) { const std::vector<int32_t>& f = tts_fix_rows[t]; for (size_t k = 0; k < f.size(); ++k) o[f[k]] = (int32_t)k; }
                    else for (size_t k = 0; k < m.parts[t].fix_short.size(); ++k) o[m.parts[t].fix_short[k].row] = (int32_t)k;
                }
                const int32_t* d_of = nullptr;
                if ((rc = upload(c, m, of.data(), of.size(), &d_of)) != HISPMV_OK) return rc;
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                m.d_fix_of_row = const_cast<int32_t*>(d_of);
            }
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        // host copies are no longer needed
        for (auto& p : m.parts) { p.st = SliceStream{}; p.fix_short = {}; p.fix_long = {}; p.plan.groups = {}; p.plan.frags = {}; p.dstream = DeviceStream{}; }
        m.dense_host = {};
        m.loaded = true;
    }
    return HISPMV_OK;
}

HISPMV_API int hispmv_select_matrix(hispmv_ctx* c, uint32_t idx) {
    if (!c) return HISPMV_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (idx >= c->mats.size()) return fail(c, HISPMV_EINVAL, "Matrix idx out of range");   // fpga_handle.cpp:267-270
    c->selected = (int)idx;
    return HISPMV_OK;
}

static int run_host_vectors(hispmv_ctx* c, Matrix& m, const float* x, int64_t num_vecs, const float* bias,
                            float* y, float alpha, float beta, bool is_linear) {
    HIP_TRY(c, hipSetDevice(c->device));
    int rc;
    // device side: [x (num_vecs * cols) | bias (rows)] in one block, y in another
    const int64_t nx = (((int64_t)m.cols * num_vecs + 63) / 64) * 64, nb = m.rows, ny = (int64_t)m.rows * num_vecs;
    if ((rc = ensure_vec(c, &c->d_x, &c->cap_x, nx + nb)) != HISPMV_OK) return rc;
    if ((rc = ensure_vec(c, &c->d_y, &c->cap_y, ny)) != HISPMV_OK) return rc;
    float* const d_x = c->d_x;
    float* const d_bias = c->d_x + nx;
    const size_t bx = (size_t)m.cols * num_vecs * sizeof(float), bb = (size_t)nb * sizeof(float), by = (size_t)ny * sizeof(float);
    const bool staged = (nx + nb + ny) * (int64_t)sizeof(float) <= (8 << 20);     // small vectors: through pinned memory
    if (staged) {
        if (nx + nb + ny > c->cap_stage) {
            host_free(c->h_stage);      // (the staging block only; the batch tables are not touched by this path)
            c->cap_stage = 0;
            const int64_t want = std::max<int64_t>(nx + nb + ny, 1 << 16);
            HIP_TRY(c, hipHostMalloc((void**)&c->h_stage, (size_t)want * sizeof(float), hipHostMallocDefault));
            c->cap_stage = want;
        }
        std::memcpy(c->h_stage, x, bx);
        if (beta != 0.0f) std::memcpy(c->h_stage + nx, bias, bb);
        HIP_TRY(c, hipMemcpyAsync(d_x, c->h_stage, beta != 0.0f ? (size_t)nx * sizeof(float) + bb : bx, hipMemcpyHostToDevice, c->stream));
    } else {
        HIP_TRY(c, hipMemcpyAsync(d_x, x, bx, hipMemcpyHostToDevice, c->stream));
        if (beta != 0.0f) HIP_TRY(c, hipMemcpyAsync(d_bias, bias, bb, hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    if ((rc = launch_matrix_vectors(c, m, num_vecs, d_x, d_bias, c->d_y, alpha, beta, c->stream, is_linear)) != HISPMV_OK) return rc;
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    float* const h_y = staged ? c->h_stage + nx + nb : y;
    HIP_TRY(c, hipMemcpyAsync(h_y, c->d_y, by, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (staged) std::memcpy(y, h_y, by);
    if (hipEventElapsedTime(&c->last_ms, c->ev0, c->ev1) != hipSuccess) c->last_ms = -1.0f;
    return check_device_error(c);
}

HISPMV_API int hispmv_run_kernel(hispmv_ctx* c, const float* x, const float* bias, float* y, float alpha, float beta) {
    if (!c) return HISPMV_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (c->selected < 0) return fail(c, HISPMV_ESTATE, "Run Kernel called before selecting a matrix");   // assert :292
    Matrix& m = *c->mats[c->selected];
    if (!m.loaded) return fail(c, HISPMV_ESTATE, "run_kernel called before load_matrices");
    if (!x || !y || (beta != 0.0f && !bias)) return fail(c, HISPMV_EINVAL, "NULL vector");
    return run_host_vectors(c, m, x, 1, bias, y, alpha, beta, false);
}

HISPMV_API int hispmv_linear(hispmv_ctx* c, int idx, const float* x, int64_t x_len, const float* bias, float* y_out) {
    if (!c) return HISPMV_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (idx < 0 || idx >= (int)c->mats.size()) return fail(c, HISPMV_EINVAL, "Matrix idx out of range");
    Matrix& m = *c->mats[idx];
    if (!m.loaded) return fail(c, HISPMV_ESTATE, "linear called before load_matrices");
    if (!x || !bias || !y_out) return fail(c, HISPMV_EINVAL, "NULL vector");
    const int64_t num_vecs = x_len / m.cols;   // fpga_handle.cpp:336
    if (num_vecs <= 0) return fail(c, HISPMV_EINVAL, "x shorter than one input vector");
    return run_host_vectors(c, m, x, num_vecs, bias, y_out, 1.0f, 1.0f, true);   // alpha = beta = 1, :351-352
}

HISPMV_API int hispmv_spmv_device(hispmv_ctx* c, int idx, const float* d_x, const float* d_bias, float* d_y,
                                  float alpha, float beta, void* stream) {
    if (!c) return HISPMV_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (idx < 0 || idx >= (int)c->mats.size()) return fail(c, HISPMV_EINVAL, "Matrix idx out of range");
    Matrix& m = *c->mats[idx];
    if (!m.loaded) return fail(c, HISPMV_ESTATE, "spmv_device called before load_matrices");
    if (!d_x || !d_y || (beta != 0.0f && !d_bias)) return fail(c, HISPMV_EINVAL, "NULL device vector");
    HIP_TRY(c, hipSetDevice(c->device));
    if (stream) c->user_stream = (hipStream_t)stream;
    return launch_matrix(c, m, d_x, d_bias, d_y, alpha, beta, stream ? (hipStream_t)stream : c->stream);
}

// Builds the launches of a batch call: every column tile of every sparse handle is one "part"; parts with the same
// workgroup size share a grid (largest first, so that the small ones fill the tail), ONE fix-up launch finishes the cut
// rows of all parts (each on its own output: y for tile 0, the handle's partial vector for tile t > 0), ONE merge launch
// adds the partial vectors of the column-tiled matrices to their y.
static int build_batch_plan(hispmv_ctx* c, hispmv_ctx::BatchPlan& plan, int32_t n, const int32_t* idx, const float* const* d_x,
                            const float* const* bias, float* const* d_y, float beta) {
    struct Ref { int i; size_t t; };
    struct Item { std::vector<Ref> refs; int threads; int64_t slices; };
    std::vector<Item> items;
    std::vector<Ref> refs;
    auto dev_of = [&](const Ref& r) -> SpmvDeviceMatrix& { return c->mats[idx[r.i]]->parts[r.t].dev; };
    auto out_of = [&](const Ref& r) -> float* {
        Matrix& m = *c->mats[idx[r.i]];
        return r.t == 0 ? d_y[r.i] : m.d_ypart + (r.t - 1) * (size_t)kMaxBatch * m.rows;
    };
    const bool pin = !std::getenv("HISPMV_NO_XCD_PIN");
    auto upload_table0 = [&](hispmv_ctx::BatchLaunch& l, const void* host, size_t bytes) -> int {
        HIP_TRY(c, hipMalloc(&l.d_table, bytes));
        HIP_TRY(c, hipMemcpy(l.d_table, host, bytes, hipMemcpyHostToDevice));
        return HISPMV_OK;
    };
    {   // dense overlay handles: one grid for all of them, the largest first (the small ones fill its tail; launched one
        // after the other, the 512..2048-wide GeMVs of cpu/run_gemv.sh cost a launch latency each: 77 us for the five
        // sizes against 56 us of streaming)
        std::vector<int> dense;
        for (int i = 0; i < n; ++i) if (c->mats[idx[i]]->dense) dense.push_back(i);
        std::stable_sort(dense.begin(), dense.end(), [&](int a, int b) {
            const Matrix& ma = *c->mats[idx[a]]; const Matrix& mb = *c->mats[idx[b]];
            return (int64_t)ma.rows * ma.cols > (int64_t)mb.rows * mb.cols;
        });
        for (size_t k0 = 0; k0 < dense.size(); k0 += kMultiMax) {
            hispmv_ctx::BatchLaunch l;
            l.kind = 4;
            for (size_t k = k0; k < std::min(dense.size(), k0 + (size_t)kMultiMax); ++k) {
                const int i = dense[k];
                const Matrix& m = *c->mats[idx[i]];
                l.gemv.push_back(GemvEntry{m.d_dense, d_x[i], bias[i], d_y[i], m.rows, m.cols, beta, 0});
            }
            plan.launches.push_back(std::move(l));
            const int rc0 = upload_table0(plan.launches.back(), plan.launches.back().gemv.data(), plan.launches.back().gemv.size() * sizeof(GemvEntry));
            if (rc0 != HISPMV_OK) return rc0;
        }
    }
    // (+ 1 << 24: matrices whose x the kernel keeps in the LDS -- another LDS size, another kernel instantiation)
    auto tts_class = [](const Matrix& m) {
        return m.parts[0].tdev.staging_floats + (m.parts.size() == 1 && tts_x_in_lds(m.parts[0].tdev, 1) ? (1 << 24) : 0) + (m.parts[0].tdev.zero_fill ? (1 << 25) : 0);
    };
    std::vector<int> tts_classes;                // staging size = the geometry (hispmv_tts.h): small 13 K, standard 28 K, tall 23 K, paired 11 K
    for (int i = 0; i < n; ++i) {
        const Matrix& m = *c->mats[idx[i]];
        if (m.dense || m.format != 1) continue;
        const int cls = tts_class(m);
        if (std::find(tts_classes.begin(), tts_classes.end(), cls) == tts_classes.end()) tts_classes.push_back(cls);
    }
    std::sort(tts_classes.begin(), tts_classes.end());
    for (const int geometry : tts_classes) {
        // transposed tile streams: their row tiles share one grid per geometry (the launch's workgroup size and LDS size
        // are those of its entries: tiles of different geometries must not ride together)
        hispmv_ctx::BatchLaunch l;
        l.kind = 3;
        auto flush = [&]() -> int {
            if (l.tts.empty()) return HISPMV_OK;
            plan.launches.push_back(std::move(l));
            const int rc0 = upload_table0(plan.launches.back(), plan.launches.back().tts.data(), plan.launches.back().tts.size() * sizeof(TtsEntry));
            l = hispmv_ctx::BatchLaunch{};
            l.kind = 3;
            return rc0;
        };
        std::vector<int> order;                  // matrices with the longest tiles first (a CU holds one tile at a time)
        for (int i = 0; i < n; ++i) {
            const Matrix& m = *c->mats[idx[i]];
            if (m.dense || m.format != 1) continue;
            if (tts_class(m) == geometry) order.push_back(i);
        }
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
            const Matrix& ma = *c->mats[idx[a]]; const Matrix& mb = *c->mats[idx[b]];
            return ma.n_elems / std::max(1, ma.parts[0].tdev.n_tiles) > mb.n_elems / std::max(1, mb.parts[0].tdev.n_tiles);
        });
        for (int i : order) {
            Matrix& m = *c->mats[idx[i]];
            if (l.tts.size() + m.parts.size() > (size_t)kMultiMax) { const int rc0 = flush(); if (rc0 != HISPMV_OK) return rc0; }
            // the column parts of a tall-geometry matrix: one item, pinned to XCD subsets (part 0 writes y with the bias,
            // part t > 0 alpha*A_t*x into the handle's partial vector: the merge launch adds it)
            const bool pinned = pin && (m.parts.size() == 2 || m.parts.size() == 4);
            for (size_t t = 0; t < m.parts.size(); ++t) {
                l.tts.push_back(TtsEntry{m.parts[t].tdev, d_x[i], t == 0 ? bias[i] : nullptr, out_of(Ref{i, t}), t == 0 ? beta : 0.0f, 0});
                if (t == 0) l.weight += m.n_slices * (int64_t)kWideSliceBytes * 5 / 2;
                if (!pinned) l.item_tiles.push_back(1);
            }
            if (pinned) l.item_tiles.push_back((uint8_t)m.parts.size());
        }
        const int rc0 = flush();
        if (rc0 != HISPMV_OK) return rc0;
    }
    for (int i = 0; i < n; ++i) {
        const Matrix& m = *c->mats[idx[i]];
        if (m.dense || m.format == 1) continue;
        if (m.l2_tiles && pin) {                 // the L2-sized column tiles of a matrix: one item, pinned to XCD subsets
            Item it{{}, m.parts[0].dev.block_threads, 0};
            for (size_t t = 0; t < m.parts.size(); ++t) { it.refs.push_back(Ref{i, t}); it.slices += m.parts[t].dev.n_slices; }
            items.push_back(std::move(it));
        } else {
            for (size_t t = 0; t < m.parts.size(); ++t) items.push_back(Item{{Ref{i, t}}, m.parts[t].dev.block_threads, m.parts[t].dev.n_slices});
        }
    }
    std::stable_sort(items.begin(), items.end(), [&](const Item& a, const Item& b) {
        if (a.threads != b.threads) return a.threads > b.threads;
        return a.slices > b.slices;
    });
    const size_t first_slice_launch = plan.launches.size();
    for (const Item& it : items) for (const Ref& r : it.refs) refs.push_back(r);
    auto upload_table = [&](hispmv_ctx::BatchLaunch& l, const void* host, size_t bytes) -> int {
        HIP_TRY(c, hipMalloc(&l.d_table, bytes));
        HIP_TRY(c, hipMemcpy(l.d_table, host, bytes, hipMemcpyHostToDevice));
        return HISPMV_OK;
    };
    int rc;
    for (size_t k = 0; k < items.size();) {                     // slice kernels, one launch per class
        const int threads = items[k].threads;
        hispmv_ctx::BatchLaunch l;
        l.kind = 0;
        std::vector<MultiEntry> entries;
        while (k < items.size() && items[k].threads == threads && entries.size() + items[k].refs.size() <= (size_t)kMultiMax) {
            for (const Ref& r : items[k].refs) {
                SpmvDeviceMatrix& d = dev_of(r);
                MultiEntry e{};
                e.words = d.words; e.hdr = d.hdr; e.groups = d.groups; e.frags = d.frags;
                e.x = d_x[r.i]; e.bias = r.t == 0 ? bias[r.i] : nullptr; e.y = out_of(r); e.carry = d.carry;
                e.n_slices = d.n_slices; e.group_slices = d.group_slices; e.lds_floats = d.lds_floats; e.ytile_floats = d.ytile_floats;
                e.cols = d.cols; e.rows = d.rows;
                e.beta = r.t == 0 ? beta : 0.0f;
                entries.push_back(e);
                l.parts.push_back(&d);
            }
            l.item_tiles.push_back((uint8_t)items[k].refs.size());
            ++k;
        }
        plan.launches.push_back(std::move(l));
        if ((rc = upload_table(plan.launches.back(), entries.data(), entries.size() * sizeof(MultiEntry))) != HISPMV_OK) return rc;
    }
    // Launch order = stream assignment (main launch k goes to lane k mod lanes, hispmv_spmv_device_batch).  HISPMV_BATCH_ORDER=
    // small_first: the slice grids of SMALL workgroups first (256 threads, then 512, 1024), ahead of the tile streams: a
    // 256-thread workgroup (4 wavefronts, a few KiB of LDS) fits on a CU NEXT TO a 1024-thread slice workgroup (91 VGPRs: five
    // wavefronts per SIMD; windows of <= 115 KiB), so launched together the two grids share CUs -- the small matrices' L2
    // gathers ride under the HBM-bound stream of the large ones -- while a tile (145 KiB, 16 x 126 VGPRs) shares with nobody.
    if (c->batch_order == 1) {
        std::vector<hispmv_ctx::BatchLaunch> slices(std::make_move_iterator(plan.launches.begin() + (long)first_slice_launch),
                                                    std::make_move_iterator(plan.launches.end()));
        plan.launches.erase(plan.launches.begin() + (long)first_slice_launch, plan.launches.end());
        std::reverse(slices.begin(), slices.end());
        // small slice grids, then the large ones, then whatever was there before (dense, tile streams)
        std::vector<hispmv_ctx::BatchLaunch> rest = std::move(plan.launches);
        plan.launches.clear();
        for (auto& l : slices) plan.launches.push_back(std::move(l));
        for (auto& l : rest) plan.launches.push_back(std::move(l));
    }
    std::vector<Ref> fixrefs = refs;                            // + tile streams that cut long rows into pieces
    for (int i = 0; i < n; ++i) {
        const Matrix& m = *c->mats[idx[i]];
        if (!m.dense && m.format == 1)
            for (size_t t = 0; t < m.parts.size(); ++t) if (m.parts[t].tdev.n_fix > 0) fixrefs.push_back(Ref{i, t});
    }
    // Lanes of the main launches: round-robin in launch order, and the lane with the most (time-weighted) bytes is the CALLER'S
    // stream with its launches ENQUEUED FIRST (HISPMV_BATCH_LANES=rr: plain round-robin), so that the tail launch follows the
    // last-finishing chain on the same queue instead of behind a cross-queue dependency.  Replayed as a graph the runtime
    // keeps the chain of the FIRST captured root on the launch stream's queue: with the 1024-thread slice grid captured first
    // (the structured set, where it finishes last) the tail starts 6.7 us behind its end instead of 12 (kernel timelines in
    // profiles/r3_experiments/step_structure.json): 0.288-0.294 against 0.291-0.299 ms per step; in the pessimistic family
    // the tile streams are the heavier chain and the order stays what it was.  (Measured and dropped: longest-processing-time
    // assignment -- the 256-thread grid then follows the 1024-thread one and the pessimistic family loses 1-4 %; the heavy
    // lane on the caller's stream but captured second: 0.302-0.303.)
    {
        plan.lanes = plan.stream_bytes >= c->batch_streams_min_bytes ? c->batch_streams : 1;
        std::vector<hispmv_ctx::BatchLaunch*> mains;
        for (auto& l : plan.launches) {
            if (l.kind == 0) for (const SpmvDeviceMatrix* d : l.parts) l.weight += d->n_slices * (int64_t)kWideSliceBytes;
            // (a tile stream runs at ~2.5 TB/s against ~6.5 for a slice stream: its bytes count 2.5-fold; set when the entries were made)
            if (l.kind == 4) for (const GemvEntry& e : l.gemv) l.weight += 4 * (int64_t)e.rows * e.cols;
            mains.push_back(&l);
        }
        plan.lanes = std::max(1, std::min<int>(plan.lanes, (int)mains.size()));
        // round-robin in launch order (tile streams, 1024-thread slices, 256-thread slices, ...) ...
        int k = 0;
        for (hispmv_ctx::BatchLaunch* l : mains) l->lane = k++ % plan.lanes;
        if (c->batch_lanes_heavy_first && plan.lanes >= 2) {
            // ... and the lane with the most (time-weighted) bytes becomes the caller's stream, its launches enqueued first
            std::vector<int64_t> load((size_t)plan.lanes, 0);
            for (hispmv_ctx::BatchLaunch* l : mains) load[(size_t)l->lane] += l->weight;
            const int heavy = (int)(std::max_element(load.begin(), load.end()) - load.begin());
            for (hispmv_ctx::BatchLaunch* l : mains) l->lane = l->lane == heavy ? 0 : l->lane == 0 ? heavy : l->lane;
            std::stable_sort(plan.launches.begin(), plan.launches.end(), [](const hispmv_ctx::BatchLaunch& x, const hispmv_ctx::BatchLaunch& y) { return x.lane < y.lane; });
        }
    }
    // The tail: ONE launch that finishes the cut rows and merges the partial vectors of column-tiled matrices (which then
    // apply their own fix-ups) when every tiled matrix of the call carries its row -> fix table and the tables fit one
    // launch; otherwise a fix-up launch and a merge launch.
    std::vector<int> tiled;
    bool fused = !std::getenv("HISPMV_NO_FUSED_TAIL");
    for (int i = 0; i < n; ++i) {
        const Matrix& m = *c->mats[idx[i]];
        if (m.dense || m.parts.size() < 2) continue;
        tiled.push_back(i);
        fused = fused && m.d_fix_of_row != nullptr;
    }
    if (fused) {
        std::vector<Ref> plain;                                 // parts whose cut rows the fix-up blocks finish
        for (const Ref& r : fixrefs) if (c->mats[idx[r.i]]->parts.size() < 2) plain.push_back(r);
        fused = plain.size() <= (size_t)kMultiMax && tiled.size() <= (size_t)kMultiMax;
        if (fused) {
            hispmv_ctx::BatchLaunch l;
            l.kind = 5;
            std::vector<MultiFixEntry> fix;
            std::vector<TailMergeEntry> mrg;
            bool any = !tiled.empty();
            for (const Ref& r : plain) {
                SpmvDeviceMatrix& d = dev_of(r);
                fix.push_back(MultiFixEntry{d.fix_short, d.carry, out_of(r), d.n_fix_short, 0});
                l.fix_counts.push_back(d.n_fix_short);
                l.parts.push_back(&d); l.ys.push_back(out_of(r));               // (long chains: their own launches behind the tail)
                any = any || d.n_fix_short > 0 || d.n_fix_long > 0;
            }
            for (int i : tiled) {
                Matrix& m = *c->mats[idx[i]];
                TailMergeEntry e{};
                e.y = d_y[i]; e.parts = m.d_ypart; e.part_stride = (long long)kMaxBatch * m.rows; e.n_parts = (int32_t)m.parts.size() - 1; e.rows = m.rows;
                e.fix_of_row = m.d_fix_of_row;
                for (size_t t = 0; t < m.parts.size(); ++t) { e.fix[t] = m.parts[t].dev.fix_short; e.carry[t] = m.parts[t].dev.carry; }
                mrg.push_back(e);
                l.rows.push_back(m.rows);
            }
            if (!any) return HISPMV_OK;
            plan.launches.push_back(std::move(l));
            hispmv_ctx::BatchLaunch& L = plan.launches.back();
            if (!fix.empty() && (rc = upload_table(L, fix.data(), fix.size() * sizeof(MultiFixEntry))) != HISPMV_OK) return rc;
            if (!mrg.empty()) {
                HIP_TRY(c, hipMalloc(&L.d_table2, mrg.size() * sizeof(TailMergeEntry)));
                HIP_TRY(c, hipMemcpy(L.d_table2, mrg.data(), mrg.size() * sizeof(TailMergeEntry), hipMemcpyHostToDevice));
            }
            return HISPMV_OK;
        }
    }
    for (size_t k = 0; k < fixrefs.size(); k += kMultiMax) {    // fix-up of the cut rows
        hispmv_ctx::BatchLaunch l;
        l.kind = 1;
        std::vector<MultiFixEntry> fix;
        bool any = false;
        for (size_t q = k; q < std::min(fixrefs.size(), k + kMultiMax); ++q) {
            SpmvDeviceMatrix& d = dev_of(fixrefs[q]);
            fix.push_back(MultiFixEntry{d.fix_short, d.carry, out_of(fixrefs[q]), d.n_fix_short, 0});
            l.parts.push_back(&d);
            l.ys.push_back(out_of(fixrefs[q]));
            any = any || d.n_fix_short > 0 || d.n_fix_long > 0;
        }
        if (!any) continue;
        plan.launches.push_back(std::move(l));
        if ((rc = upload_table(plan.launches.back(), fix.data(), fix.size() * sizeof(MultiFixEntry))) != HISPMV_OK) return rc;
    }
    std::vector<MultiMergeEntry> merges;
    std::vector<int32_t> merge_rows;
    auto flush_merges = [&]() -> int {
        if (merges.empty()) return HISPMV_OK;
        hispmv_ctx::BatchLaunch l;
        l.kind = 2; l.rows = merge_rows;
        plan.launches.push_back(std::move(l));
        const int r2 = upload_table(plan.launches.back(), merges.data(), merges.size() * sizeof(MultiMergeEntry));
        merges.clear(); merge_rows.clear();
        return r2;
    };
    for (int i : tiled) {                                       // merge of the column-tile partial vectors
        Matrix& m = *c->mats[idx[i]];
        merges.push_back(MultiMergeEntry{d_y[i], m.d_ypart, (long long)kMaxBatch * m.rows, (int32_t)m.parts.size() - 1, m.rows});
        merge_rows.push_back(m.rows);
        if ((int)merges.size() == kMultiMax && (rc = flush_merges()) != HISPMV_OK) return rc;
    }
    return flush_merges();
}

HISPMV_API int hispmv_spmv_device_batch(hispmv_ctx* c, int32_t n, const int32_t* idx, const float* const* d_x,
                                        const float* const* d_bias, float* const* d_y, float alpha, float beta, void* stream) {
    if (!c) return HISPMV_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (stream) c->user_stream = (hipStream_t)stream;
    return spmv_batch_locked(c, n, idx, d_x, d_bias, d_y, alpha, beta, stream ? (hipStream_t)stream : c->stream);
}

namespace {
int spmv_batch_locked(hispmv_ctx* c, int32_t n, const int32_t* idx, const float* const* d_x, const float* const* d_bias,
                      float* const* d_y, float alpha, float beta, hipStream_t s) {
    if (n < 0 || (n > 0 && (!idx || !d_x || !d_y || (beta != 0.0f && !d_bias)))) return fail(c, HISPMV_EINVAL, "NULL argument");
    for (int i = 0; i < n; ++i) {
        if (idx[i] < 0 || idx[i] >= (int)c->mats.size()) return fail(c, HISPMV_EINVAL, "Matrix idx out of range");
        const Matrix& m = *c->mats[idx[i]];
        if (!m.loaded) return fail(c, HISPMV_ESTATE, "spmv_device_batch called before load_matrices");
        if (!d_x[i] || !d_y[i] || (beta != 0.0f && !d_bias[i])) return fail(c, HISPMV_EINVAL, "NULL device vector");
        for (int k = 0; k < i; ++k) {
            if (d_y[k] == d_y[i]) return fail(c, HISPMV_EINVAL, "two matrices of a batch write the same y");
            // the carry buffers of cut rows belong to the handle: one SpMV per handle at a time (hispmv.h, threading)
            if (idx[k] == idx[i] && !m.dense) return fail(c, HISPMV_EINVAL, "the same sparse handle twice in one batch");
        }
    }
    HIP_TRY(c, hipSetDevice(c->device));
    const float* const* bias = d_bias;
    std::vector<const float*> no_bias;
    if (!bias) { no_bias.assign((size_t)n, nullptr); bias = no_bias.data(); }
    // the launches of this call signature: built once, replayed afterwards (beta enters the tables; alpha is a kernel argument)
    std::vector<uint64_t> key{(uint64_t)n, (uint64_t)__builtin_bit_cast(uint32_t, beta)};
    for (int i = 0; i < n; ++i) {
        key.push_back((uint64_t)idx[i]); key.push_back((uint64_t)(uintptr_t)d_x[i]);
        key.push_back((uint64_t)(uintptr_t)(beta != 0.0f ? bias[i] : nullptr)); key.push_back((uint64_t)(uintptr_t)d_y[i]);
    }
    hispmv_ctx::BatchPlan* plan = nullptr;
    for (auto& p : c->batch_plans) if (p.key == key) { plan = &p; break; }
    if (!plan) {
        if (c->batch_plans.size() >= 16) free_batch_plans(c);      // callers that keep changing their vectors: start over
        c->batch_plans.emplace_back();
        c->batch_plans.back().key = key;
        for (int i = 0; i < n; ++i) {
            const Matrix& mi = *c->mats[idx[i]];
            c->batch_plans.back().stream_bytes += mi.dense ? 4 * (int64_t)mi.rows * mi.cols : 8 * mi.nnz;
        }
        const int rc = build_batch_plan(c, c->batch_plans.back(), n, idx, d_x, bias, d_y, beta);
        if (rc != HISPMV_OK) {                                      // nothing half-built stays behind
            for (auto& l : c->batch_plans.back().launches) { dev_free(l.d_table); dev_free(l.d_table2); }
            c->batch_plans.pop_back();
            return rc;
        }
        plan = &c->batch_plans.back();
    }
    // main launches (kinds 0 and 3) are independent of each other: spread over the caller's stream and the side streams;
    // the fix-up and merge launches follow on the caller's stream behind a join
    const int lanes = plan->lanes;
    // the launches, as one function of the stream: main launches spread over the caller's stream and the side streams (forked
    // from / joined to it with events), fix-up and merge behind the join
    auto enqueue = [&]() -> int {
        if (lanes > 1) {
            HIP_TRY(c, hipEventRecord(c->ev_fork, s));
            for (int i = 0; i + 1 < lanes; ++i) HIP_TRY(c, hipStreamWaitEvent(c->side[i], c->ev_fork, 0));
        }
        bool joined = lanes <= 1;
        for (const auto& l : plan->launches) {
            hipError_t e = hipSuccess;
            const bool is_main = l.kind == 0 || l.kind == 3 || l.kind == 4;
            hipStream_t ls = s;
            if (is_main && lanes > 1) ls = l.lane == 0 ? s : c->side[l.lane - 1];
            if (!is_main && !joined) {
                for (int i = 0; i + 1 < lanes; ++i) { HIP_TRY(c, hipEventRecord(c->ev_join[i], c->side[i])); HIP_TRY(c, hipStreamWaitEvent(s, c->ev_join[i], 0)); }
                joined = true;
            }
            if (l.kind == 0) e = launch_spmv_multi(l.parts.data(), (int)l.parts.size(), l.item_tiles.data(), (int)l.item_tiles.size(), (const MultiEntry*)l.d_table, alpha, ls);
            else if (l.kind == 3) e = launch_tts_multi(l.tts.data(), (int)l.tts.size(), l.item_tiles.data(), (int)l.item_tiles.size(), (const TtsEntry*)l.d_table, alpha, ls);
            else if (l.kind == 4) e = launch_gemv_multi(l.gemv.data(), (int)l.gemv.size(), (const GemvEntry*)l.d_table, alpha, ls);
            else if (l.kind == 1) e = launch_fixup_multi(l.parts.data(), l.ys.data(), (int)l.parts.size(), (const MultiFixEntry*)l.d_table, alpha, ls);
            else if (l.kind == 5) {
                e = launch_tail_multi(l.fix_counts.data(), (int)l.fix_counts.size(), (const MultiFixEntry*)l.d_table, l.rows.data(), (int)l.rows.size(),
                                      (const TailMergeEntry*)l.d_table2, alpha, ls);
                for (size_t q = 0; e == hipSuccess && q < l.parts.size(); ++q)       // rows that span more than 32 slices: a wavefront per row
                    if (l.parts[q]->n_fix_long > 0) e = launch_fixup_long(*l.parts[q], l.ys[q], alpha, ls);
            }
            else e = launch_merge_multi(l.rows.data(), (int)l.rows.size(), (const MultiMergeEntry*)l.d_table, ls);
            if (e != hipSuccess) return hip_fail(c, e, l.kind == 0 ? "launch_spmv_multi" : l.kind == 1 ? "launch_fixup_multi" : l.kind == 3 ? "launch_tts_multi" : l.kind == 4 ? "launch_gemv_multi" : l.kind == 5 ? "launch_tail_multi" : "launch_merge_multi");
        }
        if (!joined)
            for (int i = 0; i + 1 < lanes; ++i) { HIP_TRY(c, hipEventRecord(c->ev_join[i], c->side[i])); HIP_TRY(c, hipStreamWaitEvent(s, c->ev_join[i], 0)); }
        return HISPMV_OK;
    };
    // A two-stream call is captured into a HIP graph the second time its signature is seen (the first run sets the
    // kernels' attributes) and replayed from then on: the set's step 0.315-0.320 -> 0.309-0.310 ms -- the fork/join events
    // of the two streams become graph edges.  HISPMV_BATCH_GRAPH=0 switches it off; a stream that is being captured by the
    // caller, or a capture the runtime refuses, falls back to plain launches.
    // (a stream the CALLER is capturing takes plain launches: they become nodes of the caller's graph; replaying the library's
    // own graph into a capture recorded nothing on this runtime)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool caller_captures = hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone;
    (void)hipGetLastError();
    if (c->batch_graphs && lanes > 1 && !caller_captures) {      // (one-stream calls: a graph launch costs more than their 2-4 plain launches, the model layers 50 -> 54 us)
        using Slot = hispmv_ctx::BatchPlan::GraphSlot;
        auto drop_graphs = [&]() {
            for (Slot& g : plan->slot) {
                if (g.launched && g.done) (void)hipEventSynchronize(g.done);
                if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
                g.launched = false;
            }
            if (plan->graph_src) { (void)hipGraphDestroy(plan->graph_src); plan->graph_src = nullptr; }
        };
        auto launch_slot = [&](Slot& g) -> bool {
            if (hipGraphLaunch(g.exec, s) != hipSuccess) { (void)hipGetLastError(); return false; }
            if (!g.done && hipEventCreateWithFlags(&g.done, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); g.done = nullptr; }
            // (without the event the executable can never be patched safely: its alpha stays what it is, see below)
            g.launched = g.done && hipEventRecord(g.done, s) == hipSuccess;
            if (!g.launched) (void)hipGetLastError();
            g.last_use = ++plan->use_counter;
            return true;
        };
        if (plan->graph_src) {
            // 1. an executable that already carries this alpha
            Slot* pick = nullptr;
            for (Slot& g : plan->slot) if (g.exec && g.alpha == alpha) { pick = &g; break; }
            if (!pick) {
                // 2. another alpha on the same call: no capture -- a second executable of the captured graph the first time,
                //    afterwards the executable used longest ago, patched once ITS last launch has completed
                Slot* victim = nullptr;
                for (Slot& g : plan->slot) if (!g.exec) { victim = &g; break; }
                if (victim) {
                    if (hipGraphInstantiate(&victim->exec, plan->graph_src, nullptr, nullptr, 0) == hipSuccess) { c->graph_instantiations++; victim->launched = false; }
                    else { (void)hipGetLastError(); victim->exec = nullptr; victim = nullptr; }
                }
                if (!victim) victim = plan->slot[0].last_use <= plan->slot[1].last_use ? &plan->slot[0] : &plan->slot[1];
                bool ok = victim->exec != nullptr;
                if (ok && victim->launched) ok = victim->done && hipEventSynchronize(victim->done) == hipSuccess;
                if (ok) ok = graph_set_alpha(victim->exec, plan->graph_src, alpha) == hipSuccess;
                if (ok) { victim->alpha = alpha; c->graph_alpha_updates++; pick = victim; }
                else (void)hipGetLastError();
            }
            if (pick && launch_slot(*pick)) return HISPMV_OK;
            drop_graphs();
        }
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (plan->runs >= 1 && hipStreamIsCapturing(s, &st) == hipSuccess && st == hipStreamCaptureStatusNone &&
            hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            const int rc = enqueue();
            hipGraph_t g = nullptr;
            const hipError_t e_end = hipStreamEndCapture(s, &g);
            if (rc != HISPMV_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
            drop_graphs();
            hipError_t e = e_end;
            Slot& g0 = plan->slot[0];
            if (e == hipSuccess) e = hipGraphInstantiate(&g0.exec, g, nullptr, nullptr, 0);
            if (e == hipSuccess) c->graph_instantiations++;
            plan->graph_src = g;
            if (e == hipSuccess) { g0.alpha = alpha; g0.launched = false; if (launch_slot(g0)) return HISPMV_OK; }
            (void)hipGetLastError();
            drop_graphs();
            c->batch_graphs = false;               // this runtime / stream does not take it: plain launches from here on
        } else {
            (void)hipGetLastError();
        }
    }
    plan->runs++;
    return enqueue();
}
}  // namespace

HISPMV_API int hispmv_synchronize(hispmv_ctx* c) {
    if (!c) return HISPMV_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->user_stream) {
        const hipError_t e = hipStreamSynchronize(c->user_stream);
        if (e == hipErrorInvalidHandle || e == hipErrorInvalidResourceHandle || e == hipErrorContextIsDestroyed) {
            (void)hipGetLastError();          // the caller has destroyed that stream since: nothing left to wait for
            c->user_stream = nullptr;
        } else if (e != hipSuccess) return hip_fail(c, e, "hipStreamSynchronize(caller stream)");
    }
    return check_device_error(c);
}

HISPMV_API float hispmv_last_kernel_ms(hispmv_ctx* c) { return c ? c->last_ms : -1.0f; }

HISPMV_API int hispmv_batch_graph_stats(hispmv_ctx* c, int64_t out[2]) {
    if (!c || !out) return HISPMV_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    out[0] = c->graph_instantiations; out[1] = c->graph_alpha_updates;
    return HISPMV_OK;
}

HISPMV_API float hispmv_time_device(hispmv_ctx* c, int idx, const float* d_x, const float* d_bias, float* d_y,
                                    float alpha, float beta, int reps) {
    if (!c || reps <= 0) return -1.0f;
    std::lock_guard<std::mutex> g(c->mu);
    if (idx < 0 || idx >= (int)c->mats.size() || !c->mats[idx]->loaded) { c->err = "bad matrix for time_device"; return -1.0f; }
    Matrix& m = *c->mats[idx];
    if (hipSetDevice(c->device) != hipSuccess) return -1.0f;
    if (hipEventRecord(c->ev0, c->stream) != hipSuccess) return -1.0f;
    for (int i = 0; i < reps; ++i)
        if (launch_matrix(c, m, d_x, d_bias, d_y, alpha, beta, c->stream) != HISPMV_OK) return -1.0f;
    if (hipEventRecord(c->ev1, c->stream) != hipSuccess) return -1.0f;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess) return -1.0f;
    return ms / reps;
}

HISPMV_API int hispmv_num_matrices(const hispmv_ctx* c) { return c ? (int)c->mats.size() : 0; }

HISPMV_API int hispmv_get_matrix_info(const hispmv_ctx* c, int idx, hispmv_matrix_info* out) {
    if (!c || !out || idx < 0 || idx >= (int)c->mats.size()) return HISPMV_EINVAL;
    const Matrix& m = *c->mats[idx];
    out->rows = m.rows; out->cols = m.cols; out->nnz = m.nnz; out->is_dense = m.dense; out->loaded = m.loaded;
    out->n_slices = m.n_slices; out->n_elems = m.n_elems; out->n_split_rows = m.n_split;
    out->device_bytes = m.device_bytes; out->prep_seconds = m.prep_seconds;
    out->block_threads = m.plan_threads; out->group_slices = m.plan_group; out->lds_bytes = m.plan_lds * 4;
    out->col_tiles = (int32_t)m.parts.size();
    out->carry_lookback = (!m.dense && !m.parts.empty() && m.parts[0].dev.lookback) ? 1 : 0; out->col_tile_width = m.col_tile_width; out->col_tile_base = m.col_tile_base; out->compact_slices = (int32_t)std::min<int64_t>(m.compact_slices, INT32_MAX);
    out->format = m.format; out->tts_lines_per_gather = (float)m.tts_lines_per_gather;
    out->tile_kind = m.parts.size() > 1 ? (m.tile_kind ? m.tile_kind : 1) : 0;
    return HISPMV_OK;
}

// ---- host-only preprocessor access ---------------------------------------------------------------
HISPMV_API const char* hispmv_prep_last_error(void) { return g_prep_err.c_str(); }

HISPMV_API int hispmv_host_threads(void) { return host_threads(); }

HISPMV_API int hispmv_prep_from_coo(hispmv_prep** out, const int32_t* r, const int32_t* cl, const float* v,
                                    int64_t nnz, int32_t rows, int32_t cols) {
    if (!out) return HISPMV_EINVAL;
    host_threads();
    *out = nullptr;
    if (rows <= 0 || cols <= 0 || nnz < 0) { g_prep_err = "bad sparse matrix arguments"; return HISPMV_EINVAL; }
    try {
        auto p = std::make_unique<hispmv_prep>();
        p->csr = coo_to_csr(rows, cols, nnz, r, cl, v);
        p->st = build_stream(p->csr);
        *out = p.release();
        return HISPMV_OK;
    } catch (const std::exception& ex) { g_prep_err = ex.what(); return HISPMV_EINVAL; }
}

HISPMV_API int hispmv_prep_from_coo_device(hispmv_prep** out, int device_id, const int32_t* r, const int32_t* cl, const float* v,
                                           int64_t nnz, int32_t rows, int32_t cols, double seconds[5]) {
    if (!out) return HISPMV_EINVAL;
    *out = nullptr;
    if (rows <= 0 || cols <= 0 || nnz < 0 || (nnz > 0 && (!r || !cl || !v))) { g_prep_err = "bad sparse matrix arguments"; return HISPMV_EINVAL; }
    if (hipSetDevice(device_id) != hipSuccess) { g_prep_err = "no such HIP device"; return HISPMV_EDEVICE; }
    try {
        auto p = std::make_unique<hispmv_prep>();
        DevicePrepTimes t;
        std::string err;
        if (!prep_on_device(rows, cols, nnz, r, cl, v, p->csr, p->st, t, err)) {
            g_prep_err = err;
            return err.find("outside") != std::string::npos || err.find("dimension") != std::string::npos ? HISPMV_EINVAL : HISPMV_EDEVICE;
        }
        if (seconds) { seconds[0] = t.upload; seconds[1] = t.csr_device; seconds[2] = t.offsets_host; seconds[3] = t.stream_device; seconds[4] = t.download; }
        *out = p.release();
        return HISPMV_OK;
    } catch (const std::exception& ex) { g_prep_err = ex.what(); return HISPMV_EINVAL; }
}

HISPMV_API int hispmv_prep_from_mtx(hispmv_prep** out, const char* path, int flavor) {
    host_threads();
    if (!out) return HISPMV_EINVAL;
    *out = nullptr;
    if (!path || (flavor != 0 && flavor != 1)) { g_prep_err = "bad arguments"; return HISPMV_EINVAL; }
    try {
        Coo coo = read_mtx(path, (MtxFlavor)flavor);
        auto p = std::make_unique<hispmv_prep>();
        p->csr = coo_to_csr(coo.rows, coo.cols, (int64_t)coo.r.size(), coo.r.data(), coo.c.data(), coo.v.data());
        p->st = build_stream(p->csr);
        *out = p.release();
        return HISPMV_OK;
    } catch (const std::runtime_error& ex) { g_prep_err = ex.what(); return HISPMV_EIO;
    } catch (const std::exception& ex) { g_prep_err = ex.what(); return HISPMV_EINVAL; }
}

HISPMV_API void hispmv_prep_free(hispmv_prep* p) { delete p; }

// The format / tiling decision of the loader for this matrix on a device with n_cus compute units, host-only
// (hispmv_choose.cpp: the same function hispmv_create_sparse_handle* calls).  Works on a copy of the prepared CSR.
HISPMV_API int hispmv_prep_choose_format(const hispmv_prep* p, int n_cus, int64_t out[16]) {
    if (!p || !out || n_cus <= 0) return HISPMV_EINVAL;
    try {
        Csr copy = p->csr;
        FormatOptions opt = FormatOptions::from_env();
        opt.decide_only = true;
        const FormatChoice ch = choose_format(std::move(copy), nullptr, n_cus, opt);
        int64_t n_slices = 0, n_elems = 0, n_split = 0, global_elems = 0;
        for (const HostPart& q : ch.parts) {
            if (q.is_tts) { n_slices += (int64_t)q.tts.col_base.size(); n_elems += q.tts.nnz + q.tts.n_fillers; n_split += (int64_t)q.tts.fix.size() / 4; }
            else { n_slices += q.st.n_slices; n_elems += q.st.n_elems; n_split += (int64_t)q.st.fix.size(); global_elems += q.plan.lds_floats > 0 ? q.plan.global_elems : q.st.n_slices * (int64_t)kSliceElems; }
        }
        const HostPart& p0 = ch.parts[0];
        out[0] = ch.format; out[1] = ch.parts.size() > 1 ? (ch.tile_kind ? ch.tile_kind : 1) : 0; out[2] = (int64_t)ch.parts.size();
        out[3] = ch.col_tile_width; out[4] = ch.col_tile_base; out[5] = ch.l2_tiles ? 1 : 0;
        out[6] = ch.format == 1 ? p0.tts.geometry.threads : p0.plan.block_threads;
        out[7] = ch.format == 1 ? p0.tts.geometry.max_slots / kTtsChunk : p0.plan.group_slices;
        out[8] = ch.format == 1 ? 0 : p0.plan.lds_floats;
        out[9] = n_slices; out[10] = n_elems; out[11] = n_split;
        out[12] = (int64_t)(ch.tts_lines_per_gather * 1000.0 + 0.5);
        out[13] = global_elems;        // elements that gather x through L2 (slice streams: outside their window, or no window at all)
        out[14] = 0; out[15] = 0;
        return HISPMV_OK;
    } catch (const std::exception& ex) { g_prep_err = ex.what(); return HISPMV_EINVAL; }
}

HISPMV_API int hispmv_prep_dims(const hispmv_prep* p, int64_t d[8]) {
    if (!p || !d) return HISPMV_EINVAL;
    d[0] = p->st.rows; d[1] = p->st.cols; d[2] = p->st.nnz; d[3] = p->st.n_elems; d[4] = p->st.n_slices;
    d[5] = kSliceElems; d[6] = (int64_t)p->st.fix.size(); d[7] = p->st.bytes();
    return HISPMV_OK;
}
HISPMV_API int hispmv_prep_plan(const hispmv_prep* p, int n_cus, int64_t plan[6]) {
    if (!p || !plan || n_cus <= 0) return HISPMV_EINVAL;
    SliceStream copy = p->st;                      // make_plan rewrites the words of staged groups
    const LaunchPlan g = make_plan(copy, n_cus);
    plan[0] = g.block_threads; plan[1] = g.group_slices; plan[2] = g.lds_floats; plan[3] = g.ytile_floats;
    plan[4] = (int64_t)g.groups.size();
    plan[5] = ((int64_t)g.lds_floats + (int64_t)g.ytile_floats * (g.block_threads / 64)) * 4;
    return HISPMV_OK;
}

HISPMV_API int hispmv_prep_apply_plan(hispmv_prep* p, int n_cus, int64_t counts[2]) {
    if (!p || !counts || n_cus <= 0) return HISPMV_EINVAL;
    p->plan = make_plan(p->st, n_cus);
    counts[0] = (int64_t)p->plan.groups.size(); counts[1] = (int64_t)p->plan.frags.size();
    return HISPMV_OK;
}
HISPMV_API const int32_t* hispmv_prep_groups(const hispmv_prep* p) { return (const int32_t*)p->plan.groups.data(); }
HISPMV_API const int32_t* hispmv_prep_frags(const hispmv_prep* p) { return (const int32_t*)p->plan.frags.data(); }

HISPMV_API int hispmv_prep_build_tts(hispmv_prep* p, int64_t target_tile_elems, int small_geometry, int64_t counts[8], double* lines_per_gather) {
    if (!p || !counts) return HISPMV_EINVAL;
    try {
        TtsGeometry geo;
        if (small_geometry == 1) { geo.max_slots = kTtsSmallSlots; geo.max_rows = kTtsSmallRows; geo.tiles_wanted = 512; }
        if (small_geometry == 6) geo.zero_fill = true;      // the standard sizes without filler words (HISPMV_TTS_GEOMETRY=zerofill)
        if (small_geometry >= 2 && small_geometry < 6) {         // 2 + q / 4 + q: column part q of the tall / paired geometry, as the loader builds it for a 256-CU device
            const bool paired = small_geometry >= 4;
            const int q = small_geometry - (paired ? 4 : 2);
            if (q >= kTtsTallParts) { g_prep_err = "no such column part"; return HISPMV_EINVAL; }
            const std::vector<int32_t> cuts = tts_column_cuts(p->csr, kTtsTallParts);
            const Csr part = csr_column_range(p->csr, q == 0 ? 0 : cuts[(size_t)q - 1], q + 1 == kTtsTallParts ? p->csr.cols : cuts[(size_t)q]);
            p->tts = build_tts(part, target_tile_elems, paired ? tts_paired_geometry(256) : tts_tall_geometry(256, kTtsTallParts));
        } else
        p->tts = build_tts(p->csr, target_tile_elems, geo);
    } catch (const std::exception& ex) { g_prep_err = ex.what(); return HISPMV_EINVAL; }
    const TtsStream& t = p->tts;
    counts[0] = (int64_t)t.tiles.size(); counts[1] = (int64_t)t.blocks.size(); counts[2] = (int64_t)t.col_base.size();
    counts[3] = (int64_t)t.chunk_info.size() / 2; counts[4] = t.n_fillers; counts[5] = t.n_pad_words; counts[6] = t.max_rows; counts[7] = t.max_slots;
    if (lines_per_gather) *lines_per_gather = t.lines_per_gather;
    return HISPMV_OK;
}
HISPMV_API int hispmv_prep_tts_pieces(const hispmv_prep* p, int64_t counts[2]) {
    if (!p || !counts) return HISPMV_EINVAL;
    counts[0] = (int64_t)p->tts.fix.size() / 4; counts[1] = p->tts.n_carry;
    return HISPMV_OK;
}
HISPMV_API const void* hispmv_prep_tts_array(const hispmv_prep* p, int which) {
    if (!p) return nullptr;
    switch (which) {
        case 6: return p->tts.fix.data();
        case 0: return p->tts.words.data();
        case 1: return p->tts.col_base.data();
        case 2: return p->tts.flags.data();
        case 3: return p->tts.chunk_info.data();
        case 4: return p->tts.tiles.data();
        case 5: return p->tts.blocks.data();
        default: return nullptr;
    }
}

HISPMV_API const int64_t* hispmv_prep_csr_row_ptr(const hispmv_prep* p) { return p->csr.row_ptr.data(); }
HISPMV_API const int32_t* hispmv_prep_csr_col(const hispmv_prep* p) { return p->csr.col.data(); }
HISPMV_API const float* hispmv_prep_csr_val(const hispmv_prep* p) { return p->csr.val.data(); }
HISPMV_API const uint64_t* hispmv_prep_words(const hispmv_prep* p) { return p->st.words.data(); }
HISPMV_API const int32_t* hispmv_prep_slice_hdr(const hispmv_prep* p) { return (const int32_t*)p->st.hdr.data(); }
HISPMV_API const int32_t* hispmv_prep_fix(const hispmv_prep* p) { return (const int32_t*)p->st.fix.data(); }
