// hispmv_abi.cpp -- the C ABI of libhispmv.so (include/hispmv.h): context, matrix handles,
// arena accounting, HBM upload and launches.  MI355X counterpart of the reference's
// FpgaHandle (pyhispmv/src/fpga_handle.cpp:40-388), which owns the XRT device, the
// per-channel matrix arena and the kernel run object.  Host-side HIP runtime calls only;
// the kernels live in hispmv_kernels.hip, the preprocessor in hispmv_prep.cpp.
#include "hispmv_ctx.h"

using namespace hispmv;

namespace {

thread_local std::string g_create_err;
thread_local std::string g_prep_err;

}  // namespace

namespace hispmv {
int fail(hispmv_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg; else g_create_err = msg;
    return code;
}
int hip_fail(hispmv_ctx* c, hipError_t e, const char* what) {
    return fail(c, HISPMV_EDEVICE, std::string(what) + ": " + hipGetErrorString(e));
}
std::string& prep_error() { return g_prep_err; }
std::atomic<int64_t> g_free_failures{0};
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
}  // namespace hispmv

namespace {

int64_t sparse_device_bytes(const SliceStream& st, const DeviceStream& ds) {
    return ds.n_bytes + (int64_t)st.hdr.size() * 16 + (int64_t)st.fix.size() * 16 +
           (int64_t)st.n_slices * 12 + 8;
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
            m->device_bytes += sparse_device_bytes(p.st, p.dstream) + (int64_t)p.dstream.groups.size() * 4 + (int64_t)p.plan.frags.size() * 16 + (p.dstream.any_stray ? p.st.n_slices * (int64_t)kStraySlots * 4 : 0);
            m->compact_slices += p.dstream.compact_slices;
            if (p.has_batch_layout) m->device_bytes += p.batch_dstream.n_bytes + (int64_t)p.st.hdr.size() * 16 + (int64_t)p.batch_dstream.groups.size() * 4 + (int64_t)p.batch_plan.frags.size() * 16 +
                                                       (p.batch_dstream.any_stray ? p.st.n_slices * (int64_t)kStraySlots * 4 : 0);
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
    if (const char* env = std::getenv("HISPMV_STEP_KERNEL")) c->step_kernel = std::atoi(env) != 0;
    if (const char* env = std::getenv("HISPMV_STEP_ORDER")) c->step_order = !std::strcmp(env, "lpt") ? 1 : !std::strcmp(env, "grid") ? 2 : !std::strcmp(env, "alt2") ? 3 : !std::strcmp(env, "alt3") ? 4 : std::atoi(env) >= 16 ? std::atoi(env) : 0;
    if (const char* env = std::getenv("HISPMV_BATCH_ORDER")) c->batch_order = !std::strcmp(env, "small_first") ? 1 : 0;
    if (const char* env = std::getenv("HISPMV_BATCH_LANES")) c->batch_lanes_heavy_first = std::strcmp(env, "rr") != 0;
    if (const char* env = std::getenv("HISPMV_BATCH_STREAMS")) { c->batch_streams = std::max(1, std::min(3, std::atoi(env))); c->batch_streams_min_bytes = 0; }
    // Experiment (HISPMV_CU_SPLIT=<hexA>:<hexB>, round 4): the two side streams get CU masks (the 32-bit pattern repeated over the
    // device's CUs) and a batch call sends its slice / dense grids to side stream 0 and its tile-stream grids to side stream 1 --
    // an HBM-bound grid does not need every CU to saturate the memory, a cache-bound one wants as many as it can get.
    uint32_t cu_pat[2] = {0, 0};
    if (const char* env = std::getenv("HISPMV_CU_SPLIT")) {
        char* end = nullptr;
        cu_pat[0] = (uint32_t)std::strtoul(env, &end, 16);
        if (end && *end == ':') cu_pat[1] = (uint32_t)std::strtoul(end + 1, nullptr, 16);
        c->cu_split = cu_pat[0] != 0 && cu_pat[1] != 0;
    }
    for (int i = 0; i < 2; ++i) {
        if (c->cu_split) {
            std::vector<uint32_t> mask((size_t)(prop.multiProcessorCount + 31) / 32, cu_pat[i]);
            if ((e = hipExtStreamCreateWithCUMask(&c->side[i], (uint32_t)mask.size(), mask.data())) != hipSuccess) return give_up(e, "hipExtStreamCreateWithCUMask");
        } else
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
        // (scratch of the device layout -- the uploaded host words of the parts --, freed once the stream has drained, also on an error return)
        struct Scratch { std::vector<void*> v; void push_back(void* p) { v.push_back(p); } ~Scratch() { for (void* p : v) (void)hipFree(p); } } layout_scratch;
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
            const uint16_t* dfh = nullptr;
            std::vector<uint32_t> planes;          // gap-coded row ends: both planes of the codes in one word per lane and chunk
            if (!ts.flags_hi.empty()) {
                planes.resize(ts.flags.size());
                for (size_t q = 0; q < planes.size(); ++q) planes[q] = (uint32_t)ts.flags[q] | ((uint32_t)ts.flags_hi[q] << 16);
                const uint32_t* dp = nullptr;
                if ((rc = upload(c, m, planes.data(), planes.size(), &dp)) != HISPMV_OK) return rc;
                HIP_TRY(c, hipStreamSynchronize(c->stream));      // (`planes` is a local)
                dfh = (const uint16_t*)dp;
            }
            if ((rc = upload(c, m, ts.chunk_info.data(), ts.chunk_info.size(), &dci)) != HISPMV_OK) return rc;
            if ((rc = upload(c, m, ts.tiles.data(), ts.tiles.size(), &dt)) != HISPMV_OK) return rc;
            if ((rc = upload(c, m, ts.blocks.data(), ts.blocks.size(), &db)) != HISPMV_OK) return rc;
            TtsDeviceMatrix& d = p.tdev;
            d.words = dw; d.col_base = dcb; d.flags = dfl; d.flags_hi = dfh; d.chunk_info = (const int2*)dci; d.tiles = (const int4*)dt; d.blocks = (const int4*)db;
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
            d.zero_fill = ts.geometry.zero_fill ? 1 : ts.geometry.gap_rows ? 2 : 0;
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
                // the slices in their device layout: packed on the host (HISPMV_LAYOUT=host) or laid out HERE from the uploaded host words
                const bool lay_out = p.dstream.bytes.empty() && p.dstream.n_bytes > 0;
                uint64_t* d_host_words = nullptr;
                if (!lay_out) {
                    if ((rc = upload(c, m, p.dstream.bytes.data(), p.dstream.bytes.size(), &dw)) != HISPMV_OK) return rc;
                } else {
                    if ((int64_t)p.st.words.size() != ns * kSliceElems) return fail(c, HISPMV_EINVAL, "internal: host words missing for the device layout");
                    void* blk = nullptr;
                    HIP_TRY(c, hipMalloc(&blk, (size_t)p.dstream.n_bytes));
                    m.allocs.push_back(blk);
                    dw = (const uint8_t*)blk;
                    HIP_TRY(c, hipMalloc((void**)&d_host_words, p.st.words.size() * sizeof(uint64_t)));
                    layout_scratch.push_back(d_host_words);
                    HIP_TRY(c, hipMemcpyAsync(d_host_words, p.st.words.data(), p.st.words.size() * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
                }
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
                uint32_t* d_stray_cols = nullptr;
                if (!p.dstream.any_stray) {
                    if ((rc = upload(c, m, hh.data(), hh.size(), &dh)) != HISPMV_OK) return rc;
                } else if (lay_out) {
                    void* blk = nullptr;
                    const size_t hb = hh.size() * sizeof(SliceHdr), sb = (size_t)ns * kStraySlots * sizeof(uint32_t);
                    HIP_TRY(c, hipMalloc(&blk, hb + sb));
                    m.allocs.push_back(blk);
                    HIP_TRY(c, hipMemcpyAsync(blk, hh.data(), hb, hipMemcpyHostToDevice, c->stream));
                    HIP_TRY(c, hipMemsetAsync((char*)blk + hb, 0xff, sb, c->stream));        // 0xffffffff = no stray
                    dh = (const SliceHdr*)blk;
                    d_stray_cols = (uint32_t*)((char*)blk + hb);
                } else {
                    // stray slots: the columns of every slice's strays (64 x u32 per slice) live BEHIND the headers in one
                    // allocation -- the kernels reach them as hdr + n_slices, no further pointer to carry around
                    void* blk = nullptr;
                    const size_t hb = hh.size() * sizeof(SliceHdr), sb = p.dstream.stray_cols.size() * sizeof(uint32_t);
                    HIP_TRY(c, hipMalloc(&blk, hb + sb));
                    m.allocs.push_back(blk);
                    HIP_TRY(c, hipMemcpyAsync(blk, hh.data(), hb, hipMemcpyHostToDevice, c->stream));
                    HIP_TRY(c, hipMemcpyAsync((char*)blk + hb, p.dstream.stray_cols.data(), sb, hipMemcpyHostToDevice, c->stream));
                    dh = (const SliceHdr*)blk;
                }
                if (lay_out) {
                    const int e = layout_on_device(d_host_words, ns, p.plan.group_slices, dg, p.plan.lds_floats, p.plan.block_threads / 64,
                                                   (uint8_t*)dw, d_stray_cols, c->stream);
                    if (e != 0) return hip_fail(c, (hipError_t)e, "layout_on_device");
                }
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
                d.group_slices = p.plan.group_slices; d.block_threads = p.plan.block_threads;
                d.lds_floats = p.plan.lds_floats + p.dstream.stray_floats;        // the x window + the wavefronts' stray areas behind it
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
                // stray slots: the packer placed every slice's strays by the slice's position in the ROTATED walk of the fix-up
                // variant; the look-back variant walks its groups in slice order
                d.has_strays = p.dstream.stray_floats > 0;
                if (d.has_strays) { d.lookback = false; d.use_ticket = false; }
                // the batch layout (hispmv_choose.h): its own group table, fragments, slice bytes and headers (the spill flags differ);
                // rows, carries, fix lists and the error word are the part's
                if (p.has_batch_layout) {
                    SpmvDeviceMatrix b = d;
                    const int32_t* bg = nullptr; const Frag* bf = nullptr; const uint8_t* bw = nullptr; const SliceHdr* bh = nullptr;
                    if ((rc = upload(c, m, p.batch_dstream.groups.data(), p.batch_dstream.groups.size(), &bg)) != HISPMV_OK) return rc;
                    if ((rc = upload(c, m, p.batch_plan.frags.data(), p.batch_plan.frags.size(), &bf)) != HISPMV_OK) return rc;
                    std::vector<SliceHdr> bhh = hh;
                    for (int64_t sl = 0; sl < ns; ++sl) bhh[(size_t)sl].x_span = (!p.batch_plan.slice_spills.empty() && p.batch_plan.slice_spills[(size_t)sl]) ? 1 : 0;
                    uint32_t* b_stray = nullptr;
                    {
                        void* blk = nullptr;
                        const size_t hb = bhh.size() * sizeof(SliceHdr), sb = p.batch_dstream.any_stray ? (size_t)ns * kStraySlots * sizeof(uint32_t) : 0;
                        HIP_TRY(c, hipMalloc(&blk, hb + sb));
                        m.allocs.push_back(blk);
                        HIP_TRY(c, hipMemcpy(blk, bhh.data(), hb, hipMemcpyHostToDevice));            // (`bhh` is a local: synchronous)
                        if (sb && !p.batch_dstream.stray_cols.empty()) HIP_TRY(c, hipMemcpy((char*)blk + hb, p.batch_dstream.stray_cols.data(), sb, hipMemcpyHostToDevice));
                        else if (sb) HIP_TRY(c, hipMemsetAsync((char*)blk + hb, 0xff, sb, c->stream));
                        bh = (const SliceHdr*)blk;
                        b_stray = sb ? (uint32_t*)((char*)blk + hb) : nullptr;
                    }
                    if (!p.batch_dstream.bytes.empty()) {
                        if ((rc = upload(c, m, p.batch_dstream.bytes.data(), p.batch_dstream.bytes.size(), &bw)) != HISPMV_OK) return rc;
                    } else {
                        void* blk = nullptr; uint64_t* tmp = nullptr;
                        HIP_TRY(c, hipMalloc(&blk, (size_t)p.batch_dstream.n_bytes));
                        m.allocs.push_back(blk);
                        bw = (const uint8_t*)blk;
                        HIP_TRY(c, hipMalloc((void**)&tmp, p.batch_words.size() * sizeof(uint64_t)));
                        layout_scratch.push_back(tmp);
                        HIP_TRY(c, hipMemcpyAsync(tmp, p.batch_words.data(), p.batch_words.size() * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
                        const int e = layout_on_device(tmp, ns, p.batch_plan.group_slices, bg, p.batch_plan.lds_floats, p.batch_plan.block_threads / 64, (uint8_t*)blk, b_stray, c->stream);
                        if (e != 0) return hip_fail(c, (hipError_t)e, "layout_on_device");
                    }
                    b.words = bw; b.hdr = (const int4*)bh; b.groups = (const int4*)bg; b.frags = (const int4*)bf;
                    b.group_slices = p.batch_plan.group_slices; b.n_groups = (ns + p.batch_plan.group_slices - 1) / p.batch_plan.group_slices;
                    b.lds_floats = p.batch_plan.lds_floats + p.batch_dstream.stray_floats;
                    b.has_strays = p.batch_dstream.stray_floats > 0;
                    b.lookback = false; b.use_ticket = false;
                    if ((size_t)(b.lds_floats + b.ytile_floats * (b.block_threads / 64)) * 4 <= 160 * 1024 - 256) { p.batch_dev = b; p.has_batch_dev = true; }
                }
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
        for (auto& p : m.parts) { p.st = SliceStream{}; p.fix_short = {}; p.fix_long = {}; p.plan.groups = {}; p.plan.frags = {}; p.dstream = DeviceStream{};
                                  p.batch_plan.groups = {}; p.batch_plan.frags = {}; p.batch_plan.slice_spills = {}; p.batch_dstream = DeviceStream{}; p.batch_words = WordVec(); }
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
            c->d_stage = nullptr;
            // (HISPMV_HOST_Y=copy: y through a device buffer and a copy back, as until round 4)
            const bool direct_y = !(std::getenv("HISPMV_HOST_Y") && !std::strcmp(std::getenv("HISPMV_HOST_Y"), "copy"));
            if (direct_y && hipHostGetDevicePointer((void**)&c->d_stage, c->h_stage, 0) != hipSuccess) { (void)hipGetLastError(); c->d_stage = nullptr; }
        }
        std::memcpy(c->h_stage, x, bx);
        if (beta != 0.0f) std::memcpy(c->h_stage + nx, bias, bb);
        // (x and bias cross PCIe through a copy KERNEL that reads the pinned block: the DMA path costs ~10 us per call; HISPMV_HOST_Y=copy: as before)
        if (c->d_stage && (nx + nb) * (int64_t)sizeof(float) <= (1 << 20)) {
            const hipError_t e = launch_fetch_vectors(c->d_stage, d_x, beta != 0.0f ? nx + nb : (int64_t)m.cols * num_vecs, c->stream);
            if (e != hipSuccess) return hip_fail(c, e, "launch_fetch_vectors");
        } else
            HIP_TRY(c, hipMemcpyAsync(d_x, c->h_stage, beta != 0.0f ? (size_t)nx * sizeof(float) + bb : bx, hipMemcpyHostToDevice, c->stream));
    } else {
        HIP_TRY(c, hipMemcpyAsync(d_x, x, bx, hipMemcpyHostToDevice, c->stream));
        if (beta != 0.0f) HIP_TRY(c, hipMemcpyAsync(d_bias, bias, bb, hipMemcpyHostToDevice, c->stream));
    }
    // Small vectors: the kernels write y STRAIGHT into the pinned staging block (its device address; coherent host memory), so the
    // call has no copy back -- one DMA round trip (~10 us of a ~50 us call around a 10 - 20 us kernel) less.  The few read-modify-writes
    // of y (rows cut by slice boundaries, the merge of column parts) cross PCIe; they run in the tail launch, one round trip deep.
    float* const h_y = staged ? c->h_stage + nx + nb : y;
    float* const d_y = (staged && c->d_stage) ? c->d_stage + nx + nb : c->d_y;
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    if ((rc = launch_matrix_vectors(c, m, num_vecs, d_x, d_bias, d_y, alpha, beta, c->stream, is_linear)) != HISPMV_OK) return rc;
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    if (d_y == c->d_y) HIP_TRY(c, hipMemcpyAsync(h_y, c->d_y, by, hipMemcpyDeviceToHost, c->stream));
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
    out->batch_group_slices = 0;
    if (m.parts.size() == 1 && !m.parts[0].is_tts)
        out->batch_group_slices = m.loaded ? (m.parts[0].has_batch_dev ? m.parts[0].batch_dev.group_slices : 0) : (m.parts[0].has_batch_layout ? m.parts[0].batch_plan.group_slices : 0);
    return HISPMV_OK;
}

