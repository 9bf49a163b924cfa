// kernel_lab.hip -- experiment bench for the slice kernel (not product code): times variants of the
// per-slice pipeline on one synthetic matrix inside ONE process (interleaved rounds), to find which part
// of the kernel costs what.  Build: make -C tools lab   (links the product's host packer).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../hispmv_amd/csrc/hispmv_format.h"
#include "../hispmv_amd/csrc/hispmv_prep.h"

using namespace hispmv;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ int f2i(float f) { return __builtin_bit_cast(int, f); }
__device__ __forceinline__ float i2f(int i) { return __builtin_bit_cast(float, i); }

template <int CTRL, int ROWMASK>
__device__ __forceinline__ void seg_scan_step(float& v, int& F) {
    const float vp = i2f(__builtin_amdgcn_update_dpp(0, f2i(v), CTRL, ROWMASK, 0xf, false));
    const int Fp = __builtin_amdgcn_update_dpp(0, F, CTRL, ROWMASK, 0xf, false);
    v = F ? v : v + vp;
    F |= Fp;
}
__device__ __forceinline__ void seg_scan_wave(float& v, int& F) {
    seg_scan_step<0x111, 0xf>(v, F); seg_scan_step<0x112, 0xf>(v, F); seg_scan_step<0x114, 0xf>(v, F);
    seg_scan_step<0x118, 0xf>(v, F); seg_scan_step<0x142, 0xa>(v, F); seg_scan_step<0x143, 0xc>(v, F);
}
__device__ __forceinline__ float wave_total(float v) {   // plain DPP reduction, total in lane 63
    v += i2f(__builtin_amdgcn_update_dpp(0, f2i(v), 0x111, 0xf, 0xf, false));
    v += i2f(__builtin_amdgcn_update_dpp(0, f2i(v), 0x112, 0xf, 0xf, false));
    v += i2f(__builtin_amdgcn_update_dpp(0, f2i(v), 0x114, 0xf, 0xf, false));
    v += i2f(__builtin_amdgcn_update_dpp(0, f2i(v), 0x118, 0xf, 0xf, false));
    v += i2f(__builtin_amdgcn_update_dpp(0, f2i(v), 0x142, 0xa, 0xf, false));
    v += i2f(__builtin_amdgcn_update_dpp(0, f2i(v), 0x143, 0xc, 0xf, false));
    return i2f(__builtin_amdgcn_readlane(f2i(v), 63));
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ntload(const uint4* p) {
    const u32x4 v = __builtin_nontemporal_load((const u32x4*)p);
    return uint4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ int lanes_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

// SCAN: 0 none (sum only), 1 full segmented scan every step, 2 plain reduction when the step has no row end
// OUT : 0 no bias/y traffic, 1 full output
// LDS : x window of the whole chunk in LDS (staged once), else buffer gathers
template <int SCAN, int OUT, bool LDS, bool PIPE, bool NT = false>
__global__ __launch_bounds__(1024) void lab_kernel(const uint4* __restrict__ words, const int4* __restrict__ hdr,
                                                  const float* __restrict__ x, const float* bias, float* y,
                                                  float* __restrict__ carry, float alpha, float beta,
                                                  long long n_slices, int group_slices, int x_base, int x_span, int cols, int rows) {
    extern __shared__ float xs[];
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, cols * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)bias, 0, rows * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, rows * 4, 0x00020000);
    constexpr unsigned kNoAccess = 0xffffffffu;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const long long first = (long long)blockIdx.x * group_slices;
    const long long last = (first + group_slices < n_slices) ? first + group_slices : n_slices;
    long long slice = first + wave;
    uint4 w[kSliceSteps];
    int4 h = int4{0, 0, 0, 0};
    if (slice < last) {
        const uint4* p = words + slice * (kSliceElems / 2) + lane;
#pragma unroll
        for (int j = 0; j < kSliceSteps; ++j) w[j] = NT ? ntload(p + j * 64) : p[j * 64];
        h = hdr[slice];
    }
    if (LDS) {
        const float4* src = (const float4*)(x + x_base);
        for (int i = threadIdx.x; i < (x_span >> 2); i += blockDim.x) ((float4*)xs)[i] = src[i];
        __syncthreads();
    }
    float sink = 0.f;
    while (slice < last) {
        if (!PIPE && false) {}
        int row = __builtin_amdgcn_readfirstlane(h.x);
        int r0[kSliceSteps];
        if (OUT) {
#pragma unroll
            for (int j = 0; j < kSliceSteps; ++j) {
                const unsigned long long m0 = __builtin_amdgcn_ballot_w64((w[j].y & kRowEndBit) != 0);
                const unsigned long long m1 = __builtin_amdgcn_ballot_w64((w[j].w & kRowEndBit) != 0);
                r0[j] = row + lanes_below(m0) + lanes_below(m1);
                row += __builtin_popcountll(m0) + __builtin_popcountll(m1);
            }
        }
        float x0[kSliceSteps], x1[kSliceSteps], b0[kSliceSteps], b1[kSliceSteps];
#pragma unroll
        for (int j = 0; j < kSliceSteps; ++j) {
            if (LDS) {
                x0[j] = xs[(int)(w[j].y & ~kRowEndBit) - x_base];
                x1[j] = xs[(int)(w[j].w & ~kRowEndBit) - x_base];
            } else {
                x0[j] = i2f((int)__builtin_amdgcn_raw_buffer_load_b32(rx, (w[j].y & ~kRowEndBit) << 2, 0, 0));
                x1[j] = i2f((int)__builtin_amdgcn_raw_buffer_load_b32(rx, (w[j].w & ~kRowEndBit) << 2, 0, 0));
            }
        }
        if (OUT == 1) {
#pragma unroll
            for (int j = 0; j < kSliceSteps; ++j) {
                const bool e0 = (w[j].y & kRowEndBit) != 0, e1 = (w[j].w & kRowEndBit) != 0;
                b0[j] = i2f((int)__builtin_amdgcn_raw_buffer_load_b32(rb, e0 ? (unsigned)r0[j] << 2 : kNoAccess, 0, 0));
                b1[j] = i2f((int)__builtin_amdgcn_raw_buffer_load_b32(rb, e1 ? (unsigned)(r0[j] + (e0 ? 1 : 0)) << 2 : kNoAccess, 0, 0));
            }
        }
        float p0[kSliceSteps], p1[kSliceSteps];
        unsigned ends = 0;
#pragma unroll
        for (int j = 0; j < kSliceSteps; ++j) {
            p0[j] = i2f((int)w[j].x) * x0[j];
            p1[j] = i2f((int)w[j].z) * x1[j];
            ends |= ((w[j].y >> 31) << (2 * j)) | ((w[j].w >> 31) << (2 * j + 1));
        }
        const long long cur = slice;
        slice += n_waves;
        if (PIPE && slice < last) {
            const uint4* p = words + slice * (kSliceElems / 2) + lane;
#pragma unroll
            for (int j = 0; j < kSliceSteps; ++j) w[j] = NT ? ntload(p + j * 64) : p[j * 64];
            h = hdr[slice];
        }
        float t0[kSliceSteps], t1[kSliceSteps];
        float carry_step = 0.0f;
#pragma unroll
        for (int j = 0; j < kSliceSteps; ++j) {
            const bool e0 = (ends >> (2 * j)) & 1u, e1 = (ends >> (2 * j + 1)) & 1u;
            if (SCAN == 0) {
                carry_step += p0[j] + p1[j];
                t0[j] = t1[j] = carry_step;
            } else {
                const unsigned any = (unsigned)__builtin_amdgcn_readfirstlane((int)(__builtin_amdgcn_ballot_w64(e0 | e1) != 0ull));
                if (SCAN == 2 && !any) {
                    carry_step += wave_total(p0[j] + p1[j]);
                    t0[j] = t1[j] = 0.f;
                } else {
                    float v = e1 ? 0.0f : (e0 ? p1[j] : p0[j] + p1[j]);
                    int F = (e0 | e1) ? 1 : 0;
                    seg_scan_wave(v, F);
                    v = F ? v : v + carry_step;
                    const float cin = i2f(__builtin_amdgcn_update_dpp(f2i(carry_step), f2i(v), 0x138, 0xf, 0xf, false));
                    carry_step = i2f(__builtin_amdgcn_readlane(f2i(v), 63));
                    t0[j] = cin + p0[j];
                    t1[j] = e0 ? p1[j] : cin + (p0[j] + p1[j]);
                }
            }
        }
        if (OUT == 2) {
            float* yt = xs + (LDS ? x_span : 0) + wave * 1024;
            const int row_first = __builtin_amdgcn_readfirstlane(h.x);
            const int n_rows = row - row_first;
#pragma unroll
            for (int j = 0; j < kSliceSteps; ++j) {
                const bool e0 = (ends >> (2 * j)) & 1u, e1 = (ends >> (2 * j + 1)) & 1u;
                if (e0) yt[r0[j] - row_first] = t0[j];
                if (e1) yt[r0[j] - row_first + (e0 ? 1 : 0)] = t1[j];
            }
            for (int i = lane; i < n_rows; i += 64) {
                const float t = yt[i];
                const float b = i2f((int)__builtin_amdgcn_raw_buffer_load_b32(rb, (unsigned)(row_first + i) << 2, 0, 0));
                __builtin_amdgcn_raw_buffer_store_b32((unsigned)f2i(alpha * t + beta * b), ry, (unsigned)(row_first + i) << 2, 0, 0);
            }
        } else if (OUT) {
#pragma unroll
            for (int j = 0; j < kSliceSteps; ++j) {
                const bool e0 = (ends >> (2 * j)) & 1u, e1 = (ends >> (2 * j + 1)) & 1u;
                const float y0 = alpha * t0[j] + beta * b0[j];
                const float y1 = alpha * t1[j] + beta * b1[j];
                __builtin_amdgcn_raw_buffer_store_b32((unsigned)f2i(y0), ry, e0 ? (unsigned)r0[j] << 2 : kNoAccess, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32((unsigned)f2i(y1), ry, e1 ? (unsigned)(r0[j] + (e0 ? 1 : 0)) << 2 : kNoAccess, 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < kSliceSteps; ++j) sink += t0[j] + t1[j];
        }
        if (lane == 0) carry[cur] = carry_step + sink * 1e-30f;
        if (!PIPE && slice < last) {
            const uint4* p = words + slice * (kSliceElems / 2) + lane;
#pragma unroll
            for (int j = 0; j < kSliceSteps; ++j) w[j] = NT ? ntload(p + j * 64) : p[j * 64];
            h = hdr[slice];
        }
    }
}

struct Mat { uint64_t* words; int4* hdr; float *x, *bias, *y, *carry; long long n_slices; int rows, cols, x_base, x_span; double alg_bytes; };

static Mat make(int rows, int cols, int row_len, int band, int run, unsigned seed) {
    std::mt19937 g(seed);
    Csr m; m.rows = rows; m.cols = cols; m.row_ptr.assign(rows + 1, 0);
    for (int i = 0; i < rows; ++i) m.row_ptr[i + 1] = m.row_ptr[i] + row_len;
    m.col.resize((size_t)rows * row_len); m.val.resize(m.col.size());
    std::vector<int> tmp(row_len);
    for (int i = 0; i < rows; ++i) {
        long long centre = (long long)i * cols / rows;
        int lo = (int)std::max<long long>(0, centre - band), hi = (int)std::min<long long>(cols, centre + band);
        for (int k = 0; k < row_len; k += run) { int c = lo + (int)(g() % (unsigned)std::max(1, hi - lo - run)); for (int q = 0; q < run && k + q < row_len; ++q) tmp[k + q] = c + q; }
        std::sort(tmp.begin(), tmp.end());
        for (int k = 0; k < row_len; ++k) { m.col[(size_t)i * row_len + k] = tmp[k]; m.val[(size_t)i * row_len + k] = 1.0f + (g() % 7) * 0.125f; }
    }
    SliceStream st = build_stream(m);
    Mat d{}; d.n_slices = st.n_slices; d.rows = rows; d.cols = cols;
    d.alg_bytes = 8.0 * m.nnz() + 16.0 * rows;
    CK(hipMalloc(&d.words, st.words.size() * 8)); CK(hipMemcpy(d.words, st.words.data(), st.words.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&d.hdr, st.hdr.size() * 16)); CK(hipMemcpy(d.hdr, st.hdr.data(), st.hdr.size() * 16, hipMemcpyHostToDevice));
    CK(hipMalloc(&d.x, cols * 4 + 64)); CK(hipMemset(d.x, 0, cols * 4 + 64));
    CK(hipMalloc(&d.bias, rows * 4)); CK(hipMemset(d.bias, 0, rows * 4));
    CK(hipMalloc(&d.y, rows * 4)); CK(hipMalloc(&d.carry, st.n_slices * 4));
    d.x_base = 0; d.x_span = (cols + 3) & ~3;
    return d;
}

static void* g_flush = nullptr;   // when set: 1 GiB memset between warm-up and the timed launches (cold caches)
template <int SCAN, int OUT, bool LDS, bool PIPE, bool NT = false>
static float run(const Mat& m, int threads, int per_cu, int reps, int group_override = 0) {
    const int group = group_override ? group_override : (int)((m.n_slices + 256LL * per_cu - 1) / (256LL * per_cu));
    const unsigned grid = (unsigned)((m.n_slices + group - 1) / group);
    const size_t lds = (LDS ? (size_t)m.x_span * 4 : 0) + (OUT == 2 ? (size_t)(threads / 64) * 4096 : 0);
    auto k = lab_kernel<SCAN, OUT, LDS, PIPE, NT>;
    if (lds) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, 0, (const uint4*)m.words, m.hdr, m.x, m.bias, m.y, m.carry, 0.5f, -2.f, m.n_slices, group, m.x_base, m.x_span, m.cols, m.rows);
    if (g_flush) CK(hipMemsetAsync(g_flush, 1, 1u << 30, 0));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, 0, (const uint4*)m.words, m.hdr, m.x, m.bias, m.y, m.carry, 0.5f, -2.f, m.n_slices, group, m.x_base, m.x_span, m.cols, m.rows);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

static Mat make_stencil(int rows, int row_len, int band, int run, unsigned seed) {
    std::mt19937 g(seed);
    std::vector<int> off;
    for (int k = 0; k < row_len; k += run) { int o = (int)(g() % (unsigned)(2 * band)) - band; for (int q = 0; q < run; ++q) off.push_back(o + q); }
    std::sort(off.begin(), off.end()); off.erase(std::unique(off.begin(), off.end()), off.end());
    Csr m; m.rows = rows; m.cols = rows; m.row_ptr.assign(rows + 1, 0);
    for (int i = 0; i < rows; ++i) { int n = 0; for (int o : off) n += (i + o >= 0 && i + o < rows); m.row_ptr[i + 1] = m.row_ptr[i] + n; }
    m.col.resize(m.row_ptr[rows]); m.val.resize(m.col.size());
    for (int i = 0; i < rows; ++i) { int64_t k = m.row_ptr[i]; for (int o : off) if (i + o >= 0 && i + o < rows) { m.col[k] = i + o; m.val[k] = 1.0f + (g() % 7) * 0.125f; ++k; } }
    SliceStream st = build_stream(m);
    Mat d{}; d.n_slices = st.n_slices; d.rows = rows; d.cols = rows;
    d.alg_bytes = 8.0 * m.nnz() + 16.0 * rows;
    CK(hipMalloc(&d.words, st.words.size() * 8)); CK(hipMemcpy(d.words, st.words.data(), st.words.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&d.hdr, st.hdr.size() * 16)); CK(hipMemcpy(d.hdr, st.hdr.data(), st.hdr.size() * 16, hipMemcpyHostToDevice));
    CK(hipMalloc(&d.x, rows * 4 + 64)); CK(hipMemset(d.x, 0, rows * 4 + 64));
    CK(hipMalloc(&d.bias, rows * 4)); CK(hipMemset(d.bias, 0, rows * 4));
    CK(hipMalloc(&d.y, rows * 4)); CK(hipMalloc(&d.carry, st.n_slices * 4));
    d.x_base = 0; d.x_span = (rows + 3) & ~3;
    return d;
}

int main(int argc, char** argv) {
    if (argc > 1 && argv[1][0] == 'S') {   // small short-row matrices: where do 20 us go?
        Mat A = make(682862, 682862, 4, 682862, 1, 7);      // ASIC_680k-like: uniform columns
        Mat A2 = make(682862, 682862, 4, 64, 1, 7);         // same shape, columns next to the diagonal
        Mat N = make(414604, 414604, 6, 50000, 1, 8);       // nxp1-like
        Mat L = make(20000, 20000, 128, 3000, 4, 9);        // same stream size as A, long rows
        float* flush; CK(hipMalloc(&flush, 1u << 30));
        struct { const char* name; Mat* m; } mm[] = {{"A uniform 4/row", &A}, {"A2 diagonal 4/row", &A2}, {"N banded 6/row", &N}, {"L long rows", &L}};
        for (auto& q : mm) {
            const Mat& m = *q.m;
            printf("== %s: slices %lld, alg %.1f MB\n", q.name, m.n_slices, m.alg_bytes / 1e6);
            for (int round = 0; round < 2; ++round) {
                auto rep = [&](const char* v, float ms) { printf("  r%d %-44s %8.1f us  %7.1f GB/s\n", round, v, ms * 1e3, m.alg_bytes / ms / 1e6); fflush(stdout); };
                rep("glb ytile 256t 4 sl/WG", run<1, 2, false, true, true>(m, 256, 4, 20, 4));
                rep("glb out0  256t 4 sl/WG", run<1, 0, false, true, true>(m, 256, 4, 20, 4));
                rep("glb out1  256t 4 sl/WG", run<1, 1, false, true, true>(m, 256, 4, 20, 4));
                rep("glb noscan out0 256t 4 sl/WG", run<0, 0, false, true, true>(m, 256, 4, 20, 4));
                rep("glb ytile 256t 8 sl/WG", run<1, 2, false, true, true>(m, 256, 4, 20, 8));
                rep("glb ytile 512t 16 sl/WG", run<1, 2, false, true, true>(m, 512, 2, 20, 16));
                rep("glb ytile 256t 2 sl/WG", run<1, 2, false, true, true>(m, 256, 4, 20, 2));
                rep("glb ytile 128t 2 sl/WG", run<1, 2, false, true, true>(m, 128, 4, 20, 2));
                g_flush = flush;
                rep("glb ytile 256t 4 sl/WG, 1 rep after flush", run<1, 2, false, true, true>(m, 256, 4, 1, 4));
                rep("glb out0  256t 4 sl/WG, 1 rep after flush", run<1, 0, false, true, true>(m, 256, 4, 1, 4));
                g_flush = nullptr;
            }
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'E') {   // current product structure (LDS row-total tile) and its parts
        Mat B = make(250000, 4096 * 4, 200, 8000, 4, 2);
        Mat C = make(6000000, 16384, 8, 8000, 2, 3);
        Mat D = make_stencil(1000000, 50, 30000, 4, 5);
        struct { const char* name; Mat* m; bool lds; } mm[] = {{"B long rows, x in LDS", &B, true}, {"C short rows, x in LDS", &C, true}, {"D stencil, global gather", &D, false}};
        for (auto& q : mm) {
            const Mat& m = *q.m;
            printf("== %s: slices %lld, alg %.0f MB\n", q.name, m.n_slices, m.alg_bytes / 1e6);
            for (int round = 0; round < 2; ++round) {
                auto rep = [&](const char* v, float ms) { printf("  r%d %-44s %8.1f us  %7.1f GB/s\n", round, v, ms * 1e3, m.alg_bytes / ms / 1e6); fflush(stdout); };
                if (q.lds) {
                    rep("lds scan1 ytile  512t x2/CU", run<1, 2, true, true>(m, 512, 2, 5));
                    rep("lds scan1 out0   512t x2/CU", run<1, 0, true, true>(m, 512, 2, 5));
                    rep("lds scan0 ytile  512t x2/CU nt", run<0, 2, true, true, true>(m, 512, 2, 5));
                    rep("lds scan1 ytile  512t x2/CU nt", run<1, 2, true, true, true>(m, 512, 2, 5));
                    rep("lds scan0 ytile 1024t x1/CU nt", run<0, 2, true, true, true>(m, 1024, 1, 5));
                    rep("lds scan1 ytile 1024t x1/CU nt", run<1, 2, true, true, true>(m, 1024, 1, 5));
                    rep("lds scan0 out0  1024t x1/CU nt", run<0, 0, true, true, true>(m, 1024, 1, 5));
                    rep("lds scan1 ytile 1024t x1/CU", run<1, 2, true, true>(m, 1024, 1, 5));
                    rep("lds scan1 ytile  256t x4/CU", run<1, 2, true, true>(m, 256, 4, 5));
                    rep("lds scan1 ytile nopipe 512t x2/CU", run<1, 2, true, false>(m, 512, 2, 5));
                } else {
                    rep("glb scan1 ytile  256t 8 slices/WG", run<1, 2, false, true>(m, 256, 4, 5, 8));
                    rep("glb scan1 out0   256t 8 slices/WG", run<1, 0, false, true>(m, 256, 4, 5, 8));
                    rep("glb scan0 ytile  256t 8 slices/WG nt", run<0, 2, false, true, true>(m, 256, 4, 5, 8));
                    rep("glb scan1 ytile  256t 8 slices/WG nt", run<1, 2, false, true, true>(m, 256, 4, 5, 8));
                    rep("glb scan1 ytile  512t 16 slices/WG", run<1, 2, false, true>(m, 512, 2, 5, 16));
                    rep("glb scan1 ytile nopipe 256t 8 slices/WG", run<1, 2, false, false>(m, 256, 4, 5, 8));
                }
            }
        }
        return 0;
    }
    if (argc > 1) {   // second experiment: stencil-like (PFlow-like) matrix, L2 gathers with reuse between rows
        Mat D = make_stencil(1000000, 50, 30000, 4, 5);
        printf("== D stencil 50/row band 30000 run 4: slices %lld, alg %.0f MB\n", D.n_slices, D.alg_bytes / 1e6);
        for (int round = 0; round < 2; ++round) {
            auto rep = [&](const char* v, float ms) { printf("  r%d %-44s %8.1f us  %7.1f GB/s\n", round, v, ms * 1e3, D.alg_bytes / ms / 1e6); fflush(stdout); };
            rep("glb full      256t 8 slices/WG", run<1, 1, false, true>(D, 256, 4, 5, 8));
            rep("glb full NT   256t 8 slices/WG", run<1, 1, false, true, true>(D, 256, 4, 5, 8));
            rep("glb full      512t 16 slices/WG", run<1, 1, false, true>(D, 512, 2, 5, 16));
            rep("glb full      256t 4 slices/WG", run<1, 1, false, true>(D, 256, 4, 5, 4));
            rep("glb full      256t x4/CU resident", run<1, 1, false, true>(D, 256, 4, 5));
            rep("glb full      1024t x1/CU resident", run<1, 1, false, true>(D, 1024, 1, 5));
            rep("glb scan0out0 256t 8 slices/WG", run<0, 0, false, true>(D, 256, 4, 5, 8));
            rep("glb scan1out0 256t 8 slices/WG", run<1, 0, false, true>(D, 256, 4, 5, 8));
            rep("glb nopipe    256t 8 slices/WG", run<1, 1, false, false>(D, 256, 4, 5, 8));
        }
        return 0;
    }
    // long rows, narrow window (TSOPF-like), 3x the size so that launches do not live in the Infinity Cache
    Mat A = make(114360, 114360, 424, 2400, 8, 1);      // ~388 MB stream, window whole x = 457 KB -> global; see B
    Mat B = make(250000, 4096 * 4, 200, 8000, 4, 2);    // 400 MB stream, x = 64 KiB: whole x in LDS
    Mat C = make(6000000, 16384, 8, 8000, 2, 3);        // short rows (8/row), 384 MB, x = 64 KiB in LDS: row-end heavy
    struct { const char* name; Mat* m; } mats[] = {{"B long rows, x in LDS", &B}, {"C short rows, x in LDS", &C}, {"A long rows, global gather", &A}};
    for (auto& mm : mats) {
        const Mat& m = *mm.m;
        printf("== %s: slices %lld, alg %.0f MB\n", mm.name, m.n_slices, m.alg_bytes / 1e6);
        const bool lds_ok = m.x_span * 4 <= 70 * 1024;
        for (int round = 0; round < 2; ++round) {
            auto rep = [&](const char* v, float ms) { printf("  r%d %-44s %8.1f us  %7.1f GB/s\n", round, v, ms * 1e3, m.alg_bytes / ms / 1e6); fflush(stdout); };
            if (lds_ok) {
                rep("lds scan1 out1 pipe   512t x2/CU", run<1, 1, true, true>(m, 512, 2, 5));
                rep("lds scan1 out1 nopipe 512t x2/CU", run<1, 1, true, false>(m, 512, 2, 5));
                rep("lds scan2 out1 pipe   512t x2/CU", run<2, 1, true, true>(m, 512, 2, 5));
                rep("lds scan1 out0 pipe   512t x2/CU", run<1, 0, true, true>(m, 512, 2, 5));
                rep("lds scan0 out0 pipe   512t x2/CU", run<0, 0, true, true>(m, 512, 2, 5));
                rep("lds scan2 out1 pipe   256t x4/CU", run<2, 1, true, true>(m, 256, 4, 5));
                rep("lds scan2 out1 pipe   256t x2/CU", run<2, 1, true, true>(m, 256, 2, 5));
            } else {
                rep("glb scan1 out1 pipe   256t x4/CU", run<1, 1, false, true>(m, 256, 4, 5));
                rep("glb scan1 out1 nopipe 256t x4/CU", run<1, 1, false, false>(m, 256, 4, 5));
                rep("glb scan2 out1 pipe   256t x4/CU", run<2, 1, false, true>(m, 256, 4, 5));
                rep("glb scan0 out0 pipe   256t x4/CU", run<0, 0, false, true>(m, 256, 4, 5));
                rep("glb scan2 out1 pipe   256t x8/CU", run<2, 1, false, true>(m, 256, 8, 5));
            }
        }
    }
    return 0;
}
