// tile_lab.hip -- experiment (not product code): the "accumulator tile" stream for scattered short-row matrices.
//
// The reference keeps one row-sum buffer per PE in URAM and streams {row, col, val} words in whatever order balances
// the PEs (AccumBuffer, automation_tool/assets/base_functions.cpp:439-519; encode(), common/include/spmv-helper.h:45-60).
// The same idea on a CU: a workgroup owns a ROW TILE (its row sums live in LDS, up to ~38 K rows), the tile's elements
// are stored SORTED BY COLUMN, so the 64 lanes of a gather read neighbouring columns of x -- a handful of cache lines
// per wave instruction instead of 64 -- and every product is added to its row's accumulator with ds_add_f32.
// Words: {fp32 value, row_local:16 | col_off:16}; a slice of 1024 words has a header {col_base, ...}.
// Measures: soc-Pokec-like (uniform columns), R-MAT, for several tile heights R and column splits P.
//   hipcc --offload-arch=gfx950 -O3 -fopenmp tools/tile_lab.hip -o tools/tile_lab
#include <hip/hip_runtime.h>

#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kSlice = 1024, kSteps = 8;

struct Bucket { long long first_slice; int n_slices; int row0; int n_rows; int part; int pad0, pad1; };   // 32 B

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ntload(const uint4* p) {
    const u32x4 v = __builtin_nontemporal_load((const u32x4*)p);
    return uint4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ float i2f(unsigned i) { return __builtin_bit_cast(float, i); }
typedef __attribute__((address_space(3))) float lds_f32;

// MODE 0: full; 1: no LDS add (sink); 2: no gather (x = 1)
template <int MODE>
__global__ __launch_bounds__(1024) void tile_kernel(const uint4* __restrict__ words, const int* __restrict__ col_base,
                                                   const Bucket* __restrict__ buckets, const float* __restrict__ x,
                                                   const float* __restrict__ bias, float* __restrict__ out,
                                                   float alpha, float beta, int cols, int rows, int parts) {
    extern __shared__ float acc[];
    const Bucket b = buckets[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, cols * 4, 0x00020000);
    int k = wave;
    uint4 w[kSteps];
    int cb = 0;
    if (k < b.n_slices) {
        const uint4* p = words + (b.first_slice + k) * (kSlice / 2) + lane;
#pragma unroll
        for (int j = 0; j < kSteps; ++j) w[j] = ntload(p + j * 64);
        cb = col_base[b.first_slice + k];
    }
    for (int i = threadIdx.x; i <= b.n_rows; i += blockDim.x) acc[i] = 0.0f;
    __syncthreads();
    float sink = 0.f;
    while (k < b.n_slices) {
        const int base = __builtin_amdgcn_readfirstlane(cb) << 2;
        float x0[kSteps], x1[kSteps];
#pragma unroll
        for (int j = 0; j < kSteps; ++j) {
            if (MODE == 2) { x0[j] = 1.f; x1[j] = 1.f; }
            else {
                x0[j] = i2f(__builtin_amdgcn_raw_buffer_load_b32(rx, (w[j].y >> 16) << 2, base, 0));
                x1[j] = i2f(__builtin_amdgcn_raw_buffer_load_b32(rx, (w[j].w >> 16) << 2, base, 0));
            }
        }
        float p0[kSteps], p1[kSteps];
        unsigned r0[kSteps], r1[kSteps];
#pragma unroll
        for (int j = 0; j < kSteps; ++j) {
            p0[j] = i2f(w[j].x) * x0[j];
            p1[j] = i2f(w[j].z) * x1[j];
            r0[j] = w[j].y & 0xffffu; r1[j] = w[j].w & 0xffffu;
        }
#pragma unroll
        for (int j = 0; j < kSteps; ++j) asm volatile("" : "+v"(p0[j]), "+v"(p1[j]));
        asm volatile("" ::: "memory");
        k += n_waves;
        if (k < b.n_slices) {
            const uint4* p = words + (b.first_slice + k) * (kSlice / 2) + lane;
#pragma unroll
            for (int j = 0; j < kSteps; ++j) w[j] = ntload(p + j * 64);
            cb = col_base[b.first_slice + k];
        }
#pragma unroll
        for (int j = 0; j < kSteps; ++j) {
            if (MODE == 1) { sink += p0[j] + p1[j] + (float)(r0[j] + r1[j]); }
            else if (MODE == 3) {}
            else {
                __hip_atomic_fetch_add((lds_f32*)(acc + r0[j]), p0[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add((lds_f32*)(acc + r1[j]), p1[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (MODE == 3) {      // float add by compare-and-swap (integer path of the LDS), all 16 in flight
            typedef __attribute__((address_space(3))) unsigned lds_u32;
            unsigned o0[kSteps], o1[kSteps], g0[kSteps], g1[kSteps];
#pragma unroll
            for (int j = 0; j < kSteps; ++j) { o0[j] = ((unsigned*)acc)[r0[j]]; o1[j] = ((unsigned*)acc)[r1[j]]; }
#pragma unroll
            for (int j = 0; j < kSteps; ++j) {
                g0[j] = o0[j]; g1[j] = o1[j];
                __hip_atomic_compare_exchange_strong((lds_u32*)((unsigned*)acc + r0[j]), &g0[j], __builtin_bit_cast(unsigned, i2f(o0[j]) + p0[j]), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_compare_exchange_strong((lds_u32*)((unsigned*)acc + r1[j]), &g1[j], __builtin_bit_cast(unsigned, i2f(o1[j]) + p1[j]), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
#pragma unroll
            for (int j = 0; j < kSteps; ++j) {
                while (g0[j] != o0[j]) {
                    o0[j] = g0[j];
                    __hip_atomic_compare_exchange_strong((lds_u32*)((unsigned*)acc + r0[j]), &g0[j], __builtin_bit_cast(unsigned, i2f(o0[j]) + p0[j]), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                while (g1[j] != o1[j]) {
                    o1[j] = g1[j];
                    __hip_atomic_compare_exchange_strong((lds_u32*)((unsigned*)acc + r1[j]), &g1[j], __builtin_bit_cast(unsigned, i2f(o1[j]) + p1[j]), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }
    __syncthreads();
    if (MODE == 1 && sink == 12345.f) acc[0] = sink;
    if (parts == 1) {
        for (int i = threadIdx.x; i < b.n_rows; i += blockDim.x) out[b.row0 + i] = alpha * acc[i] + beta * bias[b.row0 + i];
    } else {
        float* o = out + (size_t)b.part * rows;
        for (int i = threadIdx.x; i < b.n_rows; i += blockDim.x) o[b.row0 + i] = acc[i];
    }
}

__global__ void merge_kernel(const float* __restrict__ partial, const float* __restrict__ bias, float* __restrict__ y,
                             float alpha, float beta, int rows, int parts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    float s = 0.f;
    for (int p = 0; p < parts; ++p) s += partial[(size_t)p * rows + i];
    y[i] = alpha * s + beta * bias[i];
}

struct Host { int rows, cols; std::vector<long long> rp; std::vector<int> ci; std::vector<float> va; };

static Host gen_uniform(int rows, int cols, long long nnz, bool powerlaw, unsigned seed) {
    Host m; m.rows = rows; m.cols = cols;
    std::mt19937_64 g(seed);
    std::vector<double> w(rows);
    std::vector<int> perm(rows); std::iota(perm.begin(), perm.end(), 0); std::shuffle(perm.begin(), perm.end(), g);
    double sum = 0;
    for (int i = 0; i < rows; ++i) { w[i] = powerlaw ? 1.0 / std::pow(perm[i] + 100.0, 0.8) : 1.0; sum += w[i]; }
    m.rp.assign(rows + 1, 0);
    for (int i = 0; i < rows; ++i) { std::poisson_distribution<int> d(w[i] * nnz / sum); m.rp[i + 1] = m.rp[i] + d(g); }
    const long long n = m.rp[rows];
    m.ci.resize(n); m.va.resize(n);
#pragma omp parallel
    {
        std::mt19937_64 gg(seed * 7919 + 13 * (unsigned)omp_get_thread_num());
#pragma omp for schedule(static)
        for (int i = 0; i < rows; ++i) {
            const long long s = m.rp[i], e = m.rp[i + 1];
            for (long long k = s; k < e; ++k) { m.ci[k] = (int)(gg() % (unsigned long long)cols); m.va[k] = (float)((gg() % 2000) / 1000.0 - 1.0) + 0.0005f; }
            std::sort(m.ci.begin() + s, m.ci.begin() + e);
        }
    }
    return m;
}

static Host gen_rmat(int scale, int ef, unsigned seed) {
    const int n = 1 << scale; const long long m_e = (long long)ef * n;
    std::vector<int> r(m_e), c(m_e);
#pragma omp parallel
    {
        std::mt19937_64 g(seed + 977 * (unsigned)omp_get_thread_num());
        std::uniform_real_distribution<double> u(0, 1);
#pragma omp for schedule(static)
        for (long long e = 0; e < m_e; ++e) {
            int rr = 0, cc = 0;
            for (int l = 0; l < scale; ++l) {
                const double q = u(g);
                const int down = q >= 0.76, right = (q >= 0.57 && q < 0.76) || q >= 0.95;
                rr = (rr << 1) | down; cc = (cc << 1) | right;
            }
            r[e] = rr; c[e] = cc;
        }
    }
    Host h; h.rows = n; h.cols = n; h.rp.assign(n + 1, 0);
    for (long long e = 0; e < m_e; ++e) h.rp[r[e] + 1]++;
    for (int i = 0; i < n; ++i) h.rp[i + 1] += h.rp[i];
    h.ci.resize(m_e); h.va.resize(m_e);
    std::vector<long long> cur(h.rp.begin(), h.rp.end() - 1);
    for (long long e = 0; e < m_e; ++e) h.ci[cur[r[e]]++] = c[e];
#pragma omp parallel for schedule(dynamic, 1024)
    for (int i = 0; i < n; ++i) std::sort(h.ci.begin() + h.rp[i], h.ci.begin() + h.rp[i + 1]);
    for (long long e = 0; e < m_e; ++e) h.va[e] = (float)((e * 2654435761u % 2000) / 1000.0 - 1.0) + 0.0005f;
    return h;
}

struct Packed {
    std::vector<uint64_t> words; std::vector<int> col_base; std::vector<Bucket> buckets; int parts; int max_rows;
    double lines_per_gather = 0;   // distinct 128-B lines per 64-lane gather instruction (diagnostic)
};

// Row tiles of about `target_nnz` elements but at most `max_rows` rows; each cut into `parts` column ranges of equal
// element count; inside a bucket the elements are sorted by (column, row).
static Packed pack(const Host& m, long long target_nnz, int max_rows, int parts) {
    Packed P; P.parts = parts; P.max_rows = 0;
    struct Tile { int r0, r1; };
    std::vector<Tile> tiles;
    for (int r = 0; r < m.rows;) {
        int e = r;
        while (e < m.rows && e - r < max_rows && (e == r || m.rp[e + 1] - m.rp[r] <= target_nnz)) ++e;
        tiles.push_back({r, e});
        r = e;
    }
    const size_t nb = tiles.size() * (size_t)parts;
    std::vector<std::vector<uint64_t>> bw(nb);
    std::vector<std::vector<int>> bbase(nb);
    std::vector<double> lines(nb, 0.0); std::vector<long long> gathers(nb, 0);
#pragma omp parallel for schedule(dynamic, 1)
    for (long long t = 0; t < (long long)tiles.size(); ++t) {
        const Tile T = tiles[t];
        struct E { int col; unsigned short row; float v; };
        std::vector<E> el; el.reserve(m.rp[T.r1] - m.rp[T.r0]);
        for (int r = T.r0; r < T.r1; ++r)
            for (long long k = m.rp[r]; k < m.rp[r + 1]; ++k) el.push_back({m.ci[k], (unsigned short)(r - T.r0), m.va[k]});
        std::sort(el.begin(), el.end(), [](const E& a, const E& b) { return a.col != b.col ? a.col < b.col : a.row < b.row; });
        const size_t n = el.size();
        for (int p = 0; p < parts; ++p) {
            const size_t b0 = n * p / parts, b1 = n * (p + 1) / parts;
            std::vector<uint64_t>& W = bw[t * parts + p];
            std::vector<int>& B = bbase[t * parts + p];
            const unsigned dummy = (unsigned)(T.r1 - T.r0);          // accumulator slot that is never read
            size_t i = b0;
            while (i < b1) {
                const int base = el[i].col & ~31;                     // 128-byte aligned window base
                size_t j = i;
                while (j < b1 && j - i < (size_t)kSlice && el[j].col - base < 65536) ++j;
                B.push_back(base);
                for (size_t q = 0; q < (size_t)kSlice; ++q) {
                    uint64_t word;
                    if (i + q < j) {
                        unsigned vb; memcpy(&vb, &el[i + q].v, 4);
                        word = ((uint64_t)(((unsigned)(el[i + q].col - base) << 16) | el[i + q].row) << 32) | vb;
                    } else word = ((uint64_t)dummy << 32);
                    W.push_back(word);
                }
                // diagnostic: distinct lines per 64-lane gather (lanes take every second element of 128)
                for (size_t s = i; s < j; s += 128) {
                    for (int half = 0; half < 2; ++half) {
                        int last = -1, cnt = 0;
                        for (size_t q = s + half; q < std::min(j, s + 128); q += 2) { const int ln = el[q].col >> 5; if (ln != last) { ++cnt; last = ln; } }
                        lines[t * parts + p] += cnt; gathers[t * parts + p]++;
                    }
                }
                i = j;
            }
        }
    }
    long long slice = 0; double L = 0; long long G = 0;
    for (size_t t = 0; t < tiles.size(); ++t)
        for (int p = 0; p < parts; ++p) {
            const size_t b = t * parts + p;
            Bucket bk{slice, (int)bbase[b].size(), tiles[t].r0, tiles[t].r1 - tiles[t].r0, p, 0, 0};
            P.buckets.push_back(bk);
            P.words.insert(P.words.end(), bw[b].begin(), bw[b].end());
            P.col_base.insert(P.col_base.end(), bbase[b].begin(), bbase[b].end());
            slice += bk.n_slices;
            P.max_rows = std::max(P.max_rows, bk.n_rows);
            L += lines[b]; G += gathers[b];
        }
    P.lines_per_gather = G ? L / G : 0;
    return P;
}

template <int MODE>
static float time_kernel(const Packed& P, const uint64_t* d_words, const int* d_base, const Bucket* d_b, const float* d_x,
                         const float* d_bias, float* d_y, float* d_partial, int rows, int cols, int reps) {
    auto k = tile_kernel<MODE>;
    const size_t lds = (size_t)(P.max_rows + 1) * 4;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float* out = P.parts == 1 ? d_y : d_partial;
    auto go = [&]() {
        hipLaunchKernelGGL(k, dim3((unsigned)P.buckets.size()), dim3(1024), lds, 0, (const uint4*)d_words, d_base, d_b, d_x, d_bias, out, 0.85f, -2.06f, cols, rows, P.parts);
        if (P.parts > 1) hipLaunchKernelGGL(merge_kernel, dim3((rows + 255) / 256), dim3(256), 0, 0, d_partial, d_bias, d_y, 0.85f, -2.06f, rows, P.parts);
    };
    go(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) go();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / reps;
}

static void experiment(const char* name, const Host& m, const std::vector<std::pair<int, int>>& configs) {
    const long long nnz = m.rp[m.rows];
    const double alg = 8.0 * nnz + 16.0 * m.rows;
    printf("== %s: %d x %d, nnz %lld, alg %.1f MB\n", name, m.rows, m.cols, nnz, alg / 1e6);
    std::vector<float> x(m.cols), bias(m.rows);
    for (int j = 0; j < m.cols; ++j) x[j] = (float)(j + 1) / (float)(j + 2);
    for (int i = 0; i < m.rows; ++i) bias[i] = -2.0f * (float)(i + 1) / (float)(i + 2);
    std::vector<double> y64(m.rows), mag(m.rows);
#pragma omp parallel for schedule(dynamic, 1024)
    for (int i = 0; i < m.rows; ++i) {
        double s = 0, a = 0;
        for (long long k = m.rp[i]; k < m.rp[i + 1]; ++k) { const double p = (double)m.va[k] * x[m.ci[k]]; s += p; a += std::fabs(p); }
        y64[i] = 0.85 * s + (double)-2.06f * bias[i]; mag[i] = 0.85 * a + std::fabs((double)-2.06f * bias[i]);
    }
    float *d_x, *d_bias, *d_y, *d_partial = nullptr;
    CK(hipMalloc(&d_x, (size_t)m.cols * 4 + 256)); CK(hipMemcpy(d_x, x.data(), (size_t)m.cols * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_bias, (size_t)m.rows * 4)); CK(hipMemcpy(d_bias, bias.data(), (size_t)m.rows * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_y, (size_t)m.rows * 4));
    for (auto cfg : configs) {
        const int max_rows = cfg.first, parts = cfg.second;
        const long long target = std::max<long long>(64 * 1024, (long long)((double)nnz / m.rows * max_rows));
        Packed P = pack(m, target, max_rows, parts);
        uint64_t* d_w; int* d_base; Bucket* d_b;
        CK(hipMalloc(&d_w, P.words.size() * 8)); CK(hipMemcpy(d_w, P.words.data(), P.words.size() * 8, hipMemcpyHostToDevice));
        CK(hipMalloc(&d_base, P.col_base.size() * 4)); CK(hipMemcpy(d_base, P.col_base.data(), P.col_base.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&d_b, P.buckets.size() * sizeof(Bucket))); CK(hipMemcpy(d_b, P.buckets.data(), P.buckets.size() * sizeof(Bucket), hipMemcpyHostToDevice));
        if (parts > 1) CK(hipMalloc(&d_partial, (size_t)parts * m.rows * 4));
        int mx = 0, mn = 1 << 30; for (auto& b : P.buckets) { mx = std::max(mx, b.n_slices); mn = std::min(mn, b.n_slices); }
        const float t_full = time_kernel<0>(P, d_w, d_base, d_b, d_x, d_bias, d_y, d_partial, m.rows, m.cols, 10);
        std::vector<float> y(m.rows);
        CK(hipMemcpy(y.data(), d_y, (size_t)m.rows * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int i = 0; i < m.rows; ++i) worst = std::max(worst, std::fabs(y[i] - y64[i]) / std::max(mag[i], 1e-300));
        const float t_noadd = time_kernel<1>(P, d_w, d_base, d_b, d_x, d_bias, d_y, d_partial, m.rows, m.cols, 10);
        const float t_nogather = time_kernel<2>(P, d_w, d_base, d_b, d_x, d_bias, d_y, d_partial, m.rows, m.cols, 10);
        const float t_cas = time_kernel<3>(P, d_w, d_base, d_b, d_x, d_bias, d_y, d_partial, m.rows, m.cols, 10);
        CK(hipMemcpy(y.data(), d_y, (size_t)m.rows * 4, hipMemcpyDeviceToHost));
        double worst_cas = 0;
        for (int i = 0; i < m.rows; ++i) worst_cas = std::max(worst_cas, std::fabs(y[i] - y64[i]) / std::max(mag[i], 1e-300));
        printf("  R<=%5d P=%d: buckets %5zu slices/bucket %d..%d pad %.1f%% lines/gather %.1f | full %7.1f us %6.1f GB/s | no-add %7.1f | no-gather %7.1f | CAS %7.1f us %6.1f GB/s err %.2e | bwd err %.2e\n",
               max_rows, parts, P.buckets.size(), mn, mx, 100.0 * ((double)P.words.size() / nnz - 1.0), P.lines_per_gather,
               t_full * 1e3, alg / t_full / 1e6, t_noadd * 1e3, t_nogather * 1e3, t_cas * 1e3, alg / t_cas / 1e6, worst_cas, worst);
        fflush(stdout);
        CK(hipFree(d_w)); CK(hipFree(d_base)); CK(hipFree(d_b)); if (d_partial) { CK(hipFree(d_partial)); d_partial = nullptr; }
    }
    CK(hipFree(d_x)); CK(hipFree(d_bias)); CK(hipFree(d_y));
}

int main(int argc, char** argv) {
    const char* which = argc > 1 ? argv[1] : "all";
    const std::vector<std::pair<int, int>> cfgs = {{6400, 1}, {12800, 1}, {12800, 2}, {25600, 2}, {25600, 4}, {38000, 4}, {38000, 6}, {38000, 8}};
    if (!strcmp(which, "all") || !strcmp(which, "pokec")) {
        Host m = gen_uniform(1632803, 1632803, 30622600, true, 1);
        experiment("soc-Pokec-like (power-law rows, uniform columns)", m, cfgs);
    }
    if (!strcmp(which, "all") || !strcmp(which, "rmat")) {
        Host m = gen_rmat(20, 16, 42);
        experiment("R-MAT scale 20 ef 16", m, cfgs);
    }
    if (!strcmp(which, "all") || !strcmp(which, "asic")) {
        Host m = gen_uniform(682862, 682862, 2639000, false, 3);
        experiment("ASIC_680k-like (4/row, uniform columns)", m, {{12800, 1}, {25600, 1}, {38000, 1}, {38000, 2}});
    }
    return 0;
}
