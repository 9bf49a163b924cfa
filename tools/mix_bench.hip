// mix_bench.hip -- microbenchmark (not product code) behind the round-4 question "does a CU overlap cache-fill-bound gathers with
// an HBM-bound stream when BOTH run in the same workgroup?" (profiles/r4_experiments/mixed_workgroup.json).
// One persistent 1024-thread workgroup per CU.  Wavefronts [0, NG) are GATHER wavefronts: slices of 1024 column-sorted elements
// (columns + values streamed from HBM: 8 B per element, x gathered with one dword per lane from a 6.5 MB table, 64 consecutive
// elements of a span of 680 floats per instruction: soc-Pokec's standard tiles); wavefronts [NG, NG + NS) are STREAM wavefronts:
// 6 KiB "slices" read with dwordx4 loads, the next one requested before the current one is consumed (the slice kernels' pattern).
// Each role's work is dealt statically over the grid.  Times: gather alone, stream alone, both together.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mix_bench.hip -o tools/mix_bench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <random>
#include <vector>

typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 nt_load4(const uint4* p) { const v4u t = __builtin_nontemporal_load((const v4u*)p); return uint4{t.x, t.y, t.z, t.w}; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(1024) void mixed(const uint32_t* __restrict__ cols, const float* __restrict__ vals, const float* __restrict__ x,
                                              long long n_gslices, const uint4* __restrict__ stream, long long n_sslices,
                                              int ng, int ns, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    if (wave < ng) {
        const long long stride = (long long)gridDim.x * ng;
        long long s = (long long)blockIdx.x * ng + wave;
        uint32_t c[16]; float v[16];
        if (s < n_gslices) {
#pragma unroll
            for (int g = 0; g < 16; ++g) { c[g] = __builtin_nontemporal_load(cols + s * 1024 + g * 64 + lane); v[g] = __builtin_nontemporal_load(vals + s * 1024 + g * 64 + lane); }
        }
        while (s < n_gslices) {
            float xv[16];
#pragma unroll
            for (int g = 0; g < 16; ++g) xv[g] = x[c[g]];
            float p = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) p += xv[g] * v[g];
            acc += p;
            s += stride;
            if (s < n_gslices) {
#pragma unroll
                for (int g = 0; g < 16; ++g) { c[g] = __builtin_nontemporal_load(cols + s * 1024 + g * 64 + lane); v[g] = __builtin_nontemporal_load(vals + s * 1024 + g * 64 + lane); }
            }
        }
    } else if (wave < ng + ns) {
        const long long stride = (long long)gridDim.x * ns;
        long long s = (long long)blockIdx.x * ns + (wave - ng);
        uint4 w[6];
        if (s < n_sslices) {
#pragma unroll
            for (int j = 0; j < 6; ++j) w[j] = nt_load4(stream + s * 384 + j * 64 + lane);
        }
        while (s < n_sslices) {
            unsigned t = 0;
#pragma unroll
            for (int j = 0; j < 6; ++j) t += w[j].x ^ w[j].y ^ w[j].z ^ w[j].w;
            s += stride;
            if (s < n_sslices) {
#pragma unroll
                for (int j = 0; j < 6; ++j) w[j] = nt_load4(stream + s * 384 + j * 64 + lane);
            }
            // some LDS-free arithmetic per slice, as the scans of the slice kernel would do
#pragma unroll
            for (int k = 0; k < 24; ++k) t = t * 1664525u + 1013904223u;
            acc += (float)(t & 1);
        }
    }
    if (acc == 123.456f) out[blockIdx.x * 16 + wave] = acc;
}

int main(int argc, char** argv) {
    const long long n_gelems = 30ll << 20;                  // ~ soc-Pokec's elements
    const long long n_gslices = n_gelems / 1024;
    const long long stream_bytes = argc > 1 ? atoll(argv[1]) << 20 : 512ll << 20;
    const long long n_sslices = stream_bytes / 6144;
    const int span = argc > 2 ? atoi(argv[2]) : 680;
    const long long table_floats = (argc > 3 ? atoll(argv[3]) * 1024 : (6ll << 20) + (512 << 10)) / 4;      // argv[3]: KiB of x
    float *d_x, *d_out, *d_vals; uint32_t* d_cols; uint4* d_stream;
    CK(hipMalloc(&d_x, table_floats * 4 + 4096)); CK(hipMemset(d_x, 0, table_floats * 4 + 4096));
    CK(hipMalloc(&d_out, 1 << 20));
    CK(hipMalloc(&d_cols, n_gelems * 4)); CK(hipMalloc(&d_vals, n_gelems * 4)); CK(hipMemset(d_vals, 0, n_gelems * 4));
    CK(hipMalloc(&d_stream, n_sslices * 6144)); CK(hipMemset(d_stream, 1, n_sslices * 6144));
    std::mt19937_64 rng(1);
    std::vector<uint32_t> cols((size_t)n_gelems);
    for (long long g = 0; g < n_gelems / 64; ++g) {
        const uint32_t base = (uint32_t)(rng() % (uint64_t)(table_floats - span));
        uint32_t c[64];
        for (int l = 0; l < 64; ++l) c[l] = base + (uint32_t)(rng() % (uint64_t)span);
        std::sort(c, c + 64);
        for (int l = 0; l < 64; ++l) cols[(size_t)g * 64 + l] = c[l];
    }
    CK(hipMemcpy(d_cols, cols.data(), n_gelems * 4, hipMemcpyHostToDevice));
    int n_cus = 256; { hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); n_cus = p.multiProcessorCount; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("CUs %d; gather role: %lld elements (%.0f MB of columns + values), span %d, x %.2f MB; stream role: %.0f MB\n", n_cus, n_gelems, n_gelems * 8 / 1e6, span, table_floats * 4 / 1e6, n_sslices * 6144 / 1e6);
    auto run = [&](const char* name, int ng, int ns, long long gs, long long ss) {
        auto launch = [&] { hipLaunchKernelGGL(mixed, dim3((unsigned)n_cus), dim3(1024), 0, 0, d_cols, d_vals, d_x, gs, d_stream, ss, ng, ns, d_out); };
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int reps = 10;
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("  %-44s %8.1f us", name, ms * 1e3);
        if (gs) printf("   gather %6.1f Gelem/s", gs * 1024 / ms / 1e6);
        if (ss) printf("   stream %6.2f TB/s", ss * 6144 / ms / 1e9);
        printf("\n"); fflush(stdout);
        return ms * 1e3f;
    };
    const float g16 = run("gather alone, 16 wavefronts", 16, 0, n_gslices, 0);
    const float g12 = run("gather alone, 12 wavefronts", 12, 0, n_gslices, 0);
    const float g8 = run("gather alone, 8 wavefronts", 8, 0, n_gslices, 0);
    const float g4 = run("gather alone, 4 wavefronts", 4, 0, n_gslices, 0);
    const float s16 = run("stream alone, 16 wavefronts", 0, 16, 0, n_sslices);
    const float s8 = run("stream alone, 8 wavefronts", 0, 8, 0, n_sslices);
    const float s4 = run("stream alone, 4 wavefronts", 0, 4, 0, n_sslices);
    const float m88 = run("both: 8 gather + 8 stream wavefronts", 8, 8, n_gslices, n_sslices);
    const float m124 = run("both: 12 gather + 4 stream wavefronts", 12, 4, n_gslices, n_sslices);
    const float m412 = run("both: 4 gather + 12 stream wavefronts", 4, 12, n_gslices, n_sslices);
    printf("serial (16 + 16 wavefronts) %.1f us; mixed 8+8 %.1f (%.2fx), 12+4 %.1f (%.2fx), 4+12 %.1f (%.2fx)\n", g16 + s16, m88, (g16 + s16) / m88,
           m124, (g16 + s16) / m124, m412, (g16 + s16) / m412);
    (void)g12; (void)g8; (void)g4; (void)s8; (void)s4;
    return 0;
}
