// gather_bench.hip -- microbenchmark behind DESIGN.md's x-gather decisions (not product code).
// Each lane streams 16 B of (value, index) pairs exactly like the SpMV slice kernel and gathers
// x[index]; variants differ in how x is reached:
//   plain  : global_load_dword through L1/L2
//   nt     : non-temporal load
//   sc1    : agent-scope relaxed atomic load (bypasses L1)
//   lds    : x window staged into LDS once per workgroup, ds_read_b32 gathers
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o tools/gather_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

enum Mode { PLAIN = 0, NT = 1, SC1 = 2 };

template <int MODE>
__device__ __forceinline__ float ld(const float* p) {
    if (MODE == NT) return __builtin_nontemporal_load(p);
    if (MODE == SC1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

// one wave per 1024-element chunk: 8 x dwordx4 stream loads, 16 gathers, sum
template <int MODE>
__global__ __launch_bounds__(256) void gather_global(const uint4* __restrict__ words, const float* __restrict__ x,
                                                     float* __restrict__ out, long long n_chunks) {
    const int lane = threadIdx.x & 63;
    const long long chunk = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (chunk >= n_chunks) return;
    const uint4* p = words + chunk * 512 + lane;
    uint4 w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = p[j * 64];
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        acc += __builtin_bit_cast(float, w[j].x) * ld<MODE>(x + w[j].y);
        acc += __builtin_bit_cast(float, w[j].z) * ld<MODE>(x + w[j].w);
    }
    if (acc == 123.456f) out[chunk] = acc;   // keep it live, (almost) never store
}

// workgroup of 256 threads: stage `span` floats of x (window at x_base per chunk group) into LDS, then
// each wave processes `per_wave` chunks gathering from LDS
__global__ __launch_bounds__(256) void gather_lds(const uint4* __restrict__ words, const float* __restrict__ x,
                                                  float* __restrict__ out, long long n_chunks, int span, int per_wave) {
    extern __shared__ float xs[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long first = ((long long)blockIdx.x * 4 + wv) * per_wave;
    for (int i = threadIdx.x * 4; i < span; i += 1024) *(float4*)(xs + i) = *(const float4*)(x + i);
    __syncthreads();
    float acc = 0.f;
    for (int c = 0; c < per_wave; ++c) {
        const long long chunk = first + c;
        if (chunk >= n_chunks) break;
        const uint4* p = words + chunk * 512 + lane;
        uint4 w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = p[j * 64];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc += __builtin_bit_cast(float, w[j].x) * xs[w[j].y];
            acc += __builtin_bit_cast(float, w[j].z) * xs[w[j].w];
        }
    }
    if (acc == 123.456f) out[first] = acc;
}

// stream only (no gather): the ceiling of the 8 B/element stream
__global__ __launch_bounds__(256) void stream_only(const uint4* __restrict__ words, float* __restrict__ out, long long n_chunks) {
    const int lane = threadIdx.x & 63;
    const long long chunk = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (chunk >= n_chunks) return;
    const uint4* p = words + chunk * 512 + lane;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { uint4 w = p[j * 64]; acc += __builtin_bit_cast(float, w.x) + __builtin_bit_cast(float, w.z) + (float)(w.y ^ w.w); }
    if (acc == 123.456f) out[chunk] = acc;
}


int main(int argc, char** argv) {
    const long long n_elems = 64ll << 20;       // 64 Mi elements = 512 MiB of stream (> Infinity Cache)
    const long long n_chunks = n_elems / 1024;
    std::vector<uint64_t> h(n_elems);
    uint64_t* d_words; float *d_x, *d_out;
    CK(hipMalloc(&d_words, n_elems * 8));
    CK(hipMalloc(&d_x, 256ll << 20));
    CK(hipMalloc(&d_out, n_chunks * 4 + 1024));
    CK(hipMemset(d_x, 0, 256ll << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, long long table, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int reps = 5;
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("%-28s table %9lld B : %8.1f us  %7.1f Gelem/s  stream %7.1f GB/s\n", name, table, ms * 1e3,
               n_elems / ms / 1e6, n_elems * 8.0 / ms / 1e6);
        fflush(stdout);
    };
    std::mt19937_64 g(1);
    // pattern: 0 = uniform random in table; 1 = window-local: each 4096-element group draws from a 16 KiB window
    // sliding through the table; 2 = runs of 4 consecutive columns, uniform run starts
    std::vector<long long> tables = {16 << 10, 128 << 10, 1 << 20, 8ll << 20, 64ll << 20};
    int n_patterns = 3;
    if (argc > 1) {               // ./gather_bench <bytes>[,<bytes>...]: uniform pattern only, these table sizes
        tables.clear(); n_patterns = 1;
        for (char* tok = strtok(argv[1], ","); tok; tok = strtok(nullptr, ",")) tables.push_back(atoll(tok));
    }
    for (int pattern = 0; pattern < n_patterns; ++pattern) {
        for (long long table : tables) {
            const long long nf = table / 4;
            for (long long i = 0; i < n_elems; ++i) {
                uint32_t idx;
                if (pattern == 0) idx = (uint32_t)(g() % nf);
                else if (pattern == 1) { long long win = 4096, base = ((i / 4096) * 257) % (nf > win ? nf - win : 1); idx = (uint32_t)(base + g() % (nf > win ? win : nf)); }
                else { if ((i & 3) == 0) idx = (uint32_t)(g() % (nf - 4)); else idx = (uint32_t)(h[i - 1] >> 32) + 1; }
                float v = 1.0f; uint32_t vb; memcpy(&vb, &v, 4);
                h[i] = ((uint64_t)idx << 32) | vb;
            }
            CK(hipMemcpy(d_words, h.data(), n_elems * 8, hipMemcpyHostToDevice));
            printf("--- pattern %d (%s)\n", pattern, pattern == 0 ? "uniform" : pattern == 1 ? "sliding 16 KiB window" : "runs of 4");
            dim3 grid((unsigned)((n_chunks + 3) / 4)), blk(256);
            if (pattern == 0 && table == tables[0]) run("stream_only", 0, [&] { hipLaunchKernelGGL(stream_only, grid, blk, 0, 0, (const uint4*)d_words, d_out, n_chunks); });
            run("global plain", table, [&] { hipLaunchKernelGGL(gather_global<PLAIN>, grid, blk, 0, 0, (const uint4*)d_words, d_x, d_out, n_chunks); });
            run("global nt", table, [&] { hipLaunchKernelGGL(gather_global<NT>, grid, blk, 0, 0, (const uint4*)d_words, d_x, d_out, n_chunks); });
            run("global sc1", table, [&] { hipLaunchKernelGGL(gather_global<SC1>, grid, blk, 0, 0, (const uint4*)d_words, d_x, d_out, n_chunks); });
            if (pattern == 0 && table <= (128 << 10)) {
                for (int per_wave : {4, 16, 64}) {
                    dim3 g2((unsigned)((n_chunks + 4 * per_wave - 1) / (4 * per_wave)));
                    char nm[64]; snprintf(nm, sizeof nm, "lds window, %d chunks/wave", per_wave);
                    CK(hipFuncSetAttribute((const void*)gather_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)table));
                    run(nm, table, [&] { hipLaunchKernelGGL(gather_lds, g2, blk, (size_t)table, 0, (const uint4*)d_words, d_x, d_out, n_chunks, (int)nf, per_wave); });
                }
            }
        }
    }
    return 0;
}
