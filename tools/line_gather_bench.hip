// line_gather_bench.hip -- microbenchmark (not product code) behind the round-4 question "what does a gather of
// column-SORTED elements cost in the vector cache, per lane or per line?" (DESIGN.md 2.2).
// A wavefront takes groups of 64 elements whose columns are sorted and fall into a span of S floats of x (a tile-stream
// slice: 64 consecutive elements of the column order).  Variants:
//   lane   : one buffer/global dword load per lane (what spmv_tts_* does today)
//   line<NI>: the group's distinct 128-byte lines of x (at most 8*NI) are loaded cooperatively -- 8 lanes x 16 B per line,
//            NI load instructions --, written to a per-wavefront LDS scratch and gathered from there with ds_read_b32
//   line-only<NI>: the loads of the line variant without the LDS round trip (the cache side alone)
// Build: hipcc --offload-arch=gfx950 -O3 tools/line_gather_bench.hip -o tools/line_gather_bench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kGroupsPerWave = 16;        // one 1024-element slice

__global__ __launch_bounds__(256) void gather_lane(const uint32_t* __restrict__ cols, const float* __restrict__ x,
                                                   float* __restrict__ out, long long n_groups) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long g0 = wave * kGroupsPerWave;
    if (g0 >= n_groups) return;
    uint32_t c[kGroupsPerWave];
#pragma unroll
    for (int g = 0; g < kGroupsPerWave; ++g) c[g] = cols[(g0 + g) * 64 + lane];
    float acc = 0.f;
#pragma unroll
    for (int g = 0; g < kGroupsPerWave; ++g) acc += x[c[g]];
    if (acc == 123.456f) out[wave] = acc;
}

// the lane gather through a buffer descriptor with the cache-policy bits of the instruction (AUX: 1 = sc0, 2 = nt, 16 = sc1 on gfx94x/gfx950)
template <int AUX>
__global__ __launch_bounds__(256) void gather_lane_aux(const uint32_t* __restrict__ cols, const float* __restrict__ x,
                                                       float* __restrict__ out, long long n_groups, unsigned table_bytes) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long g0 = wave * kGroupsPerWave;
    if (g0 >= n_groups) return;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)table_bytes, 0x00020000);
    uint32_t c[kGroupsPerWave];
#pragma unroll
    for (int g = 0; g < kGroupsPerWave; ++g) c[g] = cols[(g0 + g) * 64 + lane];
    float acc = 0.f;
#pragma unroll
    for (int g = 0; g < kGroupsPerWave; ++g) acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, c[g] << 2, 0, AUX));
    if (acc == 123.456f) out[wave] = acc;
}

// the same gathers with the table cut in two halves: workgroups of XCDs 0-3 (blockIdx mod 8 < 4) read the lower half, the others the
// upper half -- what pinning the column parts of a tile stream to XCD subsets is meant to buy (an XCD's L2 holds half of x)
__global__ __launch_bounds__(256) void gather_lane_halves(const uint32_t* __restrict__ cols, const float* __restrict__ x,
                                                          float* __restrict__ out, long long n_groups, uint32_t half, int pinned) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long g0 = wave * kGroupsPerWave;
    if (g0 >= n_groups) return;
    const uint32_t off = (pinned ? (blockIdx.x & 4) != 0 : (blockIdx.x & 8) != 0) ? half : 0u;
    uint32_t c[kGroupsPerWave];
#pragma unroll
    for (int g = 0; g < kGroupsPerWave; ++g) c[g] = cols[(g0 + g) * 64 + lane] % half + off;
    float acc = 0.f;
#pragma unroll
    for (int g = 0; g < kGroupsPerWave; ++g) acc += x[c[g]];
    if (acc == 123.456f) out[wave] = acc;
}

template <int NI, bool LDS>
__global__ __launch_bounds__(256) void gather_line(const uint32_t* __restrict__ lines, const uint16_t* __restrict__ loc,
                                                   const float* __restrict__ x, float* __restrict__ out, long long n_groups) {
    __shared__ float scratch[4][2][NI * 8 * 32];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long wave = (long long)blockIdx.x * 4 + wv;
    const long long g0 = wave * kGroupsPerWave;
    if (g0 >= n_groups) return;
    float acc = 0.f;
    constexpr int U = 4;       // groups in flight
#pragma unroll 1
    for (int gb = 0; gb < kGroupsPerWave; gb += U) {
        uint32_t id[U]; uint32_t lc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            id[u] = lane < NI * 8 ? lines[(g0 + gb + u) * (NI * 8) + lane] : 0u;
            lc[u] = loc[(g0 + gb + u) * 64 + lane];
        }
        float4 v[U][NI];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const uint32_t lid = (uint32_t)__shfl((int)id[u], i * 8 + (lane >> 3));
                v[u][i] = *(const float4*)(x + (size_t)lid * 32 + (lane & 7) * 4);
            }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (LDS) {
                float* s = scratch[wv][u & 1];
#pragma unroll
                for (int i = 0; i < NI; ++i) *(float4*)(s + (i * 8 + (lane >> 3)) * 32 + (lane & 7) * 4) = v[u][i];
                acc += s[lc[u]];
            } else {
#pragma unroll
                for (int i = 0; i < NI; ++i) acc += v[u][i].x + v[u][i].y + v[u][i].z + v[u][i].w;
                acc += (float)lc[u];
            }
        }
    }
    if (acc == 123.456f) out[wave] = acc;
}

int main(int argc, char** argv) {
    const long long n_elems = 32ll << 20;
    const long long n_groups = n_elems / 64;
    const long long table_floats = (argc > 1 ? atoll(argv[1]) : (6ll << 20) + (512 << 10)) / 4;      // ~ soc-Pokec's x
    float *d_x, *d_out; uint32_t *d_cols, *d_lines; uint16_t* d_loc;
    CK(hipMalloc(&d_x, table_floats * 4 + 4096));
    CK(hipMemset(d_x, 0, table_floats * 4 + 4096));
    CK(hipMalloc(&d_out, (n_groups / kGroupsPerWave + 8) * 4));
    CK(hipMalloc(&d_cols, n_elems * 4));
    CK(hipMalloc(&d_lines, n_groups * 64 * 4));
    CK(hipMalloc(&d_loc, n_elems * 2));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int n_cus = 256; { hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); n_cus = p.multiProcessorCount; printf("CUs %d clock %d kHz\n", n_cus, p.clockRate); }
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int reps = 10;
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("  %-22s %8.1f us  %7.1f Gelem/s  %5.2f cycles/elem/CU at 2.4 GHz\n", name, ms * 1e3, n_elems / ms / 1e6,
               ms * 1e-3 * 2.4e9 * n_cus / n_elems);
        fflush(stdout);
    };
    std::mt19937_64 rng(1);
    std::vector<uint32_t> cols(n_elems), lines;
    std::vector<uint16_t> loc(n_elems);
    // spans in floats: 680 ~ soc-Pokec's standard geometry (23 64-byte lines per gather), 340 ~ the tall geometry, 170, 1400
    for (int span : {170, 340, 680, 1400}) {
        int max_lines = 0; double sum_lines = 0, sum_lines64 = 0;
        std::vector<std::vector<uint32_t>> glines((size_t)n_groups);
        for (long long g = 0; g < n_groups; ++g) {
            const uint32_t base = (uint32_t)(rng() % (uint64_t)(table_floats - span));
            uint32_t c[64];
            for (int l = 0; l < 64; ++l) c[l] = base + (uint32_t)(rng() % (uint64_t)span);
            std::sort(c, c + 64);
            auto& gl = glines[(size_t)g];
            int n64 = 0; uint32_t last64 = ~0u;
            for (int l = 0; l < 64; ++l) {
                if (gl.empty() || gl.back() != c[l] / 32) gl.push_back(c[l] / 32);
                if (c[l] / 16 != last64) { ++n64; last64 = c[l] / 16; }
                cols[(size_t)g * 64 + l] = c[l];
                loc[(size_t)g * 64 + l] = (uint16_t)((gl.size() - 1) * 32 + c[l] % 32);
            }
            max_lines = std::max(max_lines, (int)gl.size()); sum_lines += (double)gl.size(); sum_lines64 += n64;
        }
        const int NI = (max_lines + 7) / 8;
        lines.assign((size_t)n_groups * NI * 8, 0);
        for (long long g = 0; g < n_groups; ++g) {
            const auto& gl = glines[(size_t)g];
            for (int k = 0; k < NI * 8; ++k) lines[(size_t)g * NI * 8 + k] = gl[std::min((size_t)k, gl.size() - 1)];
        }
        CK(hipMemcpy(d_cols, cols.data(), n_elems * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_lines, lines.data(), lines.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_loc, loc.data(), n_elems * 2, hipMemcpyHostToDevice));
        printf("--- span %d floats: %.1f 128-byte lines (%.1f 64-byte lines) per 64 sorted elements, max %d -> %d load instructions\n", span,
               sum_lines / n_groups, sum_lines64 / n_groups, max_lines, NI);
        dim3 grid((unsigned)((n_groups / kGroupsPerWave + 3) / 4)), blk(256);
        run("lane (dword gather)", [&] { hipLaunchKernelGGL(gather_lane, grid, blk, 0, 0, d_cols, d_x, d_out, n_groups); });
#define AUXRUN(A) run("lane, buffer load aux " #A, [&] { hipLaunchKernelGGL((gather_lane_aux<A>), grid, blk, 0, 0, d_cols, d_x, d_out, n_groups, (unsigned)(table_floats * 4)); });
        AUXRUN(0) AUXRUN(1) AUXRUN(2) AUXRUN(3) AUXRUN(16) AUXRUN(17) AUXRUN(18)
        run("lane, halves by XCD", [&] { hipLaunchKernelGGL(gather_lane_halves, grid, blk, 0, 0, d_cols, d_x, d_out, n_groups, (uint32_t)(table_floats / 2), 1); });
        run("lane, halves not by XCD", [&] { hipLaunchKernelGGL(gather_lane_halves, grid, blk, 0, 0, d_cols, d_x, d_out, n_groups, (uint32_t)(table_floats / 2), 0); });
#define LINE(N) \
        if (NI == N) { \
            run("line loads + LDS", [&] { hipLaunchKernelGGL((gather_line<N, true>), grid, blk, 0, 0, d_lines, d_loc, d_x, d_out, n_groups); }); \
            run("line loads only", [&] { hipLaunchKernelGGL((gather_line<N, false>), grid, blk, 0, 0, d_lines, d_loc, d_x, d_out, n_groups); }); \
        }
        LINE(1) LINE(2) LINE(3) LINE(4) LINE(5) LINE(6) LINE(7) LINE(8)
    }
    return 0;
}
