// lds_atomic_bench.hip -- experiment: what does an LDS row-accumulator update cost on gfx950?
// Each wave issues N wave-instructions of the form acc[idx] (op)= v with idx from a precomputed table:
//   pattern 0: lane-consecutive (conflict-free), 1: random over the table, 2: random but bank-distinct inside each
//   32-lane group (what a packer could schedule), 3: every lane the same address.
//   op 0: ds_add_f32 (no return), 1: ds_add_u32, 2: ds_read_b32 + v_add + ds_write_b32 (not atomic), 3: ds_add_rtn_f32,
//   4: ds_write_b32 only, 5: ds_read_b32 only
// Prints cycles per wave-instruction per CU (16 waves per CU, 1 workgroup per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) unsigned lds_u32;

constexpr int kAcc = 32768;     // accumulators (128 KiB)
constexpr int kIter = 64;       // table entries per lane

template <int OP>
__global__ __launch_bounds__(1024) void k(const unsigned short* __restrict__ table, float* __restrict__ out, int reps) {
    extern __shared__ float acc[];
    for (int i = threadIdx.x; i < kAcc; i += blockDim.x) acc[i] = 0.f;
    __syncthreads();
    unsigned idx[kIter];
    const unsigned short* t = table + (size_t)(threadIdx.x >> 6) * kIter * 64 + (threadIdx.x & 63);
#pragma unroll
    for (int i = 0; i < kIter; ++i) idx[i] = t[i * 64];
    float s = 0.f;
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < kIter; ++i) {
            const float v = 1.0f + (float)i;
            if (OP == 0) __hip_atomic_fetch_add((lds_f32*)(acc + idx[i]), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (OP == 1) __hip_atomic_fetch_add((lds_u32*)((unsigned*)acc + idx[i]), (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (OP == 2) { acc[idx[i]] = acc[idx[i]] + v; }
            else if (OP == 3) s += __hip_atomic_fetch_add((lds_f32*)(acc + idx[i]), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (OP == 4) acc[idx[i]] = v;
            else if (OP == 5) s += acc[idx[i]];
        }
        if (OP == 6) {       // float add through compare-and-swap on the integer path, 16 elements in flight
#pragma unroll
            for (int b = 0; b < kIter; b += 16) {
                unsigned old[16], got[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) old[i] = ((unsigned*)acc)[idx[b + i]];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const unsigned nw = __builtin_bit_cast(unsigned, __builtin_bit_cast(float, old[i]) + (1.0f + (float)i));
                    got[i] = old[i];
                    __hip_atomic_compare_exchange_strong((lds_u32*)((unsigned*)acc + idx[b + i]), &got[i], nw, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    while (got[i] != old[i]) {          // lost a race (or a duplicate inside the wave): retry
                        old[i] = got[i];
                        const unsigned nw = __builtin_bit_cast(unsigned, __builtin_bit_cast(float, old[i]) + (1.0f + (float)i));
                        __hip_atomic_compare_exchange_strong((lds_u32*)((unsigned*)acc + idx[b + i]), &got[i], nw, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        }
        if (OP == 2 || OP == 5) asm volatile("" ::: "memory");
    }
    __syncthreads();
    if (s == 1234.5f || threadIdx.x == 0) out[blockIdx.x] = acc[threadIdx.x] + s;
}

int main() {
    std::mt19937 g(1);
    const int waves = 16;
    std::vector<unsigned short> tab((size_t)waves * kIter * 64);
    unsigned short* d_tab; float* d_out;
    CK(hipMalloc(&d_tab, tab.size() * 2)); CK(hipMalloc(&d_out, 4096));
    const char* pn[] = {"consecutive", "random", "random, bank-distinct per 32 lanes", "same address"};
    const char* on[] = {"ds_add_f32", "ds_add_u32", "read+add+write", "ds_add_rtn_f32", "ds_write_b32", "ds_read_b32", "cas float add x16"};
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const double mhz = prop.clockRate / 1000.0;
    for (int pat = 0; pat < 4; ++pat) {
        for (int w = 0; w < waves; ++w)
            for (int i = 0; i < kIter; ++i) {
                for (int half = 0; half < 2; ++half) {
                    std::vector<int> banks(32); for (int b = 0; b < 32; ++b) banks[b] = b;
                    std::shuffle(banks.begin(), banks.end(), g);
                    for (int l = 0; l < 32; ++l) {
                        unsigned v;
                        if (pat == 0) v = (unsigned)((w * kIter + i) * 64 + half * 32 + l) % kAcc;
                        else if (pat == 1) v = g() % kAcc;
                        else if (pat == 2) v = ((g() % (kAcc / 32)) * 32 + banks[l]);
                        else v = 77;
                        tab[((size_t)w * kIter + i) * 64 + half * 32 + l] = (unsigned short)v;
                    }
                }
            }
        CK(hipMemcpy(d_tab, tab.data(), tab.size() * 2, hipMemcpyHostToDevice));
        for (int op = 0; op < 7; ++op) {
            const int reps = (pat == 3 && (op == 0 || op == 3 || op == 6)) ? 2 : 50;
            auto launch = [&](int r) {
                const size_t lds = kAcc * 4;
#define L(OPV) case OPV: CK(hipFuncSetAttribute((const void*)k<OPV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); hipLaunchKernelGGL(k<OPV>, dim3(256), dim3(1024), lds, 0, d_tab, d_out, r); break;
                switch (op) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) }
#undef L
            };
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            launch(1); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); launch(reps); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms1; CK(hipEventElapsedTime(&ms1, e0, e1));
            CK(hipEventRecord(e0)); launch(3 * reps); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms3; CK(hipEventElapsedTime(&ms3, e0, e1));
            const double us = (ms3 - ms1) * 1e3 / (2.0 * reps);                     // per rep, launch overhead cancelled
            const double cyc = us * mhz / (kIter * waves);                          // per wave-instruction per CU
            printf("%-36s %-16s %8.2f us/rep  %7.1f cycles per wave-instruction per CU  (%.1f Gop/s chip)\n", pn[pat], on[op], us, cyc,
                   256.0 * waves * kIter * 64 / us / 1e3);
        }
    }
    return 0;
}
