// hispmv_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the SpMV hot path.
//
// What the reference does with a dataflow of TAPA tasks
// (automation_tool/assets/base_functions.cpp) happens here inside one wavefront:
//   MM2S_A  :3    stream the packed words          -> 8 x global_load_dwordx4 per lane (coalesced, 8 KiB/wave)
//   LoadB/ComputeAB :105,:158  val * x[col]         -> per-lane gather of x + v_mul
//   PreAccumulator :257 + ADD/SWB/SSW :356-437      -> in-register pair combine + DPP segmented scan
//      (tree-reduce the parts of a row that sit      (row_shr 1,2,4,8 / row_bcast15 / row_bcast31):
//       in different PEs and route to the owner)      lanes holding a row end own the row total
//   AccumBuffer :439 (per-PE URAM row sums)         -> running carry in a VGPR/SGPR across the 8 steps
//   Compute_C :521  beta*c_in + alpha*acc           -> fused into the store of the row total
//   rows "shared" between PEs                       -> rows cut by a slice boundary: each slice hands its
//                                                      open partial sum to carry[slice]; the fix-up kernel
//                                                      adds the chain to the owner row in fixed order
//                                                      (bitwise reproducible, no float atomics)
// Bandwidth-bound gather: no MFMA.  Roofline and byte accounting: DESIGN.md.
#include <hip/hip_runtime.h>

#include "hispmv_format.h"
#include "hispmv_kernels.h"

namespace hispmv {

// ---------------------------------------------------------------------------
// wave64 helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ int f2i(float f) { return __builtin_bit_cast(int, f); }
__device__ __forceinline__ float i2f(int i) { return __builtin_bit_cast(float, i); }

// One Kogge-Stone step of the segmented inclusive scan on (head flag F, value v):
//   (F1,v1) o (F2,v2) = (F1|F2, F2 ? v2 : v1+v2)
// CTRL/ROWMASK select the DPP source; lanes without a source read the identity (0,0).
template <int CTRL, int ROWMASK>
__device__ __forceinline__ void seg_scan_step(float& v, int& F) {
    const float vp = i2f(__builtin_amdgcn_update_dpp(0, f2i(v), CTRL, ROWMASK, 0xf, false));
    const int Fp = __builtin_amdgcn_update_dpp(0, F, CTRL, ROWMASK, 0xf, false);
    v = F ? v : v + vp;
    F |= Fp;
}

__device__ __forceinline__ void seg_scan_wave(float& v, int& F) {
    seg_scan_step<0x111, 0xf>(v, F);   // row_shr:1
    seg_scan_step<0x112, 0xf>(v, F);   // row_shr:2
    seg_scan_step<0x114, 0xf>(v, F);   // row_shr:4
    seg_scan_step<0x118, 0xf>(v, F);   // row_shr:8
    seg_scan_step<0x142, 0xa>(v, F);   // row_bcast:15 -> rows 1,3
    seg_scan_step<0x143, 0xc>(v, F);   // row_bcast:31 -> rows 2,3
}

__device__ __forceinline__ float wave_sum(float v) {
    v += i2f(__builtin_amdgcn_update_dpp(0, f2i(v), 0x111, 0xf, 0xf, false));
    v += i2f(__builtin_amdgcn_update_dpp(0, f2i(v), 0x112, 0xf, 0xf, false));
    v += i2f(__builtin_amdgcn_update_dpp(0, f2i(v), 0x114, 0xf, 0xf, false));
    v += i2f(__builtin_amdgcn_update_dpp(0, f2i(v), 0x118, 0xf, 0xf, false));
    v += i2f(__builtin_amdgcn_update_dpp(0, f2i(v), 0x142, 0xa, 0xf, false));
    v += i2f(__builtin_amdgcn_update_dpp(0, f2i(v), 0x143, 0xc, 0xf, false));
    return i2f(__builtin_amdgcn_readlane(f2i(v), 63));   // lane 63 holds the total
}

constexpr int kLookbackSpinMax = 1 << 22;   // ~seconds; a wait this long means a lost launch, not contention

__device__ __forceinline__ int lanes_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

// ---------------------------------------------------------------------------
// Slice kernel.  One workgroup owns a GROUP of consecutive slices (the analogue of a PE group fed by
// one x window, LoadB base_functions.cpp:105-150); its wavefronts take the group's slices round-robin.
//   phase 0 (group): the group's x window [x_base, x_base+x_span) is staged into LDS with coalesced
//            16-byte loads when it fits (USE_LDS and span <= lds_floats); otherwise x is gathered from L2.
//   phase 1 (slice): 8 x global_load_dwordx4 bring the 8 KiB slice; 16 gathers (ds_read_b32 or global).
//   phase 2: 8 steps of pair-combine + DPP segmented scan; row totals stay in registers.
//   phase 3: all bias loads of the slice, then all y stores (alpha*total + beta*bias), then carry.
// No global store sits between a slice's loads, so hipcc keeps them all in flight together.
// ---------------------------------------------------------------------------
template <bool HAS_BETA, bool USE_LDS, bool LOOKBACK>
__global__ __launch_bounds__(1024) void spmv_slices_kernel(
    const uint4* __restrict__ words, const int4* __restrict__ hdr, const int2* __restrict__ groups,
    const float* __restrict__ x, const float* __restrict__ bias, float* __restrict__ y,
    float* __restrict__ carry, float alpha, float beta, long long n_slices, int group_slices,
    int lds_floats, int cols, LookbackArgs lb) {
    extern __shared__ float xs[];
    __shared__ long long s_group;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    long long group = blockIdx.x;
    if (LOOKBACK) {
        // Groups are handed out in START order (one ticket per workgroup), so every slice a wavefront
        // may have to wait for belongs to a workgroup that is already running or done -- the carry
        // look-back below cannot deadlock whatever order the dispatcher picks.  The ticket counter is
        // never reset: launch k consumes tickets [k*n_groups, (k+1)*n_groups).
        if (threadIdx.x == 0)
            s_group = (long long)(__hip_atomic_fetch_add(lb.ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - lb.ticket_base);
        __syncthreads();
        group = s_group;
    }

    int x_base = 0;
    bool in_lds = false;
    if (USE_LDS) {
        const int2 g = groups[group];
        x_base = g.x;
        in_lds = g.y <= lds_floats;          // workgroup-uniform
        if (in_lds) {
            const int span4 = g.y >> 2;      // x_base is a multiple of 4 floats (host), x is 16-B aligned
            const float4* src = (const float4*)(x + x_base);
            for (int i = threadIdx.x; i < span4; i += blockDim.x) ((float4*)xs)[i] = src[i];
            for (int i = (span4 << 2) + threadIdx.x; i < g.y; i += blockDim.x)
                xs[i] = (x_base + i < cols) ? x[x_base + i] : 0.0f;
        }
        __syncthreads();
    }
    const long long first = group * group_slices;

    for (long long slice = first + wave; slice < first + group_slices && slice < n_slices; slice += n_waves) {
        // MM2S_A: the whole 8 KiB slice in flight at once, 16 B per lane per load
        const uint4* p = words + slice * (kSliceElems / 2) + lane;
        uint4 w[kSliceSteps];
#pragma unroll
        for (int j = 0; j < kSliceSteps; ++j) w[j] = p[j * 64];
        const int4 h = hdr[slice];
        int row = __builtin_amdgcn_readfirstlane(h.x);
        const int row_first = row;                                   // first row that ends in this slice
        const int chain_len = __builtin_amdgcn_readfirstlane(h.y);   // >0: that row began chain_len slices earlier

        // ComputeAB: val * x[col]
        float p0[kSliceSteps], p1[kSliceSteps];
        if (USE_LDS && in_lds) {
#pragma unroll
            for (int j = 0; j < kSliceSteps; ++j) {
                p0[j] = i2f((int)w[j].x) * xs[(int)(w[j].y & ~kRowEndBit) - x_base];
                p1[j] = i2f((int)w[j].z) * xs[(int)(w[j].w & ~kRowEndBit) - x_base];
            }
        } else {
            float x0[kSliceSteps], x1[kSliceSteps];
#pragma unroll
            for (int j = 0; j < kSliceSteps; ++j) {
                x0[j] = x[w[j].y & ~kRowEndBit];
                x1[j] = x[w[j].w & ~kRowEndBit];
            }
#pragma unroll
            for (int j = 0; j < kSliceSteps; ++j) {
                p0[j] = i2f((int)w[j].x) * x0[j];
                p1[j] = i2f((int)w[j].z) * x1[j];
            }
        }

        // PreAccumulator + row distribution network: segmented scan per 128-element step
        float t0[kSliceSteps], t1[kSliceSteps];
        int r0[kSliceSteps];
        float carry_step = 0.0f;       // partial sum of the row left open by the previous step
#pragma unroll
        for (int j = 0; j < kSliceSteps; ++j) {
            const bool e0 = (w[j].y & kRowEndBit) != 0;
            const bool e1 = (w[j].w & kRowEndBit) != 0;
            // what this lane hands to its right neighbour, and whether it cuts the chain
            float v = e1 ? 0.0f : (e0 ? p1[j] : p0[j] + p1[j]);
            int F = (e0 | e1) ? 1 : 0;
            seg_scan_wave(v, F);
            v = F ? v : v + carry_step;
            // incoming partial for this lane = inclusive value of the lane below (lane 0: previous step)
            const float cin = i2f(__builtin_amdgcn_update_dpp(f2i(carry_step), f2i(v), 0x138, 0xf, 0xf, false));  // wave_shr:1
            carry_step = i2f(__builtin_amdgcn_readlane(f2i(v), 63));
            const unsigned long long m0 = __builtin_amdgcn_ballot_w64(e0);
            const unsigned long long m1 = __builtin_amdgcn_ballot_w64(e1);
            r0[j] = row + lanes_below(m0) + lanes_below(m1);
            t0[j] = cin + p0[j];
            t1[j] = e0 ? p1[j] : cin + (p0[j] + p1[j]);
            row += __builtin_popcountll(m0) + __builtin_popcountll(m1);
        }

        float chain = 0.0f;
        if (LOOKBACK) {
            // publish this slice's open partial sum as ONE 8-byte {value, launch tag} granule ...
            if (lane == 0)
                __hip_atomic_store(lb.gran + slice, ((unsigned long long)lb.epoch << 32) | (unsigned)f2i(carry_step),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // ... and collect the parts of the first row that earlier slices hold (rows "shared" between
            // wavefronts): lane k polls the granule of slice (s - chain_len + k); fixed summation order.
            if (chain_len > 0) {
                float part = 0.0f;
                for (int k = lane; k < chain_len; k += 64) {
                    const unsigned long long* g = lb.gran + (slice - chain_len + k);
                    unsigned long long v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    int spins = 0;
                    while ((unsigned)(v >> 32) != lb.epoch && spins < kLookbackSpinMax) {
                        __builtin_amdgcn_s_sleep(2);
                        v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ++spins;
                    }
                    if ((unsigned)(v >> 32) != lb.epoch) *lb.err = 1;   // bounded wait: report, never hang
                    part += i2f((int)(unsigned)v);
                }
                chain = wave_sum(part);
            }
        }

        // Compute_C: beta*c_in + alpha*acc for the rows that end in this slice
        if (LOOKBACK && chain_len > 0) {
#pragma unroll
            for (int j = 0; j < kSliceSteps; ++j) {
                const bool e0 = (w[j].y & kRowEndBit) != 0;
                if (e0 && r0[j] == row_first) t0[j] += chain;
                else if (!e0 && (w[j].w & kRowEndBit) != 0 && r0[j] == row_first) t1[j] += chain;
            }
        }
        if (HAS_BETA) {
            float b0[kSliceSteps], b1[kSliceSteps];
#pragma unroll
            for (int j = 0; j < kSliceSteps; ++j) {
                const bool e0 = (w[j].y & kRowEndBit) != 0, e1 = (w[j].w & kRowEndBit) != 0;
                b0[j] = e0 ? bias[r0[j]] : 0.0f;
                b1[j] = e1 ? bias[r0[j] + (e0 ? 1 : 0)] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < kSliceSteps; ++j) {
                const bool e0 = (w[j].y & kRowEndBit) != 0, e1 = (w[j].w & kRowEndBit) != 0;
                if (e0) y[r0[j]] = alpha * t0[j] + beta * b0[j];
                if (e1) y[r0[j] + (e0 ? 1 : 0)] = alpha * t1[j] + beta * b1[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < kSliceSteps; ++j) {
                const bool e0 = (w[j].y & kRowEndBit) != 0, e1 = (w[j].w & kRowEndBit) != 0;
                if (e0) y[r0[j]] = alpha * t0[j];
                if (e1) y[r0[j] + (e0 ? 1 : 0)] = alpha * t1[j];
            }
        }
        if (!LOOKBACK && lane == 0) carry[slice] = carry_step;
    }
}

// Fix-up for rows shared between slices: y[row] += alpha * (carry[first] + ... + carry[first+len-1]),
// summed in slice order.  One thread per entry (short chains) ...
__global__ __launch_bounds__(256) void spmv_fixup_short_kernel(const int4* __restrict__ fix, int n,
                                                               const float* __restrict__ carry,
                                                               float* __restrict__ y, float alpha) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 f = fix[i];
    float s = 0.0f;
    for (int k = 0; k < f.z; ++k) s += carry[f.y + k];
    y[f.x] += alpha * s;
}
// ... or one wavefront per entry (a heavy row spanning many slices).
__global__ __launch_bounds__(256) void spmv_fixup_long_kernel(const int4* __restrict__ fix, int n,
                                                              const float* __restrict__ carry,
                                                              float* __restrict__ y, float alpha) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    const int4 f = fix[i];
    float s = 0.0f;
    for (int k = lane; k < f.z; k += 64) s += carry[f.y + k];
    s = wave_sum(s);
    if (lane == 0) y[f.x] += alpha * s;
}

template <bool HAS_BETA, bool USE_LDS, bool LOOKBACK>
static void launch_slices(const SpmvDeviceMatrix& m, const LookbackArgs& lb, const float* x, const float* bias, float* y,
                          float alpha, float beta, hipStream_t stream) {
    const size_t lds = USE_LDS ? (size_t)m.lds_floats * sizeof(float) : 0;
    hipLaunchKernelGGL((spmv_slices_kernel<HAS_BETA, USE_LDS, LOOKBACK>), dim3((unsigned)m.n_groups), dim3(m.block_threads), lds, stream,
                       (const uint4*)m.words, m.hdr, m.groups, x, bias, y, m.carry, alpha, beta,
                       (long long)m.n_slices, m.group_slices, m.lds_floats, m.cols, lb);
}

template <bool HAS_BETA, bool USE_LDS>
static void launch_slices2(const SpmvDeviceMatrix& m, const LookbackArgs& lb, const float* x, const float* bias, float* y,
                           float alpha, float beta, hipStream_t stream) {
    if (m.lookback) launch_slices<HAS_BETA, USE_LDS, true>(m, lb, x, bias, y, alpha, beta, stream);
    else launch_slices<HAS_BETA, USE_LDS, false>(m, lb, x, bias, y, alpha, beta, stream);
}

hipError_t prepare_spmv_kernels() {
    // the x window may use (almost) the whole 160 KiB LDS of a CU
    hipError_t e;
    const int max_lds = kMaxLdsFloats * (int)sizeof(float);
    if ((e = hipFuncSetAttribute((const void*)spmv_slices_kernel<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)spmv_slices_kernel<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)spmv_slices_kernel<true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)spmv_slices_kernel<false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds)) != hipSuccess) return e;
    return hipSuccess;
}

hipError_t launch_spmv(SpmvDeviceMatrix& m, const float* x, const float* bias, float* y,
                       float alpha, float beta, hipStream_t stream) {
    if (m.n_slices > 0) {
        if (m.n_groups <= 0 || m.n_groups > 0x7fffffffLL) return hipErrorInvalidValue;
        LookbackArgs lb{};
        if (m.lookback) {
            lb.gran = m.gran; lb.ticket = m.ticket; lb.err = m.err;
            lb.ticket_base = m.launches * (unsigned long long)m.n_groups;
            lb.epoch = (unsigned)(m.launches % 0xffffffffull) + 1u;   // never 0: tag 0 means "never written"
            m.launches++;
        }
        const bool lds = m.lds_floats > 0;
        if (beta != 0.0f) { if (lds) launch_slices2<true, true>(m, lb, x, bias, y, alpha, beta, stream); else launch_slices2<true, false>(m, lb, x, bias, y, alpha, beta, stream); }
        else              { if (lds) launch_slices2<false, true>(m, lb, x, bias, y, alpha, beta, stream); else launch_slices2<false, false>(m, lb, x, bias, y, alpha, beta, stream); }
        if (m.lookback) return hipGetLastError();
    }
    if (m.n_fix_short > 0)
        hipLaunchKernelGGL(spmv_fixup_short_kernel, dim3((m.n_fix_short + 255) / 256), dim3(256), 0, stream,
                           m.fix_short, m.n_fix_short, m.carry, y, alpha);
    if (m.n_fix_long > 0)
        hipLaunchKernelGGL(spmv_fixup_long_kernel, dim3((m.n_fix_long + 3) / 4), dim3(256), 0, stream,
                           m.fix_long, m.n_fix_long, m.carry, y, alpha);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Dense overlay (reference: ComputeAB dense branch base_functions.cpp:188-226, packing
// spmv-helper.cpp:717-750).  No packing here: W stays row-major; one workgroup of 4 waves
// per group of R rows, 16 B per lane loads of W and x, DPP wave reduction, LDS across waves.
// ---------------------------------------------------------------------------
template <int R, bool HAS_BETA>
__global__ __launch_bounds__(256) void gemv_rows_kernel(const float* __restrict__ W, const float* __restrict__ x,
                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                        int rows, int cols, float alpha, float beta) {
    __shared__ float part[4][R];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int row0 = blockIdx.x * R;
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0f;

    if ((cols & 3) == 0) {
        const int n4 = cols >> 2;
        const float4* x4 = (const float4*)x;
        for (int c = threadIdx.x; c < n4; c += 256) {
            const float4 xv = x4[c];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int row = min(row0 + r, rows - 1);
                const float4 a = ((const float4*)(W + (size_t)row * cols))[c];
                acc[r] += a.x * xv.x + a.y * xv.y + a.z * xv.z + a.w * xv.w;
            }
        }
    } else {   // rows are not 16-byte aligned: dword path
        for (int c = threadIdx.x; c < cols; c += 256) {
            const float xv = x[c];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int row = min(row0 + r, rows - 1);
                acc[r] += W[(size_t)row * cols + c] * xv;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float s = wave_sum(acc[r]);
        if (lane == 0) part[wv][r] = s;
    }
    __syncthreads();
    if (threadIdx.x < R) {
        const int row = row0 + threadIdx.x;
        if (row < rows) {
            const float s = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
            y[row] = HAS_BETA ? alpha * s + beta * bias[row] : alpha * s;
        }
    }
}

hipError_t launch_gemv(const float* W, int32_t rows, int32_t cols, const float* x, const float* bias,
                       float* y, float alpha, float beta, hipStream_t stream) {
    if (rows <= 0) return hipSuccess;
    constexpr int R = 4;
    const unsigned blocks = (unsigned)((rows + R - 1) / R);
    if (beta != 0.0f)
        hipLaunchKernelGGL((gemv_rows_kernel<R, true>), dim3(blocks), dim3(256), 0, stream, W, x, bias, y, rows, cols, alpha, beta);
    else
        hipLaunchKernelGGL((gemv_rows_kernel<R, false>), dim3(blocks), dim3(256), 0, stream, W, x, bias, y, rows, cols, alpha, beta);
    return hipGetLastError();
}

}  // namespace hispmv
