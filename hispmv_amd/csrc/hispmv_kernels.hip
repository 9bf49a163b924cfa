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

__device__ __forceinline__ int lanes_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

// ---------------------------------------------------------------------------
// Slice kernel: one wavefront per slice of kSliceElems elements.
// ---------------------------------------------------------------------------
template <bool HAS_BETA>
__global__ __launch_bounds__(256) void spmv_slices_kernel(
    const uint4* __restrict__ words, const int4* __restrict__ hdr, const float* __restrict__ x,
    const float* __restrict__ bias, float* __restrict__ y, float* __restrict__ carry,
    float alpha, float beta, long long n_slices) {
    const int lane = threadIdx.x & 63;
    const long long slice = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (slice >= n_slices) return;   // wave-uniform

    // MM2S_A: the whole 8 KiB slice in flight at once, 16 B per lane per load
    const uint4* p = words + slice * (kSliceElems / 2) + lane;
    uint4 w[kSliceSteps];
#pragma unroll
    for (int j = 0; j < kSliceSteps; ++j) w[j] = p[j * 64];

    int row = __builtin_amdgcn_readfirstlane(hdr[slice].x);
    float carry_step = 0.0f;       // partial sum of the row left open by the previous step

#pragma unroll
    for (int j = 0; j < kSliceSteps; ++j) {
        const float x0 = x[w[j].y & ~kRowEndBit];
        const float x1 = x[w[j].w & ~kRowEndBit];
        const float p0 = i2f((int)w[j].x) * x0;
        const float p1 = i2f((int)w[j].z) * x1;
        const bool e0 = (w[j].y & kRowEndBit) != 0;
        const bool e1 = (w[j].w & kRowEndBit) != 0;

        // what this lane hands to its right neighbour, and whether it cuts the chain
        float v = e1 ? 0.0f : (e0 ? p1 : p0 + p1);
        int F = (e0 | e1) ? 1 : 0;
        seg_scan_wave(v, F);
        v = F ? v : v + carry_step;
        // incoming partial for this lane = inclusive value of the lane below (lane 0: previous step)
        const float cin = i2f(__builtin_amdgcn_update_dpp(f2i(carry_step), f2i(v), 0x138, 0xf, 0xf, false));  // wave_shr:1
        carry_step = i2f(__builtin_amdgcn_readlane(f2i(v), 63));

        const unsigned long long m0 = __builtin_amdgcn_ballot_w64(e0);
        const unsigned long long m1 = __builtin_amdgcn_ballot_w64(e1);
        if ((m0 | m1) != 0ull) {   // wave-uniform: at least one row ends in this step
            const int r0 = row + lanes_below(m0) + lanes_below(m1);
            const int r1 = r0 + (e0 ? 1 : 0);
            const float t0 = cin + p0;
            const float t1 = e0 ? p1 : cin + (p0 + p1);
            if (e0) y[r0] = HAS_BETA ? alpha * t0 + beta * bias[r0] : alpha * t0;
            if (e1) y[r1] = HAS_BETA ? alpha * t1 + beta * bias[r1] : alpha * t1;
            row += __builtin_popcountll(m0) + __builtin_popcountll(m1);
        }
    }
    if (lane == 0) carry[slice] = carry_step;
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

hipError_t launch_spmv(const SpmvDeviceMatrix& m, const float* x, const float* bias, float* y,
                       float alpha, float beta, hipStream_t stream) {
    if (m.n_slices > 0) {
        const long long blocks = (m.n_slices + 3) / 4;
        if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
        if (beta != 0.0f)
            hipLaunchKernelGGL(spmv_slices_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream,
                               (const uint4*)m.words, m.hdr, x, bias, y, m.carry, alpha, beta, (long long)m.n_slices);
        else
            hipLaunchKernelGGL(spmv_slices_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream,
                               (const uint4*)m.words, m.hdr, x, bias, y, m.carry, alpha, beta, (long long)m.n_slices);
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
