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
//                                                      (bitwise reproducible, no float atomics); small
//                                                      co-resident launches merge them in-kernel (look-back)
// Variants: NV vectors per pass (batched linear), several matrices per launch (hispmv_spmv_device_batch).
// Bandwidth-bound gather: no MFMA.  Roofline and byte accounting: DESIGN.md.
#include <hip/hip_runtime.h>

#include "hispmv_format.h"
#include "hispmv_kernels.h"
#include <algorithm>
#include <cstdlib>
#include <vector>

#ifndef HISPMV_TTS_EXPERIMENT
#define HISPMV_TTS_EXPERIMENT 0
#endif

namespace hispmv {

// ---------------------------------------------------------------------------
// Workgroup trace (diagnostic builds only: make WGTRACE=1 -> libhispmv_wgtrace.so; the product library carries none of it).
// Every workgroup of the multi-matrix kernels records {start, end} on the 100 MHz constant clock, the CU it ran on
// (XCC_ID, HW_ID) and what it was (kernel kind, table entry, group / tile): tools/wg_timeline.py turns the records of one
// step into per-CU occupancy -- how much of a step a CU spends inside workgroups, and in which kind.
// ---------------------------------------------------------------------------
#ifdef HISPMV_WG_TRACE
__device__ unsigned long long* g_wgt_buf = nullptr;     // [0] = record counter, records of 4 x u64 from [4]
__device__ unsigned g_wgt_cap = 0;
struct WgTrace {
    unsigned long long t0;
    __device__ __forceinline__ void begin() { t0 = __builtin_amdgcn_s_memrealtime(); }
    __device__ __forceinline__ void end(int kind, int entry, long long item) {
        __syncthreads();
        if (threadIdx.x == 0 && g_wgt_buf) {
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            const unsigned long long i = __hip_atomic_fetch_add(g_wgt_buf, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (i < g_wgt_cap) {
                unsigned long long* r = g_wgt_buf + 4 + 4 * i;
                r[0] = t0; r[1] = t1;
                r[2] = ((unsigned long long)xcc << 32) | hw;
                r[3] = ((unsigned long long)kind << 56) | ((unsigned long long)(entry & 0xffff) << 40) | (unsigned long long)(item & 0xffffffffffll);
            }
        }
    }
};
#define WGT_BEGIN() WgTrace wgt; wgt.begin()
#define WGT_END(kind, entry, item) wgt.end(kind, entry, item)
#else
#define WGT_BEGIN()
#define WGT_END(kind, entry, item)
#endif

// ---------------------------------------------------------------------------
// wave64 helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ int f2i(float f) { return __builtin_bit_cast(int, f); }
__device__ __forceinline__ float i2f(int i) { return __builtin_bit_cast(float, i); }

// Segmented inclusive scan over the 64 lanes, Kogge-Stone on (head flag F, value v):
//   (F1,v1) o (F2,v2) = (F1|F2, F2 ? v2 : v1+v2)
// CTRL/ROWMASK select the DPP source; lanes without a source read the identity (0,0).
// The flags never depend on the values: the F of every stage is computed from the ballot of the head flags on the
// SCALAR unit (shifts inside the 16-lane DPP rows, then the two row broadcasts) and handed to the select through
// inverse_ballot -- per stage one v_add_f32_dpp and one v_cndmask instead of those plus a v_or_b32_dpp and a v_cmp
// (the slice kernel is VALU-bound on gfx950: 88-119 VALU instructions per 256-element step before, 51-66 after).
struct ScanFlags { unsigned long long f[7]; };    // F before stage 0..5, and after the last one
__device__ __forceinline__ ScanFlags scan_flags(unsigned long long heads) {
    ScanFlags s;
    s.f[0] = heads;
    s.f[1] = s.f[0] | ((s.f[0] << 1) & 0xfffefffefffefffeull);     // row_shr:1
    s.f[2] = s.f[1] | ((s.f[1] << 2) & 0xfffcfffcfffcfffcull);     // row_shr:2
    s.f[3] = s.f[2] | ((s.f[2] << 4) & 0xfff0fff0fff0fff0ull);     // row_shr:4
    s.f[4] = s.f[3] | ((s.f[3] << 8) & 0xff00ff00ff00ff00ull);     // row_shr:8
    s.f[5] = s.f[4] | (((s.f[4] >> 15) & 1ull) ? 0x00000000ffff0000ull : 0ull)     // row_bcast:15 -> rows 1,3
                    | (((s.f[4] >> 47) & 1ull) ? 0xffff000000000000ull : 0ull);
    s.f[6] = s.f[5] | (((s.f[5] >> 31) & 1ull) ? 0xffffffff00000000ull : 0ull);    // row_bcast:31 -> rows 2,3
    return s;
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ void seg_scan_step(float& v, unsigned long long F) {
    const float vp = i2f(__builtin_amdgcn_update_dpp(0, f2i(v), CTRL, ROWMASK, 0xf, false));
    v = __builtin_amdgcn_inverse_ballot_w64(F) ? v : v + vp;
}
__device__ __forceinline__ void seg_scan_wave(float& v, const ScanFlags& s) {
    seg_scan_step<0x111, 0xf>(v, s.f[0]);   // row_shr:1
    seg_scan_step<0x112, 0xf>(v, s.f[1]);   // row_shr:2
    seg_scan_step<0x114, 0xf>(v, s.f[2]);   // row_shr:4
    seg_scan_step<0x118, 0xf>(v, s.f[3]);   // row_shr:8
    seg_scan_step<0x142, 0xa>(v, s.f[4]);   // row_bcast:15 -> rows 1,3
    seg_scan_step<0x143, 0xc>(v, s.f[5]);   // row_bcast:31 -> rows 2,3
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

constexpr int kTtsChunkSlots = 1024;   // hispmv_tts.h: kTtsChunk
constexpr int kLookbackSpinMax = 1 << 22;   // ~seconds; a wait this long means a lost launch, not contention

__device__ __forceinline__ int lanes_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

// Sum of the carries the `chain_len` slices before `cur` published for this launch (lane k polls granule
// cur - chain_len + k, k += 64; fixed summation order).  Bounded wait: an expired poll sets *lb.err.
// Slices of this workgroup (index >= first) are read from the group's LDS mailbox `local` (a neighbour
// wavefront published them a moment ago: ~100 cycles instead of an L2-bypassing round trip per slice), slices
// of earlier workgroups from the global granules.
// The group's mailbox lives in LDS: an explicit address-space-3 pointer with relaxed workgroup-scope atomics compiles
// to ds_write_b64 / ds_read_b64.  (A `volatile unsigned long long*` became FLAT accesses with sc0 sc1 and an
// s_waitcnt vmcnt(0) after every store -- which also waited for the next slice's prefetch: +15 us on PFlow_742.)
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
__device__ __forceinline__ void mbox_store(lds_u64* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ unsigned long long mbox_load(lds_u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ float lookback_chain(const LookbackArgs& lb, lds_u64* local,
                                                long long first, long long cur, int chain_len, int lane) {
    float part = 0.0f;
    for (int k = lane; k < chain_len; k += 64) {
        const long long src = cur - chain_len + k;
        unsigned long long v;
        int spins = 0;
        if (src >= first) {
            v = mbox_load(local + (src - first));
            while ((unsigned)(v >> 32) != lb.epoch && spins < kLookbackSpinMax) {
                __builtin_amdgcn_s_sleep(1);
                v = mbox_load(local + (src - first));
                ++spins;
            }
        } else {
            const unsigned long long* g = lb.gran + src;
            v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while ((unsigned)(v >> 32) != lb.epoch && spins < kLookbackSpinMax) {
                __builtin_amdgcn_s_sleep(2);
                v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ++spins;
            }
        }
        if ((unsigned)(v >> 32) != lb.epoch) *lb.err = 1;   // report, never hang
        part += i2f((int)(unsigned)v);
    }
    return wave_sum(part);
}

// ---------------------------------------------------------------------------
// Slice kernel.  One workgroup owns a GROUP of consecutive slices (the analogue of a PE group fed by
// one x window, LoadB base_functions.cpp:105-150); its wavefronts take the group's slices round-robin.
//   group, once: the group's x FRAGMENTS (runs of the 64-byte blocks of x its slices touch, hispmv_plan.h) are
//            staged into LDS with coalesced 16-byte loads; the metas of such a group index that window.  A group
//            without fragments gathers x through L2; single elements outside the window do too (kGlobalColBit).
//   slice:   structure of arrays, 4 steps of 256 elements, 4 consecutive elements per lane and step: per step one
//            global_load_dwordx4 (values) + one dwordx2 (compact 16-bit metas: 6 bytes per element) or dwordx4 (wide
//            metas), non-temporal, requested one iteration ahead; row ids from ballots; 16 gathers (ds_read_b32 or
//            buffer loads) + the bias of the slice's rows; products; next slice requested; 4 steps of lane-local
//            combine + DPP segmented scan (totals in registers); row totals -> this wavefront's LDS tile -> coalesced
//            y = alpha*total + beta*bias stores; the partial sum left open goes to carry[slice] (fix-up launch) or to
//            the look-back mailbox/granule.
// ---------------------------------------------------------------------------
// One 16-byte / 8-byte piece of the packed stream.  The stream is read exactly once per launch: the non-temporal
// hint keeps it from evicting x, y and the fragment tables from L2 (measured: -5..8 % kernel time on
// PFlow_742, soc-Pokec and mouse_gene).
// Every pointer is cast to the GLOBAL address space first: the multi-matrix kernel takes its pointers from a table in
// memory, where hipcc cannot tell global from LDS and would emit FLAT loads -- those count on lgkmcnt as well and return
// out of order, so every wait for an LDS gather would also wait for the prefetch of the next slice.
#define HISPMV_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ uint4 load_words(const uint4* p) {
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    const u4v v = __builtin_nontemporal_load((const HISPMV_GLOBAL u4v*)p);
    return uint4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ uint2 load_words2(const uint2* p) {
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    const u2v v = __builtin_nontemporal_load((const HISPMV_GLOBAL u2v*)p);
    return uint2{v.x, v.y};
}
__device__ __forceinline__ int4 load_int4(const int4* p) {
    typedef int i4v __attribute__((ext_vector_type(4)));
    const i4v v = *(const HISPMV_GLOBAL i4v*)p;
    return int4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ float4 load_float4(const float4* p) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v v = *(const HISPMV_GLOBAL f4v*)p;
    return float4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void store_float(float* p, float v) { *(HISPMV_GLOBAL float*)p = v; }
__device__ __forceinline__ int2 load_int2(const int2* p) {
    typedef int i2v __attribute__((ext_vector_type(2)));
    const i2v v = *(const HISPMV_GLOBAL i2v*)p;
    return int2{v.x, v.y};
}

// A slice as it arrives: per step 4 values and 4 metas per lane (compact: 4 x u16 in a uint2; wide: 4 x u32).
template <bool COMPACT> struct SliceRaw;
template <> struct SliceRaw<true>  { uint4 v[kSliceSteps]; uint2 m[kSliceSteps]; };
template <> struct SliceRaw<false> { uint4 v[kSliceSteps]; uint4 m[kSliceSteps]; };
template <bool COMPACT>
__device__ __forceinline__ void request_slice(SliceRaw<COMPACT>& s, const char* base, int lane) {
    const uint4* pv = (const uint4*)base + lane;
#pragma unroll
    for (int j = 0; j < kSliceSteps; ++j) s.v[j] = load_words(pv + j * 64);
    if constexpr (COMPACT) {
        const uint2* pm = (const uint2*)(base + kSliceElems * 4) + lane;
#pragma unroll
        for (int j = 0; j < kSliceSteps; ++j) s.m[j] = load_words2(pm + j * 64);
    } else {
        const uint4* pm = (const uint4*)(base + kSliceElems * 4) + lane;
#pragma unroll
        for (int j = 0; j < kSliceSteps; ++j) s.m[j] = load_words(pm + j * 64);
    }
}
// -> metas in wide form (rowEnd<<31 | window index or column), c[4*j + k] = element k of the lane in step j
template <bool COMPACT>
__device__ __forceinline__ void decode_metas(const SliceRaw<COMPACT>& s, unsigned (&c)[kSliceSteps * kLaneElems]) {
    if constexpr (COMPACT) {
#pragma unroll
        for (int j = 0; j < kSliceSteps; ++j) {
            const unsigned a = s.m[j].x, b = s.m[j].y;
            c[4 * j + 0] = ((a & 0x8000u) << 16) | (a & 0x7fffu);
            c[4 * j + 1] = (a & 0x80000000u) | ((a >> 16) & 0x7fffu);
            c[4 * j + 2] = ((b & 0x8000u) << 16) | (b & 0x7fffu);
            c[4 * j + 3] = (b & 0x80000000u) | ((b >> 16) & 0x7fffu);
        }
    } else {
#pragma unroll
        for (int j = 0; j < kSliceSteps; ++j) { c[4 * j + 0] = s.m[j].x; c[4 * j + 1] = s.m[j].y; c[4 * j + 2] = s.m[j].z; c[4 * j + 3] = s.m[j].w; }
    }
}

// PreAccumulator + row distribution network for one step of 256 elements: each lane combines its 4 products left to
// right inside row segments, the open tails go through the DPP segmented scan, and every row end gets its total.
//   p[k], e[k]: products and row-end flags of the lane's elements; carry_step: partial of the row left open by the
//   previous step (in/out); t[k]: total of the row that ends at element k (meaningful where e[k]).
__device__ __forceinline__ void scan_step(const float (&p)[kLaneElems], bool e0, bool e1, bool e2, bool e3, float& carry_step,
                                          float (&t)[kLaneElems]) {
    const float s0 = p[0];
    const float s1 = e0 ? p[1] : s0 + p[1];
    const float s2 = e1 ? p[2] : s1 + p[2];
    const float s3 = e2 ? p[3] : s2 + p[3];
    // what this lane hands to its right neighbour, and whether it cuts the chain
    float v = e3 ? 0.0f : s3;
    const ScanFlags F = scan_flags(__builtin_amdgcn_ballot_w64(e0 | e1 | e2 | e3));
    seg_scan_wave(v, F);
    v = __builtin_amdgcn_inverse_ballot_w64(F.f[6]) ? v : v + carry_step;
    // incoming partial for this lane = inclusive value of the lane below (lane 0: previous step)
    const float cin = i2f(__builtin_amdgcn_update_dpp(f2i(carry_step), f2i(v), 0x138, 0xf, 0xf, false));  // wave_shr:1
    carry_step = i2f(__builtin_amdgcn_readlane(f2i(v), 63));
    t[0] = cin + s0;
    t[1] = e0 ? s1 : cin + s1;
    t[2] = (e0 | e1) ? s2 : cin + s2;
    t[3] = (e0 | e1 | e2) ? s3 : cin + s3;
}

// LoadB: the x fragments {col_start, len, lds_off} of a group into its LDS window.  A wavefront takes every n_waves-th
// fragment; its copies leave 8 at a time (512 float4 per round) and the next fragment's table entry travels with them.
// (One float4 per iteration -- load, wait, ds_write -- made a 2048-float fragment eight memory round trips, and every
// workgroup starts with this phase: PFlow_742 17 fragments on 16 wavefronts, ~10 us of a 48 us kernel.)
// The float4s that reach past the end of x (at most the last 64-byte block) are read element-wise, zero beyond `cols`.
__device__ __forceinline__ void stage_fragments(const int4* __restrict__ frags, int first, int count, const float* x, float* win,
                                                int cols, int lane, int wave, int n_waves) {
    constexpr int kU = 8;
    int4 fr = wave < count ? load_int4(frags + first + wave) : int4{0, 0, 0, 0};
    for (int f = wave; f < count; f += n_waves) {
        const int col0 = __builtin_amdgcn_readfirstlane(fr.x), n4 = __builtin_amdgcn_readfirstlane(fr.y) >> 2;
        const int off = __builtin_amdgcn_readfirstlane(fr.z);
        if (f + n_waves < count) fr = load_int4(frags + first + f + n_waves);
        const float4* src = (const float4*)(x + col0);
        float4* dst = (float4*)(win + off);
        const int inside = cols - col0 >= 4 * n4 ? n4 : (cols > col0 ? (cols - col0) >> 2 : 0);     // float4s entirely inside x
        // rounds of 8, 2 or 1 wave-wide copies, by what is left of the fragment (wave-uniform: no divergence; short
        // fragments -- one or two 64-byte blocks -- must not pay for eight loads)
        int i0 = 0;
        for (; inside - i0 > 128; i0 += 64 * kU) {
            float4 v[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {          // (unconditional, index clamped: a predicate per load becomes a branch per load
                const int i = i0 + 64 * u + lane;   //  and hipcc then waits for the loads before it)
                v[u] = load_float4(src + (i < inside ? i : inside - 1));
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int i = i0 + 64 * u + lane;
                if (i < inside) dst[i] = v[u];
            }
        }
        if (inside - i0 > 64) {
            const int ia = i0 + lane, ib = i0 + 64 + lane;
            const float4 va = load_float4(src + ia), vb = load_float4(src + (ib < inside ? ib : inside - 1));
            dst[ia] = va;
            if (ib < inside) dst[ib] = vb;
        } else if (inside - i0 > 0) {
            const int ia = i0 + lane;
            if (ia < inside) dst[ia] = load_float4(src + ia);
        }
        if (inside + lane < n4) {
            const int i = inside + lane, c0 = col0 + 4 * i;
            const HISPMV_GLOBAL float* xg = (const HISPMV_GLOBAL float*)x;
            const float a0 = c0 + 0 < cols ? xg[c0 + 0] : 0.f;
            const float a1 = c0 + 1 < cols ? xg[c0 + 1] : 0.f;
            const float a2 = c0 + 2 < cols ? xg[c0 + 2] : 0.f;
            dst[i] = float4{a0, a1, a2, 0.f};
        }
    }
}

// Which wavefronts of the workgroup work on a group, and where its LDS begins: the whole workgroup for the slice kernels; in
// the step kernel (below) a 1024-thread workgroup hosts FOUR groups of a 256-thread plan side by side, four wavefronts each.
struct SubBlock {
    int wave, n_waves, lds_off;       // this wavefront's index among the group's wavefronts, their number, first LDS float of the group
    int lane;                         // (handed in: the step kernel makes the thread index opaque per item, see there)
    static __device__ __forceinline__ SubBlock whole() { return SubBlock{(int)(threadIdx.x >> 6), (int)(blockDim.x >> 6), 0, (int)(threadIdx.x & 63)}; }
};

// The work of one workgroup on group `group` of a matrix (the body of the slice kernels below), for a group stored
// COMPACT (6 B per element) or wide (8 B): two instantiations, chosen per group by slices_body.
template <bool HAS_BETA, bool USE_LDS, bool LOOKBACK, bool COMPACT, bool STRAYS = false>
__device__ __forceinline__ void slices_group(
    const char* __restrict__ stream, const int4* __restrict__ hdr, const int4* __restrict__ groups,
    const int4* __restrict__ frags,
    const float* __restrict__ x, const float* bias, float* y,   // bias may alias y
    float* __restrict__ carry, float alpha, float beta, long long n_slices, int group_slices,
    int lds_floats, int ytile_floats, int cols, int rows, const LookbackArgs& lb, long long group, int4 g, const SubBlock sb) {
    // LDS: [x window: lds_floats][row totals of the slice in flight: ytile_floats per wavefront]
    extern __shared__ float xs_base[];
    float* const xs = xs_base + sb.lds_off;
    // x, bias and y are reached through buffer descriptors: 32-bit byte offsets instead of 64-bit
    // addresses (half the address VGPRs, one shift per gather), and the hardware range check turns an
    // offset of 0xffffffff into "no access" -- the predicate of the bias loads and y stores costs no branch,
    // so hipcc counts every load exactly and the prefetch below really stays in flight.
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, cols * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)bias, 0, rows * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, rows * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)carry, 0, LOOKBACK ? 0 : (int)(n_slices * 4), 0x00020000);
    constexpr unsigned kNoAccess = 0xffffffffu;
    constexpr int kE = kSliceSteps * kLaneElems;     // elements of a slice per lane (16)
    const int lane = sb.lane;
    const int wave = sb.wave, n_waves = sb.n_waves;
    float* const ytile = xs + (USE_LDS ? lds_floats : 0) + wave * ytile_floats;
    // look-back mailbox of the group: one {carry, launch tag} per slice, published by the wavefront that owns it
    lds_u64* const mbox = (lds_u64*)(xs + (USE_LDS ? lds_floats : 0) + n_waves * ytile_floats);
    const long long first = group * group_slices;
    const long long last = (first + group_slices < n_slices) ? first + group_slices : n_slices;   // exclusive
    // where the group's slices lie in the stream: the group table says so for staged plans; a plan without windows is
    // all wide, slice after slice
    constexpr int slice_bytes = COMPACT ? kCompactSliceBytes : kWideSliceBytes;
    const char* const gbase = stream + (USE_LDS ? (size_t)(unsigned)__builtin_amdgcn_readfirstlane(g.z) * kSliceUnit
                                                : (size_t)first * kWideSliceBytes);

    // MM2S_A, software-pipelined: the slice a wavefront works on was requested one iteration earlier
    // (the whole slice, plus its header), so the HBM latency of slice k+1 hides behind the gathers, scans
    // and stores of slice k.  The first request goes out BEFORE the x window is staged.
    // Each workgroup walks its group from a different starting slice (rotation by a multiple of its id): workgroups
    // that all march through their groups from slice 0 at the same pace read addresses one group stride apart at
    // every moment, and some strides alias in the HBM channel hash -- PFlow_742 with 145 slices per workgroup ran 72
    // instead of 63 us (142, 146, 149, 152 and 155 slices did not).  Not with look-back: its chains want slice order.
    const int n_here = (int)(last > first ? last - first : 0);
    const int rot = (LOOKBACK || n_here == 0) ? 0 : (int)((unsigned long long)group * 29ull % (unsigned)n_here);
    int k_slice = wave;                                         // position in the group's rotated order
    int local = k_slice < n_here ? (k_slice + rot >= n_here ? k_slice + rot - n_here : k_slice + rot) : n_here;
    SliceRaw<COMPACT> w;
    int4 h = int4{0, 0, 0, 0};
    // STRAY SLOTS (hispmv_plan.h): a compact group whose slices have a few elements outside the window.  The columns of a slice's
    // strays (<= 64, one per lane, 0xffffffff = none) live behind the headers; their x values are gathered ONE SLICE AHEAD and
    // written to this wavefront's 64-float stray area behind the window when the slice's turn comes -- the strays' metas index
    // that area, so the gathers below are plain LDS reads for every element.  Order of the requests: the columns of slice i+2
    // leave BEFORE the words of slice i+1 and the x values of slice i+1's strays BEFORE them too (vmcnt retires in issue order:
    // the wait for a slice's words then covers what that slice's turn needs, and nothing waits for anything younger).
    // (STRAYS is a template parameter: matrices without stray slots run instantiations that carry none of this -- with it in, the
    // multi-matrix kernel went from 91 to 97 VGPRs, four wavefronts per SIMD instead of five, and the step lost 0.4 %)
    const bool strays = STRAYS && COMPACT && (__builtin_amdgcn_readfirstlane(g.w) & 2) != 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(hdr + n_slices), 0, strays ? (int)(n_slices * (kStraySlots * 4)) : 0, 0x00020000);
    float* const stray_area = xs + (lds_floats - n_waves * kStraySlots) + wave * kStraySlots;
    unsigned sc_next = 0xffffffffu;        // stray columns of this wavefront's NEXT slice
    float sx_cur = 0.0f;                   // x values of the CURRENT slice's strays (lane k: stray k)
    if (strays) {
        const unsigned sc0 = local < n_here ? __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)((first + local) * kStraySlots + lane) << 2, 0, 0) : 0xffffffffu;
        const int k2 = k_slice + n_waves;
        const int l2 = k2 < n_here ? (k2 + rot >= n_here ? k2 + rot - n_here : k2 + rot) : -1;
        sc_next = l2 >= 0 ? __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)((first + l2) * kStraySlots + lane) << 2, 0, 0) : 0xffffffffu;
        sx_cur = i2f((int)__builtin_amdgcn_raw_buffer_load_b32(rx, sc0 << 2, 0, 0));        // (0xffffffff << 2 is out of range: no access)
    }
    if (local < n_here) {
        // (the header first: a component of it that an instantiation does not use is a dead register hipcc reuses at
        // once -- a write-after-write hazard with the load in flight, and waiting for the YOUNGEST load waits for the
        // whole prefetch: s_waitcnt vmcnt(0) right behind the request, seen in the compact loop)
        h = load_int4(hdr + first + local);
        request_slice<COMPACT>(w, gbase + (size_t)local * slice_bytes, lane);
    }

    if (LOOKBACK) {
        for (int i = threadIdx.x; i < group_slices; i += blockDim.x) mbox_store(mbox + i, 0ull);   // tag 0 = not published
        if (!USE_LDS) __syncthreads();
    }
    // LoadB: stage the x fragments of this group (runs of 64-byte blocks its slices touch) into LDS; the
    // metas of a staged group already carry the index into this window instead of the column.
    bool in_lds = false;
    if (USE_LDS) {
        in_lds = g.y > 0;                    // workgroup-uniform; 0 fragments = this group gathers through L2
        stage_fragments(frags, g.x, g.y, x, xs, cols, lane, wave, n_waves);
        __syncthreads();
    }

    // Look-back bookkeeping: a chain that reaches back BEFORE this group's first slice is resolved after the
    // loop (at most one row crosses the group boundary, so at most one slice per workgroup is deferred).  Inside
    // the group a slice's predecessors are handled in the same or an earlier round; the previous workgroup's last
    // slice is handled in ITS last round -- waiting for it mid-loop would stall this wavefront for the whole
    // kernel (and, chained through every workgroup, serialise the grid: measured 14x).
    long long def_slice = -1, pend_slice = -1;
    int def_row = 0, def_len = 0, pend_row = 0, pend_len = 0;
    float def_t = 0.0f, def_b = 0.0f, pend_t = 0.0f, pend_b = 0.0f;
    // Fix-up variant: the stores of a slice (its first 128 rows of y and its carry) are held in registers and issued at
    // the top of the wavefront's NEXT iteration, behind the wait for that iteration's slice.  gfx9 counts loads and
    // stores on the same vmcnt and lets them retire out of order with each other, so with a store pending every wait
    // for a load is vmcnt(0): issued at the end of the iteration, the stores made the wait for the prefetched slice a
    // wait for their write acknowledgements as well.  (An offset of kNoAccess = no store.)
    // With beta != 0 the bias values of those rows are consumed there too: nothing in the middle of an iteration waits on
    // vmcnt, where the wait would also cover the prefetch issued a moment earlier.
    float out_t0 = 0.0f, out_t1 = 0.0f, out_b0 = 0.0f, out_b1 = 0.0f, out_carry = 0.0f;
    unsigned out_d0 = kNoAccess, out_d1 = kNoAccess, out_dc = kNoAccess;

    while (local < n_here) {
        // (every component of the header stays live up to here: a component an instantiation does not use is a dead
        // register hipcc hands out as a temporary while the load is in flight -- a write-after-write hazard it covers by
        // waiting for the header, i.e. a full memory latency right behind the prefetch: "s_waitcnt vmcnt(8)" in the first
        // scan block)
        asm volatile("" :: "v"(h.x), "v"(h.y), "v"(h.z), "v"(h.w));
        int row = __builtin_amdgcn_readfirstlane(h.x);
        const int row_first = row;                                   // first row that ends in this slice
        const int chain_len = __builtin_amdgcn_readfirstlane(h.y);   // >0: that row began chain_len slices earlier
        const int n_rows = __builtin_amdgcn_readfirstlane(h.z);      // rows that end in this slice
        const bool spills = USE_LDS && __builtin_amdgcn_readfirstlane(h.w) != 0;   // elements outside the x window (wide groups only)
        // Compute_C operand: the slice's rows are consecutive, so bias is read with coalesced loads that leave
        // together with the x gathers (first 128 rows here, the rest in the epilogue loop).
        unsigned c[kE];
        decode_metas<COMPACT>(w, c);
        if (!LOOKBACK) {
            asm volatile("" : "+v"(c[0]), "+v"(c[kE - 1]) :: "memory");      // behind the wait for this slice
            __builtin_amdgcn_raw_buffer_store_b32((unsigned)f2i(HAS_BETA ? alpha * out_t0 + beta * out_b0 : alpha * out_t0), ry, out_d0, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32((unsigned)f2i(HAS_BETA ? alpha * out_t1 + beta * out_b1 : alpha * out_t1), ry, out_d1, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32((unsigned)f2i(out_carry), rc, out_dc, 0, 0);
        }
        float sx_next = 0.0f;
        if (strays) {
            // this slice's strays -> the wavefront's stray area (LDS operations of a wavefront execute in order: the gathers
            // below see them); the x values of the NEXT slice's strays leave now, a whole iteration before they are needed
            stray_area[lane] = sx_cur;
            sx_next = i2f((int)__builtin_amdgcn_raw_buffer_load_b32(rx, sc_next << 2, 0, 0));
        }
        float bpre0 = 0.0f, bpre1 = 0.0f;
        if (HAS_BETA) {
            bpre0 = i2f((int)__builtin_amdgcn_raw_buffer_load_b32(rb, lane < n_rows ? (unsigned)(row_first + lane) << 2 : kNoAccess, 0, 0));
            bpre1 = i2f((int)__builtin_amdgcn_raw_buffer_load_b32(rb, lane + 64 < n_rows ? (unsigned)(row_first + lane + 64) << 2 : kNoAccess, 0, 0));
        }
        // Local row ids of every row end (ballots + mbcnt prefix counts; no per-element row field): r0[j] = row of
        // the lane's first row end in step j, the lane's further ends follow it.
        int r0[kSliceSteps];
        bool ends[kE];         // element k of step j ends its row (kept as lane masks: no per-lane bit field to build and test)
        unsigned step_has_end = 0;   // wave-uniform: bit j = some row ends in step j
#pragma unroll
        for (int j = 0; j < kSliceSteps; ++j) {
            int below = 0, total = 0;
#pragma unroll
            for (int k = 0; k < kLaneElems; ++k) {
                ends[4 * j + k] = (c[4 * j + k] & kRowEndBit) != 0;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(ends[4 * j + k]);
                below += lanes_below(m);
                total += __builtin_popcountll(m);
            }
            r0[j] = row + below;
            row += total;
            step_has_end |= (total != 0 ? 1u : 0u) << j;
        }

        // LoadB / ComputeAB operands: x[col] (LDS window or L2 gather)
        float xv[kE];
        if (COMPACT || (USE_LDS && in_lds && !spills)) {     // (a compact group has every element in its window)
#pragma unroll
            for (int i = 0; i < kE; ++i) xv[i] = xs[c[i] & ~kRowEndBit];
        } else if (USE_LDS && in_lds) {
            // window of the group's most used blocks: the elements outside it (kGlobalColBit) gather through L2 --
            // lanes inside the window give the buffer load an out-of-range offset (no access, returns 0)
#pragma unroll
            for (int i = 0; i < kE; ++i) {
                const unsigned ci = c[i] & ~kRowEndBit;
                xv[i] = i2f((int)__builtin_amdgcn_raw_buffer_load_b32(rx, (ci & kGlobalColBit) ? ci << 2 : kNoAccess, 0, 0));
            }
#pragma unroll
            for (int i = 0; i < kE; ++i) {
                const unsigned ci = c[i] & ~kRowEndBit;
                const float l = xs[(ci & kGlobalColBit) ? 0u : ci];
                xv[i] = (ci & kGlobalColBit) ? xv[i] : l;
            }
        } else {
#pragma unroll
            for (int i = 0; i < kE; ++i) xv[i] = i2f((int)__builtin_amdgcn_raw_buffer_load_b32(rx, (c[i] & ~kRowEndBit) << 2, 0, 0));
        }
        // ComputeAB: val * x[col]; then request the next slice into the same registers (issued AFTER this slice's
        // gathers so that waiting for the gathers does not wait for the prefetch: vmcnt retires in issue order).
        float p[kE];
#pragma unroll
        for (int j = 0; j < kSliceSteps; ++j) {
            p[4 * j + 0] = i2f((int)w.v[j].x) * xv[4 * j + 0];
            p[4 * j + 1] = i2f((int)w.v[j].y) * xv[4 * j + 1];
            p[4 * j + 2] = i2f((int)w.v[j].z) * xv[4 * j + 2];
            p[4 * j + 3] = i2f((int)w.v[j].w) * xv[4 * j + 3];
        }
        // hipcc would hoist the prefetch above the products (its results land in fresh registers); the waits for the
        // gathers, placed after the point where the three gather paths merge, then count conservatively and wait for
        // the prefetch as well.  Pin the products here and keep memory operations from crossing (PFlow_742 65.3 ->
        // 62.4 us; not in the look-back variant: its one-slice-per-wavefront launches have no next slice to request
        // and lose 0.4 us to the barrier).
        if (!LOOKBACK) {
#pragma unroll
            for (int i = 0; i < kE; ++i) asm volatile("" : "+v"(p[i]));
            asm volatile("" ::: "memory");
        }
        const long long cur = first + local;
        k_slice += n_waves;
        local = k_slice < n_here ? (k_slice + rot >= n_here ? k_slice + rot - n_here : k_slice + rot) : n_here;
        if (strays) {                                  // the stray columns of the slice AFTER the next one, ahead of the next one's words
            const int k2 = k_slice + n_waves;
            const int l2 = k2 < n_here ? (k2 + rot >= n_here ? k2 + rot - n_here : k2 + rot) : -1;
            sx_cur = sx_next;
            sc_next = l2 >= 0 ? __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)((first + l2) * kStraySlots + lane) << 2, 0, 0) : 0xffffffffu;
        }
        if (local < n_here) {
            h = load_int4(hdr + first + local);       // header first: see the prologue
            request_slice<COMPACT>(w, gbase + (size_t)local * slice_bytes, lane);
        }

        // PreAccumulator + row distribution network: lane-local combine + segmented scan per 256-element step
        // A step in which no row ends (long rows: mouse_gene 642 per row, TSOPF 424) needs no scan: its 256 products join
        // the open partial sum through a plain wave reduction with the scan's own tree (lane 63 of the scan is exactly
        // that sum, so the bits do not change) -- a third of the scan's instructions; the slice kernel is VALU-bound on
        // gfx950 once the stream is compact (PFlow_742: 72 % VALU busy at 5.9 TB/s).  The row totals of the other steps go
        // to this wavefront's LDS tile right away (AccumBuffer), one ds_write per row end.
        float carry_step = 0.0f;       // partial sum of the row left open by the previous step
#pragma unroll
        for (int j = 0; j < kSliceSteps; ++j) {
            if (step_has_end & (1u << j)) {
                const float pj[kLaneElems] = {p[4 * j], p[4 * j + 1], p[4 * j + 2], p[4 * j + 3]};
                float tj[kLaneElems];
                scan_step(pj, ends[4 * j], ends[4 * j + 1], ends[4 * j + 2], ends[4 * j + 3], carry_step, tj);
                int pos = r0[j] - row_first;
#pragma unroll
                for (int k = 0; k < kLaneElems; ++k) {
                    if (ends[4 * j + k]) ytile[pos] = tj[k];
                    pos += ends[4 * j + k] ? 1 : 0;
                }
            } else {
                carry_step = wave_sum(((p[4 * j] + p[4 * j + 1]) + p[4 * j + 2]) + p[4 * j + 3]) + carry_step;
            }
        }

        bool deferred = false, rolling = false;
        if (LOOKBACK) {
            // publish this slice's open partial sum as ONE 8-byte {value, launch tag} granule ...
            if (lane == 0) {
                const unsigned long long granule = ((unsigned long long)lb.epoch << 32) | (unsigned)f2i(carry_step);
                mbox_store(mbox + (cur - first), granule);
                __hip_atomic_store(lb.gran + cur, granule, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // ... and finish the cut row of this wavefront's PREVIOUS slice: the slices it reaches back to were
            // taken by the neighbouring wavefronts a whole iteration ago, so their carries are usually there.
            // (Resolving the row in its own iteration made every wavefront wait for its left neighbour's scan each
            // round: lockstep.  One iteration of slack removes most of it, yet on long per-wave chunks a wavefront that
            // runs ahead still waits for the slower ones, and the variant itself is ~5 us slower than the fix-up
            // variant's main kernel on PFlow_742 -- so large matrices keep the fix-up launch, hispmv_abi.cpp.)
            if (pend_slice >= 0) {
                const float chain = lookback_chain(lb, mbox, first, pend_slice, pend_len, lane);
                if (lane == 0) {
                    const float tt = pend_t + chain;
                    __builtin_amdgcn_raw_buffer_store_b32((unsigned)f2i(HAS_BETA ? alpha * tt + beta * pend_b : alpha * tt), ry,
                                                          (unsigned)pend_row << 2, 0, 0);
                }
                pend_slice = -1;
            }
            deferred = chain_len > (int)(cur - first);     // the chain leaves this group: resolved after the loop
            rolling = chain_len > 0 && !deferred;          // inside the group: resolved one iteration later
            if (deferred) { def_slice = cur; def_row = row_first; def_len = chain_len; }
            if (rolling) { pend_slice = cur; pend_row = row_first; pend_len = chain_len; }
        }

        // Compute_C: the row totals leave this wavefront's LDS tile as COALESCED y = alpha*total + beta*bias stores:
        // ceil(n_rows/64) load/store pairs per slice instead of mostly-empty predicated ones (the output phase cost
        // 25-30 % that way).
        const bool held = LOOKBACK && (deferred || rolling);   // the slice's first row is stored later, with its chain
        if (!LOOKBACK) {
            out_t0 = lane < n_rows ? ytile[lane] : 0.0f;
            out_t1 = lane + 64 < n_rows ? ytile[lane + 64] : 0.0f;
            out_b0 = bpre0; out_b1 = bpre1;       // still in flight: consumed behind the next iteration's wait
            out_d0 = lane < n_rows ? (unsigned)(row_first + lane) << 2 : kNoAccess;
            out_d1 = lane + 64 < n_rows ? (unsigned)(row_first + lane + 64) << 2 : kNoAccess;
            out_carry = carry_step;
            out_dc = lane == 0 ? (unsigned)cur << 2 : kNoAccess;
        }
        // rows 128.. of a short-row slice (fix-up variant): 256 rows per round, their bias loads issued together -- one wait
        // per round instead of one per 64 rows (ASIC_680k: 260 rows per slice)
        for (int i0 = 128; !LOOKBACK && i0 < n_rows; i0 += 256) {
            float bb[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (HAS_BETA) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + 64 * u + lane;
                    bb[u] = i2f((int)__builtin_amdgcn_raw_buffer_load_b32(rb, i < n_rows ? (unsigned)(row_first + i) << 2 : kNoAccess, 0, 0));
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + 64 * u + lane;
                const float tt = i < n_rows ? ytile[i] : 0.0f;
                __builtin_amdgcn_raw_buffer_store_b32((unsigned)f2i(HAS_BETA ? alpha * tt + beta * bb[u] : alpha * tt), ry,
                                                      i < n_rows ? (unsigned)(row_first + i) << 2 : kNoAccess, 0, 0);
            }
        }
        for (int i = lane; LOOKBACK && i < n_rows; i += 64) {
            const float tt = ytile[i];
            const unsigned dst = (held && i == 0) ? kNoAccess : (unsigned)(row_first + i) << 2;
            if (HAS_BETA) {
                const float b = (i < 64) ? bpre0 : (i < 128) ? bpre1
                              : i2f((int)__builtin_amdgcn_raw_buffer_load_b32(rb, (unsigned)(row_first + i) << 2, 0, 0));
                if (LOOKBACK && deferred && i == 0) { def_t = tt; def_b = b; }
                if (LOOKBACK && rolling && i == 0) { pend_t = tt; pend_b = b; }
                __builtin_amdgcn_raw_buffer_store_b32((unsigned)f2i(alpha * tt + beta * b), ry, dst, 0, 0);
            } else {
                if (LOOKBACK && deferred && i == 0) def_t = tt;
                if (LOOKBACK && rolling && i == 0) pend_t = tt;
                __builtin_amdgcn_raw_buffer_store_b32((unsigned)f2i(alpha * tt), ry, dst, 0, 0);
            }
        }
    }
    if (!LOOKBACK) {      // the stores of this wavefront's last slice
        __builtin_amdgcn_raw_buffer_store_b32((unsigned)f2i(HAS_BETA ? alpha * out_t0 + beta * out_b0 : alpha * out_t0), ry, out_d0, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32((unsigned)f2i(HAS_BETA ? alpha * out_t1 + beta * out_b1 : alpha * out_t1), ry, out_d1, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32((unsigned)f2i(out_carry), rc, out_dc, 0, 0);
    }
    if (LOOKBACK && pend_slice >= 0) {    // the cut row of this wavefront's last slice
        const float chain = lookback_chain(lb, mbox, first, pend_slice, pend_len, lane);
        if (lane == 0) {
            const float tt = pend_t + chain;
            __builtin_amdgcn_raw_buffer_store_b32((unsigned)f2i(HAS_BETA ? alpha * tt + beta * pend_b : alpha * tt), ry,
                                                  (unsigned)pend_row << 2, 0, 0);
        }
    }
    if (LOOKBACK && def_slice >= 0) {     // wave-uniform: the row that crosses into this group from the previous one
        const float chain = lookback_chain(lb, mbox, first, def_slice, def_len, lane);
        if (lane == 0) {
            const float tt = def_t + chain;
            __builtin_amdgcn_raw_buffer_store_b32((unsigned)f2i(HAS_BETA ? alpha * tt + beta * def_b : alpha * tt), ry,
                                                  (unsigned)def_row << 2, 0, 0);
        }
    }
}

template <bool HAS_BETA, bool USE_LDS, bool LOOKBACK, bool STRAYS = false>
__device__ __forceinline__ void slices_body(
    const char* __restrict__ stream, const int4* __restrict__ hdr, const int4* __restrict__ groups,
    const int4* __restrict__ frags, const float* __restrict__ x, const float* bias, float* y,
    float* __restrict__ carry, float alpha, float beta, long long n_slices, int group_slices,
    int lds_floats, int ytile_floats, int cols, int rows, const LookbackArgs& lb, long long group, const SubBlock sb) {
    __shared__ long long s_group;
    if (LOOKBACK) {
        // Groups are handed out in START order (one ticket per workgroup), so every slice a wavefront
        // may have to wait for belongs to a workgroup that is already running or done -- the carry
        // look-back cannot deadlock whatever order the dispatcher picks.  The ticket counter is
        // never reset: launch k consumes tickets [k*n_groups, (k+1)*n_groups).
        // When the whole grid is co-resident (lb.use_ticket == 0, decided on the host from the launch
        // plan) every workgroup is running anyway and blockIdx order needs no ticket.
        if (lb.use_ticket) {
            if (threadIdx.x == 0)
                s_group = (long long)(__hip_atomic_fetch_add(lb.ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - lb.ticket_base);
            __syncthreads();
            group = s_group;
        }
    }
    if constexpr (USE_LDS) {
        // (a sub-block of the step kernel past the matrix's last group: an empty group -- it only joins the barrier)
        const int4 g = group * group_slices < n_slices ? load_int4(groups + group) : int4{0, 0, 0, 0};     // {first fragment, fragments, offset of the group's slices, compact}
        if (__builtin_amdgcn_readfirstlane(g.w) != 0)
            slices_group<HAS_BETA, true, LOOKBACK, true, STRAYS>(stream, hdr, groups, frags, x, bias, y, carry, alpha, beta, n_slices, group_slices,
                                                         lds_floats, ytile_floats, cols, rows, lb, group, g, sb);
        else
            slices_group<HAS_BETA, true, LOOKBACK, false>(stream, hdr, groups, frags, x, bias, y, carry, alpha, beta, n_slices, group_slices,
                                                          lds_floats, ytile_floats, cols, rows, lb, group, g, sb);
    } else {
        slices_group<HAS_BETA, false, LOOKBACK, false>(stream, hdr, groups, frags, x, bias, y, carry, alpha, beta, n_slices, group_slices,
                                                       lds_floats, ytile_floats, cols, rows, lb, group, int4{0, 0, 0, 0}, sb);
    }
}
template <bool HAS_BETA, bool USE_LDS, bool LOOKBACK, bool STRAYS = false>
__device__ __forceinline__ void slices_body(
    const char* __restrict__ stream, const int4* __restrict__ hdr, const int4* __restrict__ groups,
    const int4* __restrict__ frags, const float* __restrict__ x, const float* bias, float* y,
    float* __restrict__ carry, float alpha, float beta, long long n_slices, int group_slices,
    int lds_floats, int ytile_floats, int cols, int rows, const LookbackArgs& lb, long long group) {
    slices_body<HAS_BETA, USE_LDS, LOOKBACK, STRAYS>(stream, hdr, groups, frags, x, bias, y, carry, alpha, beta, n_slices, group_slices,
                                                     lds_floats, ytile_floats, cols, rows, lb, group, SubBlock::whole());
}

template <bool HAS_BETA, bool USE_LDS, bool LOOKBACK, bool STRAYS = false>
__global__ __launch_bounds__(1024) void spmv_slices_kernel(
    const char* __restrict__ words, const int4* __restrict__ hdr, const int4* __restrict__ groups,
    const int4* __restrict__ frags, const float* __restrict__ x, const float* bias, float* y,
    float* __restrict__ carry, float alpha, float beta, long long n_slices, int group_slices,
    int lds_floats, int ytile_floats, int cols, int rows, LookbackArgs lb) {
    slices_body<HAS_BETA, USE_LDS, LOOKBACK, STRAYS>(words, hdr, groups, frags, x, bias, y, carry, alpha, beta, n_slices, group_slices,
                                             lds_floats, ytile_floats, cols, rows, lb, (long long)blockIdx.x);
}

// ---------------------------------------------------------------------------
// Several matrices in ONE launch (hispmv_spmv_device_batch): the grid is the concatenation of the matrices' groups;
// a workgroup finds its matrix from the prefix of group counts (kernel argument, no memory access), reads that
// matrix's descriptor from a device table and runs the ordinary body with the fix-up carry variant.  Independent
// SpMVs -- the 20 matrices of the benchmark set, the heads of a model -- then share launch ramps and tails instead of
// paying 6-20 us of launch latency each.  All matrices of a launch have the same workgroup size.
// ---------------------------------------------------------------------------
template <bool STRAYS>
__global__ __launch_bounds__(1024) void spmv_slices_multi_kernel(const MultiEntry* __restrict__ table, MultiPrefix prefix, float alpha) {
    int k = 0;
#pragma unroll 1
    while (k + 1 < prefix.n && (long long)blockIdx.x >= prefix.begin[k + 1]) ++k;
    long long group = (long long)blockIdx.x - prefix.begin[k];
    int e = prefix.first[k];
    const int tiles = prefix.tiles[k];
    if (tiles > 1) {              // XCD-pinned column tiles (hispmv_kernels.h): tile from the block's index mod 8
        const int per = 8 / tiles, res = (int)(group & 7);
        e += res / per;
        group = (group >> 3) * per + (res % per);
    }
    const MultiEntry t = table[e];
    if (group * t.group_slices >= t.n_slices) return;      // (a pinned tile with fewer groups than its siblings)
    WGT_BEGIN();
    const LookbackArgs lb{};
    // beta is per entry: the first column tile of a matrix applies beta*bias, its other tiles write alpha*A_t*x into the
    // handle's partial vectors (no bias read) -- both kinds share the grid, a workgroup runs one of the two bodies
    if (t.beta != 0.0f)
        slices_body<true, true, false, STRAYS>((const char*)t.words, t.hdr, t.groups, t.frags, t.x, t.bias, t.y, t.carry, alpha, t.beta,
                                       t.n_slices, t.group_slices, t.lds_floats, t.ytile_floats, t.cols, t.rows, lb, group);
    else
        slices_body<false, true, false, STRAYS>((const char*)t.words, t.hdr, t.groups, t.frags, t.x, t.y, t.y, t.carry, alpha, 0.0f,
                                        t.n_slices, t.group_slices, t.lds_floats, t.ytile_floats, t.cols, t.rows, lb, group);
    WGT_END(blockDim.x == 1024 ? 1 : 2, e, group);
}

// Fix-up of all matrices of a multi launch: thread blocks are concatenated the same way.
__global__ __launch_bounds__(256) void spmv_fixup_multi_kernel(const MultiFixEntry* __restrict__ table, MultiPrefix prefix, float alpha) {
    int e = 0;
#pragma unroll 1
    while (e + 1 < prefix.n && (long long)blockIdx.x >= prefix.begin[e + 1]) ++e;
    const MultiFixEntry t = table[e];
    const int i = (int)(blockIdx.x - prefix.begin[e]) * blockDim.x + threadIdx.x;
    if (i >= t.n) return;
    const int4 f = t.fix[i];
    float s = 0.0f;
    for (int k = 0; k < f.z; ++k) s += t.carry[f.y + k];
    t.y[f.x] += alpha * s;
}

// Column tiles t > 0 of a matrix write alpha*A_t*x into partial vectors of the handle (so that every tile runs in the
// same round as tile 0 instead of accumulating in place behind it); after the fix-up of the cut rows this pass adds them:
// y = ((y + part_1) + part_2) + ...   parts[t] = parts + t*part_stride; grid.y = vector of a batched pass.
__global__ __launch_bounds__(256) void spmv_merge_parts_kernel(float* __restrict__ y, const float* __restrict__ parts, int n_parts,
                                                               long long part_stride, int rows, long long y_stride, long long vec_stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    y += blockIdx.y * y_stride;
    parts += blockIdx.y * vec_stride;
    float s = y[i];
    for (int t = 0; t < n_parts; ++t) s += parts[(size_t)t * part_stride + i];
    y[i] = s;
}
__global__ __launch_bounds__(256) void spmv_merge_multi_kernel(const MultiMergeEntry* __restrict__ table, MultiPrefix prefix) {
    int e = 0;
#pragma unroll 1
    while (e + 1 < prefix.n && (long long)blockIdx.x >= prefix.begin[e + 1]) ++e;
    const MultiMergeEntry t = table[e];
    const int i = (int)(blockIdx.x - prefix.begin[e]) * blockDim.x + threadIdx.x;
    if (i >= t.rows) return;
    float s = t.y[i];
    for (int k = 0; k < t.n_parts; ++k) s += t.parts[(size_t)k * t.part_stride + i];
    t.y[i] = s;
}

// The tail of a batch call in one launch (TailMergeEntry, hispmv_kernels.h): blocks [0, fix_prefix.begin[n]) finish cut rows
// as spmv_fixup_multi_kernel does; the blocks behind them merge the partial vectors of column-tiled matrices and apply the
// fix-ups of those matrices' rows on the way (same expression order as fix-up launch + merge launch: same bits).  One launch
// instead of two behind the join of the main launches: the step's serial tail is ~4.5 us shorter.
__global__ __launch_bounds__(256) void spmv_tail_multi_kernel(const MultiFixEntry* __restrict__ fix_table, MultiPrefix fix_prefix,
                                                              const TailMergeEntry* __restrict__ merge_table, MultiPrefix merge_prefix, float alpha) {
    const long long n_fix_blocks = fix_prefix.begin[fix_prefix.n];
    if ((long long)blockIdx.x < n_fix_blocks) {
        int e = 0;
#pragma unroll 1
        while (e + 1 < fix_prefix.n && (long long)blockIdx.x >= fix_prefix.begin[e + 1]) ++e;
        const MultiFixEntry t = fix_table[e];
        const int i = (int)(blockIdx.x - fix_prefix.begin[e]) * blockDim.x + threadIdx.x;
        if (i >= t.n) return;
        const int4 f = t.fix[i];
        float s = 0.0f;
        for (int k = 0; k < f.z; ++k) s += t.carry[f.y + k];
        t.y[f.x] += alpha * s;
        return;
    }
    const long long b = (long long)blockIdx.x - n_fix_blocks;
    int e = 0;
#pragma unroll 1
    while (e + 1 < merge_prefix.n && b >= merge_prefix.begin[e + 1]) ++e;
    const TailMergeEntry& t = merge_table[e];
    const int rows = t.rows, n_parts = t.n_parts;
    const int i = (int)(b - merge_prefix.begin[e]) * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const int32_t* fr = t.fix_of_row;
    float s = t.y[i];
    for (int q = 0; q <= n_parts; ++q) {
        float v = q == 0 ? s : t.parts[(size_t)(q - 1) * t.part_stride + i];
        const int fi = fr ? fr[(size_t)q * rows + i] : -1;
        if (fi >= 0) {
            const int4 f = t.fix[q][fi];
            const float* cy = t.carry[q];
            float c = 0.0f;
            for (int k = 0; k < f.z; ++k) c += cy[f.y + k];
            v += alpha * c;
        }
        s = q == 0 ? v : s + v;
    }
    t.y[i] = s;
}

// ---------------------------------------------------------------------------
// Batched slice kernel: NV input vectors per pass over the stream (FpgaHandle::linear with num_vecs > 1; the
// reference runs its kernel once per vector, fpga_handle.cpp:366-379 -- A is read num_vecs times).  Vector v
// is x + v*cols, its result y + v*rows, its carries carry + v*n_slices; bias is shared (bias_stride 0) or per
// vector (bias_stride = rows: column tiles t > 0 accumulate on y).  The slice's words stay in registers while
// the NV vectors go through gather, product, scan and output one after the other, so every vector sees exactly
// the arithmetic of the single-vector kernel with the fix-up carry variant (bitwise the same y); the next
// slice is requested after the last vector's products.  The x windows of the NV vectors sit side by side in LDS.
// Host guarantees: cols % 4 == 0 when USE_LDS, cols*NV < 2^30, rows*NV < 2^30, NV*lds_floats + tiles fit LDS.
// ---------------------------------------------------------------------------
template <bool HAS_BETA, bool USE_LDS, int NV, bool COMPACT>
__device__ __forceinline__ void batched_group(
    const char* __restrict__ stream, const int4* __restrict__ hdr,
    const int4* __restrict__ frags,
    const float* __restrict__ x, const float* bias, float* y,   // bias may alias y
    float* __restrict__ carry, float alpha, float beta, long long n_slices, int group_slices,
    int lds_floats, int ytile_floats, int cols, int rows, int bias_stride, int4 g) {
    extern __shared__ float xs[];
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, cols * NV * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)bias, 0, (bias_stride ? rows * NV : rows) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, rows * NV * 4, 0x00020000);
    constexpr unsigned kNoAccess = 0xffffffffu;
    constexpr int kE = kSliceSteps * kLaneElems;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    float* const ytile = xs + (USE_LDS ? NV * lds_floats : 0) + wave * ytile_floats;
    const long long group = blockIdx.x;
    const long long first = group * group_slices;
    const long long last = (first + group_slices < n_slices) ? first + group_slices : n_slices;   // exclusive
    constexpr int slice_bytes = COMPACT ? kCompactSliceBytes : kWideSliceBytes;
    const char* const gbase = stream + (USE_LDS ? (size_t)(unsigned)__builtin_amdgcn_readfirstlane(g.z) * kSliceUnit
                                                : (size_t)first * kWideSliceBytes);

    // the walk of slices_group (rotated start: the packer places a slice's strays by its position in this walk)
    const int n_here = (int)(last > first ? last - first : 0);
    const int rot = n_here == 0 ? 0 : (int)((unsigned long long)group * 29ull % (unsigned)n_here);
    int k_slice = wave;
    auto local_at = [&](int k) { return k < n_here ? (k + rot >= n_here ? k + rot - n_here : k + rot) : n_here; };
    long long slice = first + local_at(k_slice);
    // stray slots (hispmv_plan.h): the columns of the next slice's strays travel with its words; their x values are gathered per
    // vector at the slice's turn (this kernel is VALU-bound: the exposed round trip hides behind the other wavefronts)
    const bool strays = COMPACT && (__builtin_amdgcn_readfirstlane(g.w) & 2) != 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(hdr + n_slices), 0, strays ? (int)(n_slices * (kStraySlots * 4)) : 0, 0x00020000);
    unsigned sc = 0xffffffffu;
    SliceRaw<COMPACT> w;
    int4 h = int4{0, 0, 0, 0};
    if (slice < last) {
        if (strays) sc = __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)(slice * kStraySlots + lane) << 2, 0, 0);
        h = load_int4(hdr + slice);
        request_slice<COMPACT>(w, gbase + (size_t)(slice - first) * slice_bytes, lane);
    }
    bool in_lds = false;
    if (USE_LDS) {
        in_lds = g.y > 0;
        for (int v = 0; v < NV; ++v) {
            const float* xv = x + (size_t)v * cols;
            stage_fragments(frags, g.x, g.y, xv, xs + v * lds_floats, cols, lane, wave, n_waves);
        }
        __syncthreads();
    }

    while (slice < last) {
        int row = __builtin_amdgcn_readfirstlane(h.x);
        const int row_first = row;
        const int n_rows = __builtin_amdgcn_readfirstlane(h.z);
        const bool spills = USE_LDS && __builtin_amdgcn_readfirstlane(h.w) != 0;
        unsigned c[kE];
        decode_metas<COMPACT>(w, c);
        float val[kE];
#pragma unroll
        for (int j = 0; j < kSliceSteps; ++j) {
            val[4 * j + 0] = i2f((int)w.v[j].x); val[4 * j + 1] = i2f((int)w.v[j].y);
            val[4 * j + 2] = i2f((int)w.v[j].z); val[4 * j + 3] = i2f((int)w.v[j].w);
        }
        int r0[kSliceSteps];
        // (Round 4: taking each vector's row-end masks through an opaque copy, so that the 4 x 7 scan flag words are recomputed per
        // vector instead of living -- spilled -- across the unrolled vector loop, cut the SGPR reloads inside the loop from 739 to
        // 520 per slice (tools/spill_report.py) and measured no gain: the 8192 x 8192 layer with 8 vectors 54.4 us against 50.8.)
        bool ends[kE];
#pragma unroll
        for (int j = 0; j < kSliceSteps; ++j) {
            int below = 0, total = 0;
#pragma unroll
            for (int k = 0; k < kLaneElems; ++k) {
                ends[4 * j + k] = (c[4 * j + k] & kRowEndBit) != 0;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(ends[4 * j + k]);
                below += lanes_below(m);
                total += __builtin_popcountll(m);
            }
            r0[j] = row + below;
            row += total;
        }
        const long long cur = slice;
        if (strays) {
            float sx[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v)
                sx[v] = i2f((int)__builtin_amdgcn_raw_buffer_load_b32(rx, sc == 0xffffffffu ? kNoAccess : (sc + (unsigned)v * (unsigned)cols) << 2, 0, 0));
#pragma unroll
            for (int v = 0; v < NV; ++v) (xs + v * lds_floats + (lds_floats - n_waves * kStraySlots) + wave * kStraySlots)[lane] = sx[v];
        }

#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const unsigned xoff = (unsigned)v * (unsigned)cols, yoff = (unsigned)v * (unsigned)rows;
            const unsigned boff = (unsigned)v * (unsigned)bias_stride;
            float bpre0 = 0.0f, bpre1 = 0.0f;
            if (HAS_BETA) {
                bpre0 = i2f((int)__builtin_amdgcn_raw_buffer_load_b32(rb, lane < n_rows ? (boff + row_first + lane) << 2 : kNoAccess, 0, 0));
                bpre1 = i2f((int)__builtin_amdgcn_raw_buffer_load_b32(rb, lane + 64 < n_rows ? (boff + row_first + lane + 64) << 2 : kNoAccess, 0, 0));
            }
            float xg[kE];
            const float* xw = xs + v * lds_floats;
            if (USE_LDS && in_lds && !spills) {
#pragma unroll
                for (int i = 0; i < kE; ++i) xg[i] = xw[c[i] & ~kRowEndBit];
            } else if (USE_LDS && in_lds) {
#pragma unroll
                for (int i = 0; i < kE; ++i) {
                    const unsigned ci = c[i] & ~kRowEndBit;
                    xg[i] = i2f((int)__builtin_amdgcn_raw_buffer_load_b32(rx, (ci & kGlobalColBit) ? ((ci & ~kGlobalColBit) + xoff) << 2 : kNoAccess, 0, 0));
                }
#pragma unroll
                for (int i = 0; i < kE; ++i) {
                    const unsigned ci = c[i] & ~kRowEndBit;
                    const float l = xw[(ci & kGlobalColBit) ? 0u : ci];
                    xg[i] = (ci & kGlobalColBit) ? xg[i] : l;
                }
            } else {
#pragma unroll
                for (int i = 0; i < kE; ++i) xg[i] = i2f((int)__builtin_amdgcn_raw_buffer_load_b32(rx, ((c[i] & ~kRowEndBit) + xoff) << 2, 0, 0));
            }
            if (v == NV - 1) {
                // the slice's values and metas live in val[] / c[]: request this wavefront's next slice behind the last
                // vector's gathers (vmcnt retires in issue order: waiting for the gathers must not wait for the prefetch)
                k_slice += n_waves;
                slice = first + local_at(k_slice);
                if (slice < last) {
                    if (strays) sc = __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)(slice * kStraySlots + lane) << 2, 0, 0);
                    h = load_int4(hdr + slice);
                    request_slice<COMPACT>(w, gbase + (size_t)(slice - first) * slice_bytes, lane);
                }
            }
            float t[kE];
            float carry_step = 0.0f;
#pragma unroll
            for (int j = 0; j < kSliceSteps; ++j) {
                const float pj[kLaneElems] = {val[4 * j] * xg[4 * j], val[4 * j + 1] * xg[4 * j + 1], val[4 * j + 2] * xg[4 * j + 2], val[4 * j + 3] * xg[4 * j + 3]};
                float tj[kLaneElems];
                scan_step(pj, ends[4 * j], ends[4 * j + 1], ends[4 * j + 2], ends[4 * j + 3], carry_step, tj);
                t[4 * j] = tj[0]; t[4 * j + 1] = tj[1]; t[4 * j + 2] = tj[2]; t[4 * j + 3] = tj[3];
            }
#pragma unroll
            for (int j = 0; j < kSliceSteps; ++j) {
                int pos = r0[j] - row_first;
#pragma unroll
                for (int k = 0; k < kLaneElems; ++k) {
                    if (ends[4 * j + k]) ytile[pos] = t[4 * j + k];
                    pos += ends[4 * j + k] ? 1 : 0;
                }
            }
            for (int i = lane; i < n_rows; i += 64) {
                const float tt = ytile[i];
                if (HAS_BETA) {
                    const float b = (i < 64) ? bpre0 : (i < 128) ? bpre1
                                  : i2f((int)__builtin_amdgcn_raw_buffer_load_b32(rb, (boff + row_first + i) << 2, 0, 0));
                    __builtin_amdgcn_raw_buffer_store_b32((unsigned)f2i(alpha * tt + beta * b), ry, (yoff + row_first + i) << 2, 0, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b32((unsigned)f2i(alpha * tt), ry, (yoff + row_first + i) << 2, 0, 0);
                }
            }
            if (lane == 0) store_float(carry + (long long)v * n_slices + cur, carry_step);
        }
    }
}

template <bool HAS_BETA, bool USE_LDS, int NV>
__global__ __launch_bounds__(1024) void spmv_slices_batched_kernel(
    const char* __restrict__ stream, const int4* __restrict__ hdr, const int4* __restrict__ groups,
    const int4* __restrict__ frags, const float* __restrict__ x, const float* bias, float* y,
    float* __restrict__ carry, float alpha, float beta, long long n_slices, int group_slices,
    int lds_floats, int ytile_floats, int cols, int rows, int bias_stride) {
    if constexpr (USE_LDS) {
        const int4 g = load_int4(groups + blockIdx.x);
        if (__builtin_amdgcn_readfirstlane(g.w) != 0)
            batched_group<HAS_BETA, true, NV, true>(stream, hdr, frags, x, bias, y, carry, alpha, beta, n_slices, group_slices, lds_floats,
                                                    ytile_floats, cols, rows, bias_stride, g);
        else
            batched_group<HAS_BETA, true, NV, false>(stream, hdr, frags, x, bias, y, carry, alpha, beta, n_slices, group_slices, lds_floats,
                                                     ytile_floats, cols, rows, bias_stride, g);
    } else {
        batched_group<HAS_BETA, false, NV, false>(stream, hdr, frags, x, bias, y, carry, alpha, beta, n_slices, group_slices, lds_floats,
                                                  ytile_floats, cols, rows, bias_stride, int4{0, 0, 0, 0});
    }
}

// Fix-up for rows shared between slices: y[row] += alpha * (carry[first] + ... + carry[first+len-1]),
// summed in slice order.  One thread per entry (short chains) ...
__global__ __launch_bounds__(256) void spmv_fixup_short_kernel(const int4* __restrict__ fix, int n,
                                                               const float* __restrict__ carry,
                                                               float* __restrict__ y, float alpha,
                                                               long long carry_stride, long long y_stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    carry += blockIdx.y * carry_stride;     // blockIdx.y = vector of a batched pass (0 otherwise)
    y += blockIdx.y * y_stride;
    const int4 f = fix[i];
    float s = 0.0f;
    for (int k = 0; k < f.z; ++k) s += carry[f.y + k];
    y[f.x] += alpha * s;
}
// ... or one wavefront per entry (a heavy row spanning many slices).
__global__ __launch_bounds__(256) void spmv_fixup_long_kernel(const int4* __restrict__ fix, int n,
                                                              const float* __restrict__ carry,
                                                              float* __restrict__ y, float alpha,
                                                              long long carry_stride, long long y_stride) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    carry += blockIdx.y * carry_stride;
    y += blockIdx.y * y_stride;
    const int4 f = fix[i];
    float s = 0.0f;
    for (int k = lane; k < f.z; k += 64) s += carry[f.y + k];
    s = wave_sum(s);
    if (lane == 0) y[f.x] += alpha * s;
}

template <bool HAS_BETA, bool USE_LDS, bool LOOKBACK>
static void launch_slices(const SpmvDeviceMatrix& m, const LookbackArgs& lb, const float* x, const float* bias, float* y,
                          float alpha, float beta, hipStream_t stream) {
    const size_t lds = ((USE_LDS ? (size_t)m.lds_floats : 0) + (size_t)m.ytile_floats * (m.block_threads / 64)) * sizeof(float) +
                       (LOOKBACK ? (size_t)m.group_slices * 8 : 0);
    if (USE_LDS && !LOOKBACK && m.has_strays)       // (stray slots: the instantiation that fetches them; never with look-back, hispmv_abi.cpp)
        hipLaunchKernelGGL((spmv_slices_kernel<HAS_BETA, USE_LDS && !LOOKBACK, false, USE_LDS && !LOOKBACK>), dim3((unsigned)m.n_groups), dim3(m.block_threads), lds, stream,
                           (const char*)m.words, m.hdr, m.groups, m.frags, x, bias, y, m.carry, alpha, beta,
                           (long long)m.n_slices, m.group_slices, m.lds_floats, m.ytile_floats, m.cols, m.rows, lb);
    else
    hipLaunchKernelGGL((spmv_slices_kernel<HAS_BETA, USE_LDS, LOOKBACK>), dim3((unsigned)m.n_groups), dim3(m.block_threads), lds, stream,
                       (const char*)m.words, m.hdr, m.groups, m.frags, x, bias, y, m.carry, alpha, beta,
                       (long long)m.n_slices, m.group_slices, m.lds_floats, m.ytile_floats, m.cols, m.rows, lb);
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
    const int max_lds = 160 * 1024 - 256;   // the kernels also hold a few bytes of static LDS
    if ((e = hipFuncSetAttribute((const void*)spmv_slices_kernel<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)spmv_slices_kernel<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)spmv_slices_kernel<true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)spmv_slices_kernel<false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)spmv_slices_kernel<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)spmv_slices_kernel<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)spmv_slices_kernel<true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)spmv_slices_kernel<false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)spmv_slices_kernel<true, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)spmv_slices_kernel<false, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds)) != hipSuccess) return e;
    return hipSuccess;
}

hipError_t launch_spmv(SpmvDeviceMatrix& m_in, const float* x, const float* bias, float* y,
                       float alpha, float beta, hipStream_t stream, bool fixup_only) {
    (void)hipGetLastError();   // the status returned below must be this launch's, not a stale one of the thread (e.g. PyTorch's pointer queries)
    // fixup_only (FpgaHandle::linear): the carry hand-off through the fix-up launch even where a single launch would merge
    // the cut rows in-kernel -- the summation order of the batched kernels, so that every vector of a `linear` call has
    // the same bits whatever the number of vectors in the call
    SpmvDeviceMatrix forced;
    if (fixup_only && m_in.lookback) { forced = m_in; forced.lookback = false; }
    SpmvDeviceMatrix& m = (fixup_only && m_in.lookback) ? forced : m_in;
    if (m.n_slices > 0) {
        if (m.n_groups <= 0 || m.n_groups > 0x7fffffffLL) return hipErrorInvalidValue;
        LookbackArgs lb{};
        if (m.lookback) {
            lb.gran = m.gran; lb.ticket = m.ticket; lb.err = m.err;
            lb.use_ticket = m.use_ticket ? 1 : 0;
            lb.ticket_base = m.ticket_launches * (unsigned long long)m.n_groups;
            if (m.use_ticket) m.ticket_launches++;
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
                           m.fix_short, m.n_fix_short, m.carry, y, alpha, 0LL, 0LL);
    if (m.n_fix_long > 0)
        hipLaunchKernelGGL(spmv_fixup_long_kernel, dim3((m.n_fix_long + 3) / 4), dim3(256), 0, stream,
                           m.fix_long, m.n_fix_long, m.carry, y, alpha, 0LL, 0LL);
    return hipGetLastError();
}

template <bool HAS_BETA, bool USE_LDS, int NV>
static hipError_t launch_batched(const SpmvDeviceMatrix& m, const float* x, const float* bias, int bias_stride, float* y,
                                 float alpha, float beta, hipStream_t stream) {
    const size_t lds = ((USE_LDS ? (size_t)m.lds_floats * NV : 0) + (size_t)m.ytile_floats * (m.block_threads / 64)) * sizeof(float);
    static bool raised = false;      // per instantiation
    if (!raised) {
        hipError_t e = hipFuncSetAttribute((const void*)spmv_slices_batched_kernel<HAS_BETA, USE_LDS, NV>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        if (e != hipSuccess) return e;
        raised = true;
    }
    hipLaunchKernelGGL((spmv_slices_batched_kernel<HAS_BETA, USE_LDS, NV>), dim3((unsigned)m.n_groups), dim3(m.block_threads), lds, stream,
                       (const char*)m.words, m.hdr, m.groups, m.frags, x, bias, y, m.carry, alpha, beta,
                       (long long)m.n_slices, m.group_slices, m.lds_floats, m.ytile_floats, m.cols, m.rows, bias_stride);
    return hipGetLastError();
}

int spmv_batch_width(const SpmvDeviceMatrix& m, int64_t vecs) {
    if (vecs < 2) return 1;
    if (m.lds_floats > 0 && (m.cols & 3)) return 1;   // the windows of vectors 1.. would be staged from unaligned rows of x
    for (int nv = kMaxBatch; nv >= 2; nv >>= 1) {
        if (nv > vecs) continue;
        if ((int64_t)m.cols * nv >= (1 << 30) || (int64_t)m.rows * nv >= (1 << 30)) continue;
        const size_t lds = ((size_t)m.lds_floats * nv + (size_t)m.ytile_floats * (m.block_threads / 64)) * sizeof(float);
        if (lds <= 160 * 1024 - 256) return nv;
    }
    return 1;
}

hipError_t launch_spmv_batched(SpmvDeviceMatrix& m, int nv, const float* x, const float* bias, int bias_stride, float* y,
                               float alpha, float beta, hipStream_t stream) {
    (void)hipGetLastError();   // the status returned below must be this launch's, not a stale one of the thread (e.g. PyTorch's pointer queries)
    if (nv != 2 && nv != 4) return hipErrorInvalidValue;
    if (m.n_slices > 0) {
        if (m.n_groups <= 0 || m.n_groups > 0x7fffffffLL) return hipErrorInvalidValue;
        const bool lds = m.lds_floats > 0;
        hipError_t e;
        const bool hb = beta != 0.0f;
        if (!hb) bias = y;      // never read; keeps the buffer descriptor well-formed
        if (nv == 4) e = hb ? (lds ? launch_batched<true, true, 4>(m, x, bias, bias_stride, y, alpha, beta, stream)
                                   : launch_batched<true, false, 4>(m, x, bias, bias_stride, y, alpha, beta, stream))
                            : (lds ? launch_batched<false, true, 4>(m, x, bias, bias_stride, y, alpha, beta, stream)
                                   : launch_batched<false, false, 4>(m, x, bias, bias_stride, y, alpha, beta, stream));
        else         e = hb ? (lds ? launch_batched<true, true, 2>(m, x, bias, bias_stride, y, alpha, beta, stream)
                                   : launch_batched<true, false, 2>(m, x, bias, bias_stride, y, alpha, beta, stream))
                            : (lds ? launch_batched<false, true, 2>(m, x, bias, bias_stride, y, alpha, beta, stream)
                                   : launch_batched<false, false, 2>(m, x, bias, bias_stride, y, alpha, beta, stream));
        if (e != hipSuccess) return e;
    }
    if (m.n_fix_short > 0)     // one fix-up launch for all vectors of the pass (grid.y = vector)
        hipLaunchKernelGGL(spmv_fixup_short_kernel, dim3((m.n_fix_short + 255) / 256, nv), dim3(256), 0, stream,
                           m.fix_short, m.n_fix_short, m.carry, y, alpha, (long long)m.n_slices, (long long)m.rows);
    if (m.n_fix_long > 0)
        hipLaunchKernelGGL(spmv_fixup_long_kernel, dim3((m.n_fix_long + 3) / 4, nv), dim3(256), 0, stream,
                           m.fix_long, m.n_fix_long, m.carry, y, alpha, (long long)m.n_slices, (long long)m.rows);
    return hipGetLastError();
}

hipError_t launch_spmv_multi(const SpmvDeviceMatrix* const* parts, int n, const uint8_t* item_tiles, int n_items,
                             const MultiEntry* d_table, float alpha, hipStream_t stream) {
    (void)hipGetLastError();   // the status returned below must be this launch's, not a stale one of the thread (e.g. PyTorch's pointer queries)
    if (n <= 0) return hipSuccess;
    if (n > kMultiMax) return hipErrorInvalidValue;
    if (!item_tiles) n_items = n;
    MultiPrefix px{};
    px.n = n_items;
    long long g = 0;
    size_t lds = 0;
    const int threads = parts[0]->block_threads;
    int e = 0;
    for (int k = 0; k < n_items; ++k) {
        const int tiles = item_tiles ? item_tiles[k] : 1;
        if (tiles != 1 && tiles != 2 && tiles != 4 && tiles != 8) return hipErrorInvalidValue;
        if (e + tiles > n) return hipErrorInvalidValue;
        if (tiles > 1) g = (g + 7) & ~7LL;                 // a pinned set starts at a multiple of 8
        px.begin[k] = g; px.first[k] = (uint8_t)e; px.tiles[k] = (uint8_t)tiles;
        long long most = 0;
        for (int q = 0; q < tiles; ++q, ++e) {
            const SpmvDeviceMatrix& m = *parts[e];
            if (m.block_threads != threads || m.n_groups < 0) return hipErrorInvalidValue;
            most = std::max<long long>(most, m.n_slices > 0 ? m.n_groups : 0);
            lds = std::max(lds, ((size_t)m.lds_floats + (size_t)m.ytile_floats * (threads / 64)) * sizeof(float));
        }
        if (tiles == 1) g += most;
        else { const int per = 8 / tiles; g += 8 * ((most + per - 1) / per); }
    }
    if (e != n) return hipErrorInvalidValue;
    px.begin[n_items] = g;
    if (g > 0x7fffffffLL) return hipErrorInvalidValue;
    static bool raised = false;
    if (!raised) {
        hipError_t err;
        if ((err = hipFuncSetAttribute((const void*)spmv_slices_multi_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256)) != hipSuccess) return err;
        if ((err = hipFuncSetAttribute((const void*)spmv_slices_multi_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256)) != hipSuccess) return err;
        raised = true;
    }
    bool strays = false;          // (a launch with a stray-slot part takes the instantiation that fetches them; hispmv_batch.cpp keeps such parts in grids of their own)
    for (int i = 0; i < n; ++i) strays = strays || parts[i]->has_strays;
    if (g > 0 && strays) hipLaunchKernelGGL(spmv_slices_multi_kernel<true>, dim3((unsigned)g), dim3(threads), lds, stream, d_table, px, alpha);
    else if (g > 0) hipLaunchKernelGGL(spmv_slices_multi_kernel<false>, dim3((unsigned)g), dim3(threads), lds, stream, d_table, px, alpha);
    return hipGetLastError();
}

hipError_t launch_merge_parts(float* y, const float* parts, int n_parts, int64_t part_stride, int32_t rows, int nv,
                              int64_t y_stride, int64_t vec_stride, hipStream_t stream) {
    (void)hipGetLastError();
    if (n_parts <= 0 || rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(spmv_merge_parts_kernel, dim3((unsigned)((rows + 255) / 256), (unsigned)nv), dim3(256), 0, stream,
                       y, parts, n_parts, (long long)part_stride, rows, (long long)y_stride, (long long)vec_stride);
    return hipGetLastError();
}

hipError_t launch_merge_multi(const int32_t* rows, int n, const MultiMergeEntry* d_table, hipStream_t stream) {
    (void)hipGetLastError();
    if (n <= 0) return hipSuccess;
    if (n > kMultiMax) return hipErrorInvalidValue;
    MultiPrefix px{};
    px.n = n;
    long long b = 0;
    for (int i = 0; i < n; ++i) { px.begin[i] = b; b += (rows[i] + 255) / 256; }
    px.begin[n] = b;
    if (b > 0) hipLaunchKernelGGL(spmv_merge_multi_kernel, dim3((unsigned)b), dim3(256), 0, stream, d_table, px);
    return hipGetLastError();
}

hipError_t launch_fixup_long(const SpmvDeviceMatrix& m, float* y, float alpha, hipStream_t stream) {
    (void)hipGetLastError();
    if (m.n_fix_long > 0)
        hipLaunchKernelGGL(spmv_fixup_long_kernel, dim3((m.n_fix_long + 3) / 4), dim3(256), 0, stream, m.fix_long, m.n_fix_long, m.carry, y, alpha, 0LL, 0LL);
    return hipGetLastError();
}

hipError_t launch_tail_multi(const int32_t* fix_counts, int n_fix, const MultiFixEntry* d_fix_table, const int32_t* merge_rows, int n_merge,
                             const TailMergeEntry* d_merge_table, float alpha, hipStream_t stream) {
    (void)hipGetLastError();
    if (n_fix > kMultiMax || n_merge > kMultiMax) return hipErrorInvalidValue;
    MultiPrefix fx{}, mx{};
    fx.n = n_fix; mx.n = n_merge;
    long long fb = 0, mb = 0;
    for (int i = 0; i < n_fix; ++i) { fx.begin[i] = fb; fb += (fix_counts[i] + 255) / 256; }
    fx.begin[n_fix] = fb;
    for (int i = 0; i < n_merge; ++i) { mx.begin[i] = mb; mb += (merge_rows[i] + 255) / 256; }
    mx.begin[n_merge] = mb;
    if (fb + mb > 0x7fffffffLL) return hipErrorInvalidValue;
    if (fb + mb > 0) hipLaunchKernelGGL(spmv_tail_multi_kernel, dim3((unsigned)(fb + mb)), dim3(256), 0, stream, d_fix_table, fx, d_merge_table, mx, alpha);
    return hipGetLastError();
}

hipError_t launch_fixup_multi(const SpmvDeviceMatrix* const* parts, float* const* ys, int n, const MultiFixEntry* d_fix_table,
                              float alpha, hipStream_t stream) {
    (void)hipGetLastError();
    if (n <= 0) return hipSuccess;
    if (n > kMultiMax) return hipErrorInvalidValue;
    MultiPrefix fx{};
    fx.n = n;
    long long fb = 0;
    for (int i = 0; i < n; ++i) { fx.begin[i] = fb; fb += (parts[i]->n_fix_short + 255) / 256; }
    fx.begin[n] = fb;
    if (fb > 0) hipLaunchKernelGGL(spmv_fixup_multi_kernel, dim3((unsigned)fb), dim3(256), 0, stream, d_fix_table, fx, alpha);
    for (int i = 0; i < n; ++i) {
        const SpmvDeviceMatrix& m = *parts[i];
        if (m.n_fix_long > 0)
            hipLaunchKernelGGL(spmv_fixup_long_kernel, dim3((m.n_fix_long + 3) / 4), dim3(256), 0, stream,
                               m.fix_long, m.n_fix_long, m.carry, ys[i], alpha, 0LL, 0LL);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Transposed tile stream (hispmv_tts.h): one workgroup of 16 wavefronts per row tile.
//   LDS: [row accumulators of the tile][staging: products of the block in flight, by row-major slot][chunk tails]
//   per block  phase A: the block's words in COLUMN order, 1024 per slice (values + {col_off:16|slot:16} metas, one
//                       dwordx4 each per lane and step, requested one slice ahead): x gathered through a buffer
//                       descriptor with the slice's column base as scalar offset -- neighbouring lanes read
//                       neighbouring columns, a few cache lines per gather --, product -> staging[slot]
//              barrier
//              phase B: the staging in ROW-MAJOR order, 1024 slots per chunk (ds_read_b128), row ends from 16 flag bits
//                       per lane, scan_step as in the slice kernel, acc[row] += total by the lane holding the row end
//                       (rows are distinct inside a block pass); the open tail of a chunk goes to tails[chunk]
//              barrier, then one wavefront adds the tails of rows cut by chunk boundaries (in chunk order)
//   end: y = alpha*acc + beta*bias, coalesced.  No atomics, fixed summation order.
// ---------------------------------------------------------------------------
struct TtsSlice { uint4 v[kSliceSteps]; uint4 m[kSliceSteps]; };
__device__ __forceinline__ void tts_request(TtsSlice& s, const char* words, int slice, int lane) {
    const uint4* pv = (const uint4*)(words + (size_t)slice * (kSliceElems * 8)) + lane;
#pragma unroll
    for (int j = 0; j < kSliceSteps; ++j) s.v[j] = load_words(pv + j * 64);
#pragma unroll
    for (int j = 0; j < kSliceSteps; ++j) s.m[j] = load_words(pv + 256 + j * 64);
}

// NV > 1: NV input vectors per pass over the tile's words (FpgaHandle::linear with num_vecs > 1 on a matrix whose tiles are
// small enough: NV copies of the accumulators and of the staging fit the LDS).  Inside phase A a wavefront takes its slice
// through the vectors one after the other -- gathers from x + v*cols, products into staging area v -- and phase B reduces
// every staging area into accumulator set v: the barriers, the slice loads, the flag words and the latency chain of a block
// are shared by the NV vectors, and every vector sees exactly the arithmetic of the single-vector kernel (same bits).
// The staging areas are `stage_stride` floats apart (the matrix's largest block rounded up to whole chunks, plus the dummy
// slot padding words write to: their slot number -- the geometry's maximum -- is clamped to it).
// XLDS: x is short enough (M.xlds_floats = cols rounded up, <= kTtsXldsMax) to sit in the LDS next to the accumulators and the
// staging: the workgroup copies it there once (coalesced) and phase A gathers with ds_read_b32 -- the vector cache takes the
// lanes of a scattered global gather one per cycle or two, the LDS 32 per cycle.  The wide, short layers of
// apps/model_test.py (1024 x 8192).  Staging areas are sized by the matrix's largest block (batch_stage_floats), as for NV > 1.
template <bool HAS_BETA, bool ZERO_FILL = false, int NV = 1, bool XLDS = false, bool GAP = false>
__device__ __forceinline__ void tts_tile_body(const TtsDeviceMatrix& M, const float* __restrict__ x, const float* bias, float* y,
                                              float alpha, float beta, int tile_index, const unsigned tid) {
    extern __shared__ float xs[];
    float* const acc0 = xs;
    float* const staging0 = xs + M.acc_floats * NV;
    const int stage_stride = (NV > 1 || XLDS) ? M.batch_stage_floats : M.staging_floats;
    float* const tails = staging0 + stage_stride * NV;
    float* const xw0 = tails + 64 * NV;                      // XLDS: NV copies of x, xlds_floats apart
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);     // provably wave-uniform: table reads become scalar loads
    const int n_waves = blockDim.x >> 6;
    constexpr int kE = kSliceSteps * kLaneElems;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, M.cols * NV * 4, 0x00020000);
    const char* const words = (const char*)M.words;
    const int4 tile = load_int4(M.tiles + tile_index);
    const int row0 = tile.x, n_rows = tile.y, block_begin = tile.z, n_blocks = tile.w;      // row0 < 0: carry tile (one row)

    // the first two slices a wavefront has in the first block are requested before anything else: a wavefront keeps TWO
    // slice buffers (slices `wave` and `wave + n_waves` of a block) and both are requested one block ahead, so phase A
    // never waits for HBM (with one buffer the second slice of every block was requested and waited for inside phase A)
    int4 blk = load_int4(M.blocks + 2 * (size_t)block_begin);
    TtsSlice wA, wB;
    int cbA = 0, cbB = 0;
    if (n_blocks > 0) {
        if (wave < blk.y) { tts_request(wA, words, blk.x + wave, lane); cbA = *(const HISPMV_GLOBAL int*)(M.col_base + blk.x + wave); }
        if (wave + n_waves < blk.y) { tts_request(wB, words, blk.x + wave + n_waves, lane); cbB = *(const HISPMV_GLOBAL int*)(M.col_base + blk.x + wave + n_waves); }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v)
        for (int i = tid; i < n_rows; i += blockDim.x) acc0[v * M.acc_floats + i] = 0.0f;
    if (XLDS) {
        // all loads of a thread in flight, then the LDS writes (a load-store loop waited out one L2 round trip per 1024 floats)
        const HISPMV_GLOBAL float* xg = (const HISPMV_GLOBAL float*)x;
        constexpr int kU = kTtsXldsMax / 1024;           // dwords per thread of a 1024-thread workgroup
        for (int v = 0; v < NV; ++v) {
            float t[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) { const int i = (int)tid + u * (int)blockDim.x; t[u] = i < M.cols ? xg[(size_t)v * M.cols + i] : 0.0f; }
#pragma unroll
            for (int u = 0; u < kU; ++u) { const int i = (int)tid + u * (int)blockDim.x; if (i < M.xlds_floats) xw0[v * M.xlds_floats + i] = t[u]; }
        }
    }
    // zero-fill geometry (hispmv_tts.h): rows absent from a block own a slot of its row-major order but no stream word --
    // the staging is all zero whenever a phase A starts: zeroed here, and phase B writes zeros back over what it has read
    // (a template parameter: as a run-time flag its zero vector lived across the block loop and the multi-matrix kernel spilled)
    if (ZERO_FILL)
        for (int i = tid; i < ((stage_stride * NV) >> 2); i += blockDim.x) ((float4*)staging0)[i] = float4{0.0f, 0.0f, 0.0f, 0.0f};
    __syncthreads();

    // phase A for one slice: gathers, products -> staging, the request that reuses the buffer (next >= 0)
    auto phase_a = [&](TtsSlice& w, int& cb, int next) {
        const int base0 = __builtin_amdgcn_readfirstlane(cb) << 2;
        const unsigned dummy = (unsigned)stage_stride - 64u;      // (NV > 1: where padding words go in a staging area)
#pragma unroll 1
        for (int v = 0; v < NV; ++v) {      // (not unrolled: the gathers of several vectors in flight at once spilled registers)
        const int base = base0 + v * (M.cols << 2);
        float* const staging = staging0 + v * stage_stride;
        float xv[kE];
        if (XLDS) {
            const float* const xw = xw0 + v * M.xlds_floats + (base0 >> 2);
#pragma unroll
            for (int j = 0; j < kSliceSteps; ++j) {
                xv[4 * j + 0] = xw[w.m[j].x >> 16]; xv[4 * j + 1] = xw[w.m[j].y >> 16];
                xv[4 * j + 2] = xw[w.m[j].z >> 16]; xv[4 * j + 3] = xw[w.m[j].w >> 16];
            }
        } else {
#if HISPMV_TTS_EXPERIMENT == 1            // timing experiment (tools/experiments/r4_tts_phases.sh): no x gathers -- the results are wrong
#pragma unroll
        for (int j = 0; j < kE; ++j) xv[j] = 1.0f;
        (void)rx; (void)base;
#else
#pragma unroll
        for (int j = 0; j < kSliceSteps; ++j) {
            xv[4 * j + 0] = i2f((int)__builtin_amdgcn_raw_buffer_load_b32(rx, (w.m[j].x >> 16) << 2, base, 0));
            xv[4 * j + 1] = i2f((int)__builtin_amdgcn_raw_buffer_load_b32(rx, (w.m[j].y >> 16) << 2, base, 0));
            xv[4 * j + 2] = i2f((int)__builtin_amdgcn_raw_buffer_load_b32(rx, (w.m[j].z >> 16) << 2, base, 0));
            xv[4 * j + 3] = i2f((int)__builtin_amdgcn_raw_buffer_load_b32(rx, (w.m[j].w >> 16) << 2, base, 0));
        }
#endif
        }
        // products -> staging[slot] (the transposition), THEN the request that reuses the buffer: the slots are read from
        // the buffer itself (16 registers less than keeping them across the request; two buffers + 16 products fit 128)
#pragma unroll
        for (int j = 0; j < kSliceSteps; ++j) {
            if (NV > 1 || XLDS) {
                staging[min(w.m[j].x & 0xffffu, dummy)] = i2f((int)w.v[j].x) * xv[4 * j + 0];
                staging[min(w.m[j].y & 0xffffu, dummy)] = i2f((int)w.v[j].y) * xv[4 * j + 1];
                staging[min(w.m[j].z & 0xffffu, dummy)] = i2f((int)w.v[j].z) * xv[4 * j + 2];
                staging[min(w.m[j].w & 0xffffu, dummy)] = i2f((int)w.v[j].w) * xv[4 * j + 3];
            } else {
                staging[w.m[j].x & 0xffffu] = i2f((int)w.v[j].x) * xv[4 * j + 0];
                staging[w.m[j].y & 0xffffu] = i2f((int)w.v[j].y) * xv[4 * j + 1];
                staging[w.m[j].z & 0xffffu] = i2f((int)w.v[j].z) * xv[4 * j + 2];
                staging[w.m[j].w & 0xffffu] = i2f((int)w.v[j].w) * xv[4 * j + 3];
            }
        }
        }
        asm volatile("" ::: "memory");
        if (next >= 0) {
            tts_request(w, words, next, lane);
            cb = *(const HISPMV_GLOBAL int*)(M.col_base + next);
        }
    };
    // phase B for one chunk of 1024 staged products in row-major order
    // GAP (TtsGeometry::gap_rows): a row end carries a 2-bit code = distance to the previous slot-owning row (`ends` bit 0, `ends_hi`
    // bit 1); its accumulator is ci.x + the running SUM of the codes, instead of ci.x + the running count of row ends
    // (GAP: `ends` holds both planes, bit 1 of the codes in its upper half -- one register, one load: M.flags_hi is the interleaved array)
    auto phase_b = [&](int c, int2 ci, unsigned ends) {
#pragma unroll 1
        for (int v = 0; v < NV; ++v) {
        float* const acc = acc0 + v * M.acc_floats;
        int row = __builtin_amdgcn_readfirstlane(ci.x);
        const float4* st4 = (const float4*)(staging0 + v * stage_stride + c * kTtsChunkSlots) + lane;
        float carry_step = 0.0f;
#pragma unroll
        for (int j = 0; j < kSliceSteps; ++j) {
            const float4 q = st4[j * 64];
            if (ZERO_FILL) ((float4*)st4)[j * 64] = float4{0.0f, 0.0f, 0.0f, 0.0f};
            const float pj[kLaneElems] = {q.x, q.y, q.z, q.w};
            bool e[kLaneElems];
            int code[kLaneElems];          // GAP: 1..3 at a row end, 0 elsewhere
            int below = 0, total = 0;
#pragma unroll
            for (int k = 0; k < kLaneElems; ++k) {
                if (GAP) {
                    const bool lo = ((ends >> (4 * j + k)) & 1u) != 0, hi = ((ends >> (16 + 4 * j + k)) & 1u) != 0;
                    e[k] = lo | hi;
                    code[k] = (lo ? 1 : 0) + (hi ? 2 : 0);
                    const unsigned long long ml = __builtin_amdgcn_ballot_w64(lo), mh = __builtin_amdgcn_ballot_w64(hi);
                    below += lanes_below(ml) + 2 * lanes_below(mh);
                    total += __builtin_popcountll(ml) + 2 * __builtin_popcountll(mh);
                } else {
                    e[k] = ((ends >> (4 * j + k)) & 1u) != 0;
                    const unsigned long long mk = __builtin_amdgcn_ballot_w64(e[k]);
                    below += lanes_below(mk);
                    total += __builtin_popcountll(mk);
                }
            }
            float tj[kLaneElems];
            scan_step(pj, e[0], e[1], e[2], e[3], carry_step, tj);
            // the row ends of a step are distinct rows (and a row ends once per block): the four read-modify-writes
            // are independent -- all reads, then all writes, one LDS round trip instead of four
            int r[kLaneElems];
            float a[kLaneElems];
            if (GAP) {
                r[0] = row + below + code[0];
#pragma unroll
                for (int k = 1; k < kLaneElems; ++k) r[k] = r[k - 1] + code[k];
            } else {
            r[0] = row + below;
#pragma unroll
            for (int k = 1; k < kLaneElems; ++k) r[k] = r[k - 1] + (e[k - 1] ? 1 : 0);
            }
#pragma unroll
            for (int k = 0; k < kLaneElems; ++k) a[k] = e[k] ? acc[r[k]] : 0.0f;
#pragma unroll
            for (int k = 0; k < kLaneElems; ++k)
                if (e[k]) acc[r[k]] = a[k] + tj[k];
            row += total;
        }
        if (lane == 0) tails[v * 64 + c] = carry_step;
        }
    };

    for (int b = 0; b < n_blocks; ++b) {
        const int slice_begin = __builtin_amdgcn_readfirstlane(blk.x), n_slices = __builtin_amdgcn_readfirstlane(blk.y);
        const int chunk_begin = __builtin_amdgcn_readfirstlane(blk.z), n_chunks = __builtin_amdgcn_readfirstlane(blk.w);
        int4 nxt = int4{0, 0, 0, 0};
        if (b + 1 < n_blocks) nxt = load_int4(M.blocks + 2 * (size_t)(block_begin + b + 1));
        // what phase B and the tail pass of THIS block read from memory leaves now: it is older than the gathers, so it has
        // arrived when they have (loaded behind the barrier, every wavefront sat out a memory latency there -- and, the
        // loads counting in order, the arrival of the next block's slices as well)
        int2 ciA = int2{0, 0}, ciB = int2{0, 0}, ciT = int2{0, 0};
        unsigned endsA = 0, endsB = 0;
        auto load_ends = [&](int c) -> unsigned {
            if (GAP) return *(const HISPMV_GLOBAL unsigned*)((const unsigned*)M.flags_hi + (size_t)(chunk_begin + c) * 64 + lane);
            return *(const HISPMV_GLOBAL unsigned short*)(M.flags + (size_t)(chunk_begin + c) * 64 + lane);
        };
        if (wave < n_chunks) {
            ciA = load_int2(M.chunk_info + chunk_begin + wave_u);
            endsA = load_ends(wave);
        }
        if (wave + n_waves < n_chunks) {
            ciB = load_int2(M.chunk_info + chunk_begin + wave_u + n_waves);
            endsB = load_ends(wave + n_waves);
        }
        if (wave == 0 && lane < n_chunks) ciT = load_int2(M.chunk_info + chunk_begin + lane);
        // ---- phase A: column order --------------------------------------------------------------------------------
        const bool more = wave + 2 * n_waves < n_slices;       // a third slice in this block (partly filled slices): rare
        if (wave < n_slices) phase_a(wA, cbA, wave < nxt.y ? nxt.x + wave : -1);
        else if (wave < nxt.y) { tts_request(wA, words, nxt.x + wave, lane); cbA = *(const HISPMV_GLOBAL int*)(M.col_base + nxt.x + wave); }
        if (wave + n_waves < n_slices) {
            phase_a(wB, cbB, more ? slice_begin + wave + 2 * n_waves : (wave + n_waves < nxt.y ? nxt.x + wave + n_waves : -1));
            for (int s = wave + 2 * n_waves; s < n_slices; s += n_waves)
                phase_a(wB, cbB, s + n_waves < n_slices ? slice_begin + s + n_waves : (wave + n_waves < nxt.y ? nxt.x + wave + n_waves : -1));
        } else if (wave + n_waves < nxt.y) {
            tts_request(wB, words, nxt.x + wave + n_waves, lane); cbB = *(const HISPMV_GLOBAL int*)(M.col_base + nxt.x + wave + n_waves);
        }
        __syncthreads();
        // ---- phase B: row-major order -----------------------------------------------------------------------------
#if HISPMV_TTS_EXPERIMENT != 2            // (2: timing experiment without the row-order pass -- the results are wrong)
        if (wave < n_chunks) phase_b(wave, ciA, endsA);
        if (wave + n_waves < n_chunks) phase_b(wave + n_waves, ciB, endsB);
        for (int c = wave + 2 * n_waves; c < n_chunks; c += n_waves)
            phase_b(c, load_int2(M.chunk_info + chunk_begin + c), load_ends(c));
#else
        (void)endsA; (void)endsB; (void)ciA; (void)ciB;
#endif
        __syncthreads();
        // rows cut by a chunk boundary: the chunk that holds the row end added its own part; the tails of the chunks
        // before it follow here, in chunk order (one lane per chunk; a block has at most 48 chunks)
        if (wave == 0 && lane < n_chunks) {
            // (GAP: chunk_info = {last slot-owning row before the chunk, chain | code of the chunk's first row end << 16})
            const int t_chain = GAP ? (ciT.y & 0xffff) : ciT.y;
            const int t_row = GAP ? ciT.x + (ciT.y >> 16) : ciT.x;
            if (t_chain > 0) {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    float* const acc = acc0 + v * M.acc_floats;
                    float s = 0.0f;
                    for (int k = lane - t_chain; k < lane; ++k) s += tails[v * 64 + k];
                    acc[t_row] = acc[t_row] + s;
                }
            }
        }
        blk = nxt;
        // (the next phase A writes staging only; tails and acc are next touched behind the barrier that follows it)
    }
    __syncthreads();
    if (row0 < 0) {         // a piece of a long row: its raw sum waits in carry[] for the fix-up launch (vector v: carry + v * n_carry)
        if ((int)tid < NV) *(HISPMV_GLOBAL float*)(M.carry + (size_t)tid * M.n_carry + (-row0 - 1)) = acc0[tid * M.acc_floats];
        return;
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const float* const acc = acc0 + v * M.acc_floats;
        float* const yv = y + (size_t)v * M.rows;
        for (int i = tid; i < n_rows; i += blockDim.x) {
            const float t = acc[i];
            if (HAS_BETA) *(HISPMV_GLOBAL float*)(yv + row0 + i) = alpha * t + beta * *(const HISPMV_GLOBAL float*)(bias + row0 + i);
            else *(HISPMV_GLOBAL float*)(yv + row0 + i) = alpha * t;
        }
    }
}

// NV vectors per pass over the words (tts_tile_body<., ., NV>), x through the cache or from the LDS
template <bool HAS_BETA, bool ZERO_FILL = false, int NV = 1, bool XLDS = false, bool GAP = false>
__device__ __forceinline__ void tts_tile_body(const TtsDeviceMatrix& M, const float* __restrict__ x, const float* bias, float* y,
                                              float alpha, float beta, int tile_index) {
    tts_tile_body<HAS_BETA, ZERO_FILL, NV, XLDS, GAP>(M, x, bias, y, alpha, beta, tile_index, threadIdx.x);
}
template <bool HAS_BETA, int NV, bool XLDS>
__global__ __launch_bounds__(1024) void spmv_tts_nv_kernel(TtsDeviceMatrix M, const float* __restrict__ x, const float* bias, float* y,
                                                           float alpha, float beta) {
    tts_tile_body<HAS_BETA, false, NV, XLDS>(M, x, bias, y, alpha, beta, (int)blockIdx.x);
}

template <bool HAS_BETA>
__global__ __launch_bounds__(1024) void spmv_tts_kernel(TtsDeviceMatrix M, const float* __restrict__ x, const float* bias, float* y,
                                                        float alpha, float beta) {
    tts_tile_body<HAS_BETA>(M, x, bias, y, alpha, beta, (int)blockIdx.x);
}

// `nv` vectors in ONE launch (FpgaHandle::linear with num_vecs > 1; the reference relaunches its kernel per vector,
// fpga_handle.cpp:366-379): a workgroup takes its tile through the vectors one after the other -- vector v reads x + v*cols
// and writes y + v*rows, with exactly the arithmetic of the single-vector kernel (same bits).  The tile's words are read
// from HBM for the first vector and from L2 / the Infinity Cache for the others when the matrix fits there (the model
// layers do), and the launch latency is paid once.  (Keeping a block's slices in registers across the vectors instead
// -- two slice buffers live through phase B -- spilled 236 bytes per lane at the 128 registers a 16-wavefront workgroup has.)
template <bool HAS_BETA>
__global__ __launch_bounds__(1024) void spmv_tts_batched_kernel(TtsDeviceMatrix M, const float* __restrict__ x, const float* bias, float* y,
                                                                float alpha, float beta, int nv) {
#pragma unroll 1
    for (int v = 0; v < nv; ++v) {
        tts_tile_body<HAS_BETA>(M, x + (size_t)v * M.cols, bias, y + (size_t)v * M.rows, alpha, beta, (int)blockIdx.x);
        __syncthreads();          // the next vector's prologue zeroes the accumulators this one's epilogue reads
        // a carry tile of vector v writes carry[-row0 - 1]; the fix-up launch reads one set per vector
        M.carry += M.n_carry;
    }
}

template <bool ZERO_FILL, bool XLDS = false, bool GAP = false>
__global__ __launch_bounds__(1024) void spmv_tts_multi_kernel(const TtsEntry* __restrict__ table, MultiPrefix prefix, float alpha) {
    int k = 0;
#pragma unroll 1
    while (k + 1 < prefix.n && (long long)blockIdx.x >= prefix.begin[k + 1]) ++k;
    // An item with tiles[k] = 2 is the two column parts of one matrix (tall geometry, hispmv_tts.h): consecutive table
    // entries pinned to XCDs 0-3 / 4-7 (blockIdx mod 8, as the slice kernel's pinned column tiles), so that an XCD's L2
    // holds one part's half of x.
    int tile = (int)(blockIdx.x - prefix.begin[k]);
    int entry = prefix.first[k];
    const int parts = prefix.tiles[k];
    if (parts > 1) {
        const int per = 8 / parts, res = tile & 7;
        entry += res / per;
        tile = (tile >> 3) * per + (res % per);
    }
    const TtsEntry e = table[entry];      // (once per workgroup; every load in the body is cast to the global address space)
    if (tile >= e.m.n_tiles) return;      // (a part with fewer tiles than its sibling)
    WGT_BEGIN();
    if (e.beta != 0.0f) tts_tile_body<true, ZERO_FILL, 1, XLDS, GAP>(e.m, e.x, e.bias, e.y, alpha, e.beta, tile);
    else tts_tile_body<false, ZERO_FILL, 1, XLDS, GAP>(e.m, e.x, e.y, e.y, alpha, 0.0f, tile);
    WGT_END(3, entry, tile);
}

// A wave-uniform table read through the CONSTANT address space (dword by dword: scalar loads).
template <class T>
__device__ __forceinline__ T load_const(const T* p) {
    static_assert(sizeof(T) % 4 == 0, "dword-sized tables");
    typedef const __attribute__((address_space(4))) unsigned* CP;
    union U { T t; unsigned w[sizeof(T) / 4]; __device__ U() {} } u;
    CP src = (CP)(const void*)p;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) u.w[i] = src[i];
    return u.t;
}

// ---------------------------------------------------------------------------
// The step kernel: ALL main work of a batch call -- slice groups of every workgroup size and tiles of every tile stream -- in
// ONE launch of one persistent 1024-thread workgroup per CU that draws ITEMS from a queue (a ticket counter in device memory)
// until the queue is empty.  An item = one group of a 1024-thread plan, FOUR consecutive groups of a 256-thread plan (four
// wavefronts each, their LDS side by side: SubBlock), or one tile.  Every item runs the very body of the multi-matrix kernels
// above (same instantiations, same arithmetic: same bits); what changes is who hands out the work:
//   * the dispatcher needs 1.4 - 3.4 us to replace a finished workgroup by the next (tools/wg_timeline.py: 7.6 changes per CU and
//     step, 15 us of every CU's 282) -- here the next item's ticket is drawn while the current item runs, a change costs a barrier;
//   * the queue is ordered by the host (longest items first, kinds mixed: hispmv_batch.cpp), so the step ends with short items
//     instead of with whatever the last grid's tail happens to be (9.4 us of idle per CU at the end of a step before);
//   * one launch on the caller's stream: no fork to / join from a side stream in front of the tail launch.
// The last workgroup to leave resets the counters (self-cleaning: graph replays and plain launches alike, no host state).
// ---------------------------------------------------------------------------
template <bool STRAYS>        // some slice part of the call has stray slots: the instantiation that fetches them (it serves parts without them too)
__global__ __launch_bounds__(1024) void spmv_step_kernel(StepArgs args) {
    // the ticket's LDS word: the LAST four bytes of the dynamic LDS, behind everything the items use.  (As a static __shared__ variable
    // it was placed FIRST and pushed the dynamic LDS to offset 4: every 16-byte LDS access of the bodies -- window staging, the tiles'
    // ds_read_b128 -- became misaligned and the tiles ran 10 % slower than in the grids.)
    extern __shared__ float step_lds[];
    unsigned& s_next = *(unsigned*)(step_lds + args.ticket_word);
    if (threadIdx.x == 0) s_next = __hip_atomic_fetch_add(args.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
#pragma unroll 1
    for (;;) {
        // Nothing of the queue lives in registers across an item (the tile body runs at the 128-VGPR ceiling of a 1024-thread
        // workgroup and within two SGPRs of theirs: loop state carried across it spilled 187 SGPRs and 12 VGPRs to scratch):
        // the arguments are read again from the kernel-argument segment at the top of every item (scalar loads that hit the
        // constant cache) and the ticket travels through the LDS.
        typedef const __attribute__((address_space(4))) StepArgs* KernArgs;
        KernArgs ap = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ap));
        const StepArgs a{ap->slice_table, ap->tts_table, ap->items, ap->sync, ap->n_items, ap->alpha, 0, ap->ticket_word};
        unsigned& s_next = *(unsigned*)(step_lds + a.ticket_word);
        const unsigned it = s_next;
        if (it >= a.n_items) break;
        __syncthreads();                    // every thread has read the ticket: its slot may take the next one
        // the NEXT item's ticket is drawn first thing: its round trip runs under the loads of this item's descriptor and table entry
        // (a tile parks it in the LDS before its body starts -- no register to hold it across -- and by then most of the wait is over)
        unsigned next = 0;
        if (threadIdx.x == 0) next = __hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int2 item = load_const(a.items + __builtin_amdgcn_readfirstlane((int)it));
        const unsigned kind = (unsigned)__builtin_amdgcn_readfirstlane(item.x) & 3u;
        const int entry = (int)((unsigned)__builtin_amdgcn_readfirstlane(item.x) >> 8);
        const long long index = (long long)(unsigned)__builtin_amdgcn_readfirstlane(item.y);
        // the thread index, opaque per item: everything the bodies derive from it (lane offsets, masks) would otherwise be
        // hoisted out of this loop, held across the items and -- at the tile body's register ceiling -- spilled to scratch, with
        // the reloads (and their vmcnt(0)) right behind the slice requests
        unsigned tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
        WGT_BEGIN();
        if (kind == 2) {
            // a tile: the next ticket is parked in the LDS BEFORE the tile starts (no register to hold it across the body)
            if (tid == 0) s_next = next;
            // (the tables through the CONSTANT address space: scalar loads -- as generic pointers read from memory they became
            // flat loads into some forty VGPRs)
            const TtsEntry e = load_const(a.tts_table + entry);
            if (e.beta != 0.0f) tts_tile_body<true>(e.m, e.x, e.bias, e.y, a.alpha, e.beta, (int)index, tid);
            else tts_tile_body<false>(e.m, e.x, e.y, e.y, a.alpha, 0.0f, (int)index, tid);
        } else {
            // slice groups (91 - 97 VGPRs): the next ticket stays in flight while the item runs and is consumed behind it
            const MultiEntry t = load_const(a.slice_table + entry);
            const LookbackArgs lb{};
            // kind 0: the workgroup is the group's; kind 1: four groups of a 256-thread plan, wavefronts 4q .. 4q+3 take group index + q
            const SubBlock sb = kind == 0 ? SubBlock{wave, 16, 0, (int)(tid & 63)} : SubBlock{wave & 3, 4, (wave >> 2) * (t.lds_floats + 4 * t.ytile_floats), (int)(tid & 63)};
            const long long group = kind == 0 ? index : index + (wave >> 2);
            if (t.beta != 0.0f)
                slices_body<true, true, false, STRAYS>((const char*)t.words, t.hdr, t.groups, t.frags, t.x, t.bias, t.y, t.carry, a.alpha, t.beta,
                                                       t.n_slices, t.group_slices, t.lds_floats, t.ytile_floats, t.cols, t.rows, lb, group, sb);
            else
                slices_body<false, true, false, STRAYS>((const char*)t.words, t.hdr, t.groups, t.frags, t.x, t.y, t.y, t.carry, a.alpha, 0.0f,
                                                        t.n_slices, t.group_slices, t.lds_floats, t.ytile_floats, t.cols, t.rows, lb, group, sb);
            if (tid == 0) s_next = next;
        }
        // the item's LDS is free and the next ticket visible behind this barrier
#ifdef HISPMV_WG_TRACE
        WGT_END(kind == 2 ? 3 : kind == 0 ? 1 : 2, entry, index);        // (contains the barrier)
#else
        __syncthreads();
#endif
    }
    // every workgroup draws exactly one ticket past the end; the last one out rearms the queue for the next launch
    if (threadIdx.x == 0) {
        unsigned* const sync = ((const __attribute__((address_space(4))) StepArgs*)__builtin_amdgcn_kernarg_segment_ptr())->sync;
        const unsigned done = __hip_atomic_fetch_add(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1) {
            __hip_atomic_store(sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int NV, bool XLDS>
static hipError_t launch_tts_nv2(const TtsDeviceMatrix& m, const float* x, const float* bias, float* y, float alpha, float beta, hipStream_t stream);

static size_t tts_lds_bytes(const TtsDeviceMatrix& m) { return ((size_t)m.acc_floats + (size_t)m.staging_floats + 64) * sizeof(float); }

hipError_t launch_tts(const TtsDeviceMatrix& m, const float* x, const float* bias, float* y, float alpha, float beta, hipStream_t stream) {
    (void)hipGetLastError();
    if (m.n_tiles <= 0) return hipSuccess;
    static bool raised = false;
    if (!raised) {
        hipError_t e;
        if ((e = hipFuncSetAttribute((const void*)spmv_tts_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256)) != hipSuccess) return e;
        if ((e = hipFuncSetAttribute((const void*)spmv_tts_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256)) != hipSuccess) return e;
        raised = true;
    }
    const size_t lds = tts_lds_bytes(m);
    if (lds > 160 * 1024 - 256 || m.zero_fill) return hipErrorInvalidValue;      // (zero-fill geometries have column parts: launch_tts_multi)
    if (tts_x_in_lds(m, 1)) {           // a short x: gathered from the LDS
        hipError_t e = launch_tts_nv2<1, true>(m, x, bias, y, alpha, beta, stream);
        if (e != hipSuccess) return e;
    } else
    if (beta != 0.0f) hipLaunchKernelGGL(spmv_tts_kernel<true>, dim3((unsigned)m.n_tiles), dim3(m.threads), lds, stream, m, x, bias, y, alpha, beta);
    else hipLaunchKernelGGL(spmv_tts_kernel<false>, dim3((unsigned)m.n_tiles), dim3(m.threads), lds, stream, m, x, y, y, alpha, beta);
    if (m.n_fix > 0)
        hipLaunchKernelGGL(spmv_fixup_short_kernel, dim3((m.n_fix + 255) / 256), dim3(256), 0, stream, m.fix, m.n_fix, m.carry, y, alpha, 0LL, 0LL);
    return hipGetLastError();
}

// LDS of the NV-vector / x-in-LDS kernels: per vector the accumulators, one staging area (largest block of the matrix), 64
// tails and -- xlds -- a copy of x
static size_t tts_nv_lds_bytes(const TtsDeviceMatrix& m, int nv, bool xlds) {
    return ((size_t)m.acc_floats + (size_t)m.batch_stage_floats + 64 + (xlds ? (size_t)m.xlds_floats : 0)) * nv * sizeof(float);
}
bool tts_x_in_lds(const TtsDeviceMatrix& m, int nv) {
    return !m.zero_fill && m.xlds_floats > 0 && m.batch_stage_floats > 0 && tts_nv_lds_bytes(m, nv, true) <= 160 * 1024 - 256;
}
int tts_batch_width(const TtsDeviceMatrix& m, int64_t vecs) {
    if (vecs < 2 || m.zero_fill || m.batch_stage_floats <= 0) return 1;
    for (int nv = 4; nv >= 2; nv >>= 1)         // x from the LDS first: fewer vectors per pass, but no gather through the cache
        if (nv <= vecs && tts_x_in_lds(m, nv)) return nv;
    for (int nv = 4; nv >= 2; nv >>= 1) {
        if (nv > vecs) continue;
        if ((int64_t)m.cols * nv >= (1 << 30) || (int64_t)m.rows * nv >= (1 << 30)) continue;
        if (tts_nv_lds_bytes(m, nv, false) <= 160 * 1024 - 256) return nv;
    }
    return 1;
}

template <bool HAS_BETA, int NV, bool XLDS>
static hipError_t launch_tts_nv(const TtsDeviceMatrix& m, const float* x, const float* bias, float* y, float alpha, float beta, hipStream_t stream) {
    static bool raised = false;
    if (!raised) {
        hipError_t e = hipFuncSetAttribute((const void*)spmv_tts_nv_kernel<HAS_BETA, NV, XLDS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        if (e != hipSuccess) return e;
        raised = true;
    }
    const size_t lds = tts_nv_lds_bytes(m, NV, XLDS);
    hipLaunchKernelGGL((spmv_tts_nv_kernel<HAS_BETA, NV, XLDS>), dim3((unsigned)m.n_tiles), dim3(m.threads), lds, stream, m, x, HAS_BETA ? bias : y, y, alpha, beta);
    return hipGetLastError();
}
template <int NV, bool XLDS>
static hipError_t launch_tts_nv2(const TtsDeviceMatrix& m, const float* x, const float* bias, float* y, float alpha, float beta, hipStream_t stream) {
    return beta != 0.0f ? launch_tts_nv<true, NV, XLDS>(m, x, bias, y, alpha, beta, stream) : launch_tts_nv<false, NV, XLDS>(m, x, bias, y, alpha, beta, stream);
}

hipError_t launch_tts_batched(const TtsDeviceMatrix& m, int nv, const float* x, const float* bias, float* y, float alpha, float beta, hipStream_t stream) {
    (void)hipGetLastError();
    if (nv < 2 || nv > kTtsMaxVectors || m.zero_fill) return hipErrorInvalidValue;
    if (m.n_tiles <= 0) return hipSuccess;
    if ((nv == 2 || nv == 4) && tts_batch_width(m, nv) == nv) {        // the vectors share every pass over the words
        const bool xl = tts_x_in_lds(m, nv);
        hipError_t e = nv == 4 ? (xl ? launch_tts_nv2<4, true>(m, x, bias, y, alpha, beta, stream) : launch_tts_nv2<4, false>(m, x, bias, y, alpha, beta, stream))
                               : (xl ? launch_tts_nv2<2, true>(m, x, bias, y, alpha, beta, stream) : launch_tts_nv2<2, false>(m, x, bias, y, alpha, beta, stream));
        if (e != hipSuccess) return e;
        if (m.n_fix > 0)
            hipLaunchKernelGGL(spmv_fixup_short_kernel, dim3((m.n_fix + 255) / 256, nv), dim3(256), 0, stream, m.fix, m.n_fix, m.carry, y, alpha,
                               (long long)m.n_carry, (long long)m.rows);
        return hipGetLastError();
    }
    static bool raised = false;
    if (!raised) {
        hipError_t e;
        if ((e = hipFuncSetAttribute((const void*)spmv_tts_batched_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256)) != hipSuccess) return e;
        if ((e = hipFuncSetAttribute((const void*)spmv_tts_batched_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256)) != hipSuccess) return e;
        raised = true;
    }
    const size_t lds = tts_lds_bytes(m);
    if (lds > 160 * 1024 - 256) return hipErrorInvalidValue;
    if (beta != 0.0f) hipLaunchKernelGGL(spmv_tts_batched_kernel<true>, dim3((unsigned)m.n_tiles), dim3(m.threads), lds, stream, m, x, bias, y, alpha, beta, nv);
    else hipLaunchKernelGGL(spmv_tts_batched_kernel<false>, dim3((unsigned)m.n_tiles), dim3(m.threads), lds, stream, m, x, y, y, alpha, beta, nv);
    if (m.n_fix > 0)      // rows cut into pieces: one fix-up launch for all vectors (grid.y = vector)
        hipLaunchKernelGGL(spmv_fixup_short_kernel, dim3((m.n_fix + 255) / 256, nv), dim3(256), 0, stream, m.fix, m.n_fix, m.carry, y, alpha,
                           (long long)m.n_carry, (long long)m.rows);
    return hipGetLastError();
}

hipError_t launch_tts_multi(const TtsEntry* entries, int n, const uint8_t* item_parts, int n_items, const TtsEntry* d_table, float alpha,
                            hipStream_t stream) {
    (void)hipGetLastError();
    if (n <= 0) return hipSuccess;
    if (n > kMultiMax) return hipErrorInvalidValue;
    static bool raised = false;
    if (!raised) {
        hipError_t e;
        if ((e = hipFuncSetAttribute((const void*)spmv_tts_multi_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256)) != hipSuccess) return e;
        if ((e = hipFuncSetAttribute((const void*)spmv_tts_multi_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256)) != hipSuccess) return e;
        if ((e = hipFuncSetAttribute((const void*)spmv_tts_multi_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256)) != hipSuccess) return e;
        if ((e = hipFuncSetAttribute((const void*)spmv_tts_multi_kernel<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256)) != hipSuccess) return e;
        raised = true;
    }
    if (!item_parts) n_items = n;
    const bool xlds = tts_x_in_lds(entries[0].m, 1);       // the launch's matrices gather x from the LDS (a class of their own)
    MultiPrefix px{};
    px.n = n_items;
    long long g = 0;
    size_t lds = 0;
    int e = 0;
    for (int k = 0; k < n_items; ++k) {
        const int parts = item_parts ? item_parts[k] : 1;
        if ((parts != 1 && parts != 2 && parts != 4) || e + parts > n) return hipErrorInvalidValue;
        if (parts > 1) g = (g + 7) & ~7LL;                 // a pinned set starts at a multiple of 8
        px.begin[k] = g; px.first[k] = (uint8_t)e; px.tiles[k] = (uint8_t)parts;
        long long most = 0;
        for (int q = 0; q < parts; ++q, ++e) {
            most = std::max<long long>(most, entries[e].m.n_tiles);
            lds = std::max(lds, xlds ? tts_nv_lds_bytes(entries[e].m, 1, true) : tts_lds_bytes(entries[e].m));
            if (tts_x_in_lds(entries[e].m, 1) != xlds) return hipErrorInvalidValue;      // (one class per launch: hispmv_abi.cpp)
        }
        if (parts == 1) g += most;
        else { const int per = 8 / parts; g += 8 * ((most + per - 1) / per); }
    }
    if (e != n) return hipErrorInvalidValue;
    px.begin[n_items] = g;
    if (g > 0x7fffffffLL || lds > 160 * 1024 - 256) return hipErrorInvalidValue;
    for (int i = 1; i < n; ++i) if (entries[i].m.zero_fill != entries[0].m.zero_fill || entries[i].m.threads != entries[0].m.threads) return hipErrorInvalidValue;
    if (g > 0 && xlds) hipLaunchKernelGGL((spmv_tts_multi_kernel<false, true>), dim3((unsigned)g), dim3(entries[0].m.threads), lds, stream, d_table, px, alpha);
    else if (g > 0 && entries[0].m.zero_fill == 2) hipLaunchKernelGGL((spmv_tts_multi_kernel<false, false, true>), dim3((unsigned)g), dim3(entries[0].m.threads), lds, stream, d_table, px, alpha);
    else if (g > 0 && entries[0].m.zero_fill) hipLaunchKernelGGL(spmv_tts_multi_kernel<true>, dim3((unsigned)g), dim3(entries[0].m.threads), lds, stream, d_table, px, alpha);
    else if (g > 0) hipLaunchKernelGGL(spmv_tts_multi_kernel<false>, dim3((unsigned)g), dim3(entries[0].m.threads), lds, stream, d_table, px, alpha);
    return hipGetLastError();
}

hipError_t launch_spmv_step(const MultiEntry* d_slice_table, const TtsEntry* d_tts_table, const void* d_items, unsigned n_items,
                            unsigned* d_sync, int workgroups, size_t lds_bytes, bool strays, float alpha, hipStream_t stream) {
    (void)hipGetLastError();
    if (n_items == 0) return hipSuccess;
    if (workgroups <= 0 || lds_bytes > 160 * 1024 - 256) return hipErrorInvalidValue;
    static bool raised = false;
    if (!raised) {
        hipError_t e;
        if ((e = hipFuncSetAttribute((const void*)spmv_step_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
        if ((e = hipFuncSetAttribute((const void*)spmv_step_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
        raised = true;
    }
    // (+ 16 bytes behind the items' LDS for the ticket word)
    const size_t ticket_byte = (lds_bytes + 15) & ~(size_t)15;
    lds_bytes = ticket_byte + 16;
    if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
    const StepArgs a{d_slice_table, d_tts_table, (const int2*)d_items, d_sync, n_items, alpha, 0, (int)(ticket_byte / 4)};
    if (strays) hipLaunchKernelGGL(spmv_step_kernel<true>, dim3((unsigned)workgroups), dim3(1024), lds_bytes, stream, a);
    else hipLaunchKernelGGL(spmv_step_kernel<false>, dim3((unsigned)workgroups), dim3(1024), lds_bytes, stream, a);
    return hipGetLastError();
}
size_t tts_tile_lds_bytes(const TtsDeviceMatrix& m) { return tts_lds_bytes(m); }

// ---------------------------------------------------------------------------
// Dense overlay (reference: ComputeAB dense branch base_functions.cpp:188-226, packing
// spmv-helper.cpp:717-750).  No packing here: W stays row-major; one workgroup of 4 waves
// per group of R rows, 16 B per lane loads of W and x, DPP wave reduction, LDS across waves.
// ---------------------------------------------------------------------------
// NV input vectors per pass over W (linear with num_vecs > 1): vector v is x + v*cols, its result y + v*rows; every
// (row, vector) pair accumulates in exactly the order of the single-vector kernel, so the results are bitwise the same.
template <int R, bool HAS_BETA, int NV>
__device__ __forceinline__ void gemv_rows_body(const float* __restrict__ W, const float* __restrict__ x,
                                               const float* __restrict__ bias, float* __restrict__ y,
                                               int rows, int cols, float alpha, float beta, int block) {
    __shared__ float part[4][R][NV];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int row0 = block * R;
    float acc[R][NV];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[r][v] = 0.0f;

    if ((cols & 3) == 0) {
        const int n4 = cols >> 2;
        const float4* x4 = (const float4*)x;
        for (int c = threadIdx.x; c < n4; c += 256) {
            float4 xv[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) xv[v] = x4[(size_t)v * n4 + c];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int row = min(row0 + r, rows - 1);
                const float4 a = ((const float4*)(W + (size_t)row * cols))[c];
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[r][v] += a.x * xv[v].x + a.y * xv[v].y + a.z * xv[v].z + a.w * xv[v].w;
            }
        }
    } else {   // rows are not 16-byte aligned: dword path
        for (int c = threadIdx.x; c < cols; c += 256) {
            float xv[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) xv[v] = x[(size_t)v * cols + c];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int row = min(row0 + r, rows - 1);
                const float a = W[(size_t)row * cols + c];
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[r][v] += a * xv[v];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float s = wave_sum(acc[r][v]);
            if (lane == 0) part[wv][r][v] = s;
        }
    __syncthreads();
    if (threadIdx.x < R * NV) {
        const int r = threadIdx.x / NV, v = threadIdx.x % NV;
        const int row = row0 + r;
        if (row < rows) {
            const float s = (part[0][r][v] + part[1][r][v]) + (part[2][r][v] + part[3][r][v]);
            y[(size_t)v * rows + row] = HAS_BETA ? alpha * s + beta * bias[row] : alpha * s;
        }
    }
}

template <int R, bool HAS_BETA, int NV>
__global__ __launch_bounds__(256) void gemv_rows_kernel(const float* __restrict__ W, const float* __restrict__ x,
                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                        int rows, int cols, float alpha, float beta) {
    gemv_rows_body<R, HAS_BETA, NV>(W, x, bias, y, rows, cols, alpha, beta, (int)blockIdx.x);
}

// The dense handles of a batch call in one grid: block ranges per matrix (prefix.begin), largest matrix first so that the
// small ones fill the tail.
__global__ __launch_bounds__(256) void gemv_rows_multi_kernel(const GemvEntry* __restrict__ table, MultiPrefix prefix, float alpha) {
    int k = 0;
#pragma unroll 1
    while (k + 1 < prefix.n && (long long)blockIdx.x >= prefix.begin[k + 1]) ++k;
    const GemvEntry e = table[k];
    const int block = (int)((long long)blockIdx.x - prefix.begin[k]);
    if (e.beta != 0.0f) gemv_rows_body<4, true, 1>(e.W, e.x, e.bias, e.y, e.rows, e.cols, alpha, e.beta, block);
    else gemv_rows_body<4, false, 1>(e.W, e.x, e.bias, e.y, e.rows, e.cols, alpha, 0.0f, block);
}

template <int NV>
static void launch_gemv_nv(const float* W, int32_t rows, int32_t cols, const float* x, const float* bias,
                           float* y, float alpha, float beta, hipStream_t stream) {
    constexpr int R = 4;
    const unsigned blocks = (unsigned)((rows + R - 1) / R);
    if (beta != 0.0f)
        hipLaunchKernelGGL((gemv_rows_kernel<R, true, NV>), dim3(blocks), dim3(256), 0, stream, W, x, bias, y, rows, cols, alpha, beta);
    else
        hipLaunchKernelGGL((gemv_rows_kernel<R, false, NV>), dim3(blocks), dim3(256), 0, stream, W, x, bias, y, rows, cols, alpha, beta);
}

hipError_t launch_gemv(const float* W, int32_t rows, int32_t cols, const float* x, const float* bias,
                       float* y, float alpha, float beta, hipStream_t stream) {
    (void)hipGetLastError();   // the status returned below must be this launch's, not a stale one of the thread (e.g. PyTorch's pointer queries)
    if (rows <= 0) return hipSuccess;
    launch_gemv_nv<1>(W, rows, cols, x, bias, y, alpha, beta, stream);
    return hipGetLastError();
}

hipError_t launch_gemv_multi(const GemvEntry* entries, int n, const GemvEntry* d_table, float alpha, hipStream_t stream) {
    (void)hipGetLastError();
    if (n <= 0) return hipSuccess;
    if (n > kMultiMax) return hipErrorInvalidValue;
    MultiPrefix prefix{};
    prefix.n = n;
    long long total = 0;
    for (int i = 0; i < n; ++i) { prefix.begin[i] = total; total += (entries[i].rows + 3) / 4; }
    prefix.begin[n] = total;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(gemv_rows_multi_kernel, dim3((unsigned)total), dim3(256), 0, stream, d_table, prefix, alpha);
    return hipGetLastError();
}

hipError_t launch_gemv_batched(const float* W, int32_t rows, int32_t cols, int64_t vecs, const float* x, const float* bias,
                               float* y, float alpha, float beta, hipStream_t stream) {
    (void)hipGetLastError();   // the status returned below must be this launch's, not a stale one of the thread (e.g. PyTorch's pointer queries)
    if (rows <= 0) return hipSuccess;
    int64_t k = 0;
    while (k < vecs) {      // 8, 4, 2, 1 vectors per pass over W
        const float* xk = x + (size_t)k * cols;
        float* yk = y + (size_t)k * rows;
        if (vecs - k >= 8) { launch_gemv_nv<8>(W, rows, cols, xk, bias, yk, alpha, beta, stream); k += 8; }
        else if (vecs - k >= 4) { launch_gemv_nv<4>(W, rows, cols, xk, bias, yk, alpha, beta, stream); k += 4; }
        else if (vecs - k >= 2) { launch_gemv_nv<2>(W, rows, cols, xk, bias, yk, alpha, beta, stream); k += 2; }
        else { launch_gemv_nv<1>(W, rows, cols, xk, bias, yk, alpha, beta, stream); k += 1; }
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// A batch call replayed as a HIP graph (hispmv_abi.cpp) with another alpha: alpha is a by-value argument of every kernel of
// the call, so the instantiated graph is patched in place -- hipGraphExecKernelNodeSetParams on each of its kernel nodes --
// instead of being captured and instantiated again (a solver that changes alpha every step would pay an instantiation
// per step).  `graph` is the captured graph the executable was instantiated from (its node handles address the
// executable's nodes).  Any failure makes the caller fall back to a fresh capture.
// ---------------------------------------------------------------------------
hipError_t graph_set_alpha(hipGraphExec_t exec, hipGraph_t graph, float alpha) {
    size_t n = 0;
    hipError_t e = hipGraphGetNodes(graph, nullptr, &n);
    if (e != hipSuccess) return e;
    std::vector<hipGraphNode_t> nodes(n);
    if (n && (e = hipGraphGetNodes(graph, nodes.data(), &n)) != hipSuccess) return e;
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType type;
        if ((e = hipGraphNodeGetType(nodes[i], &type)) != hipSuccess) return e;
        if (type != hipGraphNodeTypeKernel) continue;
        hipKernelNodeParams p{};
        if ((e = hipGraphKernelNodeGetParams(nodes[i], &p)) != hipSuccess) return e;
        int idx, n_args;
        if (p.func == (void*)spmv_slices_multi_kernel<false> || p.func == (void*)spmv_slices_multi_kernel<true> || p.func == (void*)spmv_tts_multi_kernel<false> || p.func == (void*)spmv_tts_multi_kernel<true> ||
            p.func == (void*)spmv_tts_multi_kernel<false, true> || p.func == (void*)spmv_tts_multi_kernel<false, false, true> ||
            p.func == (void*)gemv_rows_multi_kernel ||
            p.func == (void*)spmv_fixup_multi_kernel) { idx = 2; n_args = 3; }
        else if (p.func == (void*)spmv_fixup_long_kernel) { idx = 4; n_args = 7; }
        else if (p.func == (void*)spmv_tail_multi_kernel) { idx = 4; n_args = 5; }
        else if (p.func == (void*)spmv_merge_multi_kernel) continue;          // no alpha
        else return hipErrorInvalidValue;                                      // a kernel this function does not know: do not guess
        if (!p.kernelParams) return hipErrorInvalidValue;
        void* args[8];
        for (int k = 0; k < n_args; ++k) args[k] = p.kernelParams[k];
        float a = alpha;
        args[idx] = &a;
        p.kernelParams = args;
        p.extra = nullptr;
        if ((e = hipGraphExecKernelNodeSetParams(exec, nodes[i], &p)) != hipSuccess) return e;
    }
    return hipSuccess;
}

// ---------------------------------------------------------------------------
// Multi-GPU boundary rows: the two tiny launches around the all-gather of tails (hispmv.h, hispmv_amd/dist.py).
// ---------------------------------------------------------------------------
__global__ void boundary_pack_kernel(const float* const* __restrict__ last, const float* __restrict__ mask,
                                     float* __restrict__ send, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) send[i] = last[i] ? mask[i] * *last[i] : 0.0f;
}

__global__ void boundary_apply_kernel(float* const* __restrict__ first, const float* __restrict__ recv,
                                      const float* __restrict__ weights, int n, int world) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !first[i]) return;
    float s = 0.0f;
    for (int r = 0; r < world; ++r) s += recv[(size_t)r * n + i] * weights[(size_t)i * world + r];
    *first[i] += s;
}

// Host-vector path (hispmv_run_kernel / hispmv_linear with small vectors): x and bias from the pinned staging block (its device
// address) into device memory -- a kernel instead of a DMA copy, whose engine costs ~10 us per call whatever the size.
__global__ __launch_bounds__(256) void fetch_vectors_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int n4) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i < n4) dst[i] = src[i];
}
hipError_t launch_fetch_vectors(const float* src, float* dst, int64_t n_floats, hipStream_t stream) {
    const int n4 = (int)((n_floats + 3) / 4);
    if (n4 <= 0) return hipSuccess;
    hipLaunchKernelGGL(fetch_vectors_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, (const float4*)src, (float4*)dst, n4);
    return hipGetLastError();
}

hipError_t launch_boundary_pack(const float* const* last, const float* mask, float* send, int n, hipStream_t stream) {
    (void)hipGetLastError();   // the status returned below must be this launch's, not a stale one of the thread (e.g. PyTorch's pointer queries)
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(boundary_pack_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, last, mask, send, n);
    return hipGetLastError();
}

hipError_t launch_boundary_apply(float* const* first, const float* recv, const float* weights, int n, int world,
                                 hipStream_t stream) {
    (void)hipGetLastError();   // the status returned below must be this launch's, not a stale one of the thread (e.g. PyTorch's pointer queries)
    if (n <= 0 || world <= 0) return hipSuccess;
    hipLaunchKernelGGL(boundary_apply_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, first, recv, weights, n, world);
    return hipGetLastError();
}

}  // namespace hispmv

#ifdef HISPMV_WG_TRACE
// Diagnostic builds only (see WgTrace above): hand the kernels a device buffer of 4 + 4*cap u64 (zeroed by the caller), or NULL.
extern "C" __attribute__((visibility("default"))) int hispmv_wg_trace_set(void* buf, unsigned cap) {
    unsigned long long* p = (unsigned long long*)buf;
    if (hipMemcpyToSymbol(HIP_SYMBOL(hispmv::g_wgt_buf), &p, sizeof(p)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(hispmv::g_wgt_cap), &cap, sizeof(cap)) != hipSuccess) return -1;
    return hipDeviceSynchronize() == hipSuccess ? 0 : -1;
}
#endif
