// hispmv_prep_device.hip -- the preprocessor's two heavy stages on the MI355X:
//   COO -> CSR   the reference does this serially with a vector per (tile, row), ~87 % of its 18 s for soc-Pokec
//                (common/src/spmv-helper.cpp:139-227); here: one stable LSD radix sort of (row << 32 | col) keys with
//                the values as payload (rocPRIM), row pointers by binary search on the sorted keys
//   CSR -> slice stream (hispmv_prep.cpp: build_stream): one thread per slice for the headers, one thread per
//                stream position for the words
// The row offsets of the stream (fillers, row-aligned slices) are a sequential O(rows) pass and stay on the host
// (stream_row_offsets), between the two device stages.  The result is the SAME Csr / SliceStream the host path builds
// -- tests compare them byte for byte (tests/test_gpu_prep.py) -- and is planned and uploaded like it.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <chrono>
#include <string>

#include "hispmv_prep.h"
#include "hispmv_prep_device.h"

namespace hispmv {

namespace {

#define PD_TRY(call)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); return false; } \
    } while (0)

struct DevBuf {            // frees on scope exit
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    template <class T> T* as() const { return (T*)p; }
};

__global__ void make_keys_kernel(const int32_t* __restrict__ r, const int32_t* __restrict__ c, int64_t n, int32_t rows, int32_t cols,
                                 unsigned long long* __restrict__ keys, int* __restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t ri = r[i], ci = c[i];
    if (ri < 0 || ri >= rows || ci < 0 || ci >= cols) { *bad = 1; keys[i] = ~0ull; return; }
    keys[i] = ((unsigned long long)(unsigned)ri << 32) | (unsigned)ci;
}

// row_ptr[i] = first sorted position whose row is >= i (i = 0..rows)
__global__ void row_ptr_kernel(const unsigned long long* __restrict__ keys, int64_t n, int32_t rows, long long* __restrict__ row_ptr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > rows) return;
    const unsigned long long want = (unsigned long long)i << 32;
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (keys[mid] < want) lo = mid + 1; else hi = mid; }
    row_ptr[i] = lo;
}

__global__ void split_keys_kernel(const unsigned long long* __restrict__ keys, int64_t n, int32_t* __restrict__ col) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) col[i] = (int32_t)(unsigned)keys[i];
}

__device__ __forceinline__ int64_t upper_bound_rows(const long long* __restrict__ eoff1, int32_t R, long long v) {
    // number of entries of eoff[1..R] that are <= v  ==  std::upper_bound(eoff + 1, eoff + R + 1, v) - (eoff + 1)
    int64_t lo = 0, hi = R;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (eoff1[mid] <= v) lo = mid + 1; else hi = mid; }
    return lo;
}

// One thread per slice: {row_base, chain_len, x_base, x_span} exactly as build_stream computes them.
__global__ void slice_hdr_kernel(const long long* __restrict__ row_ptr, const int32_t* __restrict__ col, const long long* __restrict__ eoff,
                                 int32_t R, long long n_elems, long long n_slices, int4* __restrict__ hdr) {
    const long long sl = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (sl >= n_slices) return;
    const long long S = kSliceElems, b = sl * S, e = (b + S < n_elems) ? b + S : n_elems;
    const int64_t rb = upper_bound_rows(eoff + 1, R, b), rb_next = upper_bound_rows(eoff + 1, R, b + S);
    int4 h = int4{(int)rb, 0, 0, 1};
    if (rb < R && eoff[rb] < b && rb_next > rb) h.y = (int)(sl - eoff[rb] / S);
    int cmin = INT32_MAX, cmax = -1;
    for (int64_t r = rb; r < R && eoff[r] < e; ++r) {
        const long long len = row_ptr[r + 1] - row_ptr[r];
        if (len == 0) continue;
        const long long d0 = b - eoff[r], d1 = e - eoff[r];
        const long long k0 = row_ptr[r] + (d0 > 0 ? d0 : 0);
        const long long k1 = row_ptr[r] + (d1 < len ? d1 : len);
        if (k0 < k1) { cmin = min(cmin, col[k0]); cmax = max(cmax, col[k1 - 1]); }
    }
    if (cmax < 0) { cmin = 0; cmax = 0; }
    h.z = cmin; h.w = cmax - cmin + 1;
    hdr[sl] = h;
}

// One thread per stream position: the element word (fp32 bits | (rowEnd << 31 | column) << 32).
__global__ void slice_words_kernel(const long long* __restrict__ row_ptr, const int32_t* __restrict__ col, const float* __restrict__ val,
                                   const long long* __restrict__ eoff, const int4* __restrict__ hdr, int32_t R, long long n_elems,
                                   long long n_words, unsigned long long* __restrict__ words) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_words) return;
    const int cmin = hdr[k / kSliceElems].z;                 // fillers and padding take the slice's window base
    unsigned vb = 0, meta;
    if (k >= n_elems) {
        meta = (unsigned)cmin;                               // tail padding: inert, never a row end
    } else {
        const int64_t r = upper_bound_rows(eoff + 1, R, k);  // the row position k belongs to: eoff[r] <= k < eoff[r+1]
        const long long s = row_ptr[r], e = row_ptr[r + 1], j = k - eoff[r];
        const bool last = k + 1 == eoff[r + 1];
        if (j < e - s) { vb = __float_as_uint(val[s + j]); meta = (unsigned)col[s + j]; }
        else if (e > s) meta = (unsigned)col[e - 1];         // extension of a row up to the slice boundary: its last column
        else meta = (unsigned)cmin;                          // filler of an empty row
        if (last) meta |= kRowEndBit;
    }
    words[k] = ((unsigned long long)meta << 32) | vb;
}

double secs_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace

bool prep_on_device(int32_t rows, int32_t cols, int64_t nnz, const int32_t* r, const int32_t* c, const float* v,
                    Csr& csr, SliceStream& st, DevicePrepTimes& times, std::string& err) {
    if (rows <= 0 || cols <= 0 || nnz < 0) { err = "bad sparse matrix arguments"; return false; }
    if (rows >= (1 << 30) || cols >= (1 << 30)) { err = "dimension >= 2^30 is not supported"; return false; }
    if (nnz >= (1LL << 31)) { err = "more than 2^31 entries"; return false; }
    hipStream_t stream = nullptr;     // the default stream of the current device
    const size_t n = (size_t)nnz, n1 = std::max<size_t>(n, 1);
    auto t0 = std::chrono::steady_clock::now();
    DevBuf d_r, d_c, d_v, d_keys, d_keys2, d_v2, d_tmp, d_bad, d_rp, d_col;
    PD_TRY(hipMalloc(&d_r.p, n1 * 4)); PD_TRY(hipMalloc(&d_c.p, n1 * 4)); PD_TRY(hipMalloc(&d_v.p, n1 * 4));
    PD_TRY(hipMalloc(&d_keys.p, n1 * 8)); PD_TRY(hipMalloc(&d_keys2.p, n1 * 8)); PD_TRY(hipMalloc(&d_v2.p, n1 * 4));
    PD_TRY(hipMalloc(&d_bad.p, 4)); PD_TRY(hipMalloc(&d_rp.p, ((size_t)rows + 1) * 8)); PD_TRY(hipMalloc(&d_col.p, n1 * 4));
    PD_TRY(hipMemsetAsync(d_bad.p, 0, 4, stream));
    if (n) {
        PD_TRY(hipMemcpyAsync(d_r.p, r, n * 4, hipMemcpyHostToDevice, stream));
        PD_TRY(hipMemcpyAsync(d_c.p, c, n * 4, hipMemcpyHostToDevice, stream));
        PD_TRY(hipMemcpyAsync(d_v.p, v, n * 4, hipMemcpyHostToDevice, stream));
    }
    PD_TRY(hipStreamSynchronize(stream));
    times.upload = secs_since(t0);

    // ---- COO -> CSR ------------------------------------------------------------------------------------------------
    t0 = std::chrono::steady_clock::now();
    const unsigned blocks_n = (unsigned)((n + 255) / 256);
    if (n) hipLaunchKernelGGL(make_keys_kernel, dim3(blocks_n), dim3(256), 0, stream, d_r.as<int32_t>(), d_c.as<int32_t>(), (int64_t)nnz, rows, cols,
                              d_keys.as<unsigned long long>(), d_bad.as<int>());
    int bad = 0;
    PD_TRY(hipMemcpyAsync(&bad, d_bad.p, 4, hipMemcpyDeviceToHost, stream));
    PD_TRY(hipStreamSynchronize(stream));
    if (bad) { err = "COO index outside matrix dimensions"; return false; }
    if (n) {
        // stable sort by (row, column): duplicates keep their input order, like the host's counting sort + stable row sort
        int row_bits = 1;
        while ((1LL << row_bits) < rows) ++row_bits;
        size_t tmp_bytes = 0;
        PD_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys.as<unsigned long long>(), d_keys2.as<unsigned long long>(), d_v.as<float>(),
                                         d_v2.as<float>(), n, 0u, (unsigned)(32 + row_bits), stream));
        PD_TRY(hipMalloc(&d_tmp.p, std::max<size_t>(tmp_bytes, 16)));
        PD_TRY(rocprim::radix_sort_pairs(d_tmp.p, tmp_bytes, d_keys.as<unsigned long long>(), d_keys2.as<unsigned long long>(), d_v.as<float>(),
                                         d_v2.as<float>(), n, 0u, (unsigned)(32 + row_bits), stream));
        hipLaunchKernelGGL(split_keys_kernel, dim3(blocks_n), dim3(256), 0, stream, d_keys2.as<unsigned long long>(), (int64_t)nnz, d_col.as<int32_t>());
    }
    hipLaunchKernelGGL(row_ptr_kernel, dim3((unsigned)(((size_t)rows + 1 + 255) / 256)), dim3(256), 0, stream, d_keys2.as<unsigned long long>(),
                       (int64_t)nnz, rows, d_rp.as<long long>());
    PD_TRY(hipGetLastError());
    csr = Csr{};
    csr.rows = rows; csr.cols = cols;
    csr.row_ptr.resize((size_t)rows + 1); csr.col.resize(n); csr.val.resize(n);
    prefault_parallel(csr.col.data(), (size_t)n * 4); prefault_parallel(csr.val.data(), (size_t)n * 4);
    PD_TRY(hipStreamSynchronize(stream));
    times.csr_device = secs_since(t0);
    t0 = std::chrono::steady_clock::now();
    PD_TRY(hipMemcpyAsync(csr.row_ptr.data(), d_rp.p, ((size_t)rows + 1) * 8, hipMemcpyDeviceToHost, stream));
    if (n) {
        PD_TRY(hipMemcpyAsync(csr.col.data(), d_col.p, n * 4, hipMemcpyDeviceToHost, stream));
        PD_TRY(hipMemcpyAsync(csr.val.data(), d_v2.p, n * 4, hipMemcpyDeviceToHost, stream));
    }
    PD_TRY(hipStreamSynchronize(stream));
    times.download = secs_since(t0);

    // ---- CSR -> slice stream ---------------------------------------------------------------------------------------
    t0 = std::chrono::steady_clock::now();
    const std::vector<int64_t> eoff = stream_row_offsets(rows, csr.row_ptr.data());      // sequential O(rows): host
    times.offsets_host = secs_since(t0);
    t0 = std::chrono::steady_clock::now();
    st = SliceStream{};
    st.rows = rows; st.cols = cols; st.nnz = nnz;
    st.n_elems = eoff[(size_t)rows];
    st.n_slices = (st.n_elems + kSliceElems - 1) / kSliceElems;
    const long long n_words = st.n_slices * kSliceElems;
    DevBuf d_eoff, d_hdr, d_words;
    PD_TRY(hipMalloc(&d_eoff.p, ((size_t)rows + 1) * 8));
    PD_TRY(hipMalloc(&d_hdr.p, std::max<size_t>((size_t)st.n_slices, 1) * 16));
    PD_TRY(hipMalloc(&d_words.p, std::max<size_t>((size_t)n_words, 1) * 8));
    PD_TRY(hipMemcpyAsync(d_eoff.p, eoff.data(), ((size_t)rows + 1) * 8, hipMemcpyHostToDevice, stream));
    if (st.n_slices > 0) {
        hipLaunchKernelGGL(slice_hdr_kernel, dim3((unsigned)((st.n_slices + 63) / 64)), dim3(64), 0, stream, d_rp.as<long long>(), d_col.as<int32_t>(),
                           d_eoff.as<long long>(), rows, (long long)st.n_elems, (long long)st.n_slices, d_hdr.as<int4>());
        hipLaunchKernelGGL(slice_words_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, stream, d_rp.as<long long>(), d_col.as<int32_t>(),
                           d_v2.as<float>(), d_eoff.as<long long>(), d_hdr.as<int4>(), rows, (long long)st.n_elems, n_words,
                           d_words.as<unsigned long long>());
    }
    PD_TRY(hipGetLastError());
    PD_TRY(hipStreamSynchronize(stream));
    times.stream_device = secs_since(t0);
    t0 = std::chrono::steady_clock::now();
    st.words.resize((size_t)n_words);
    prefault_parallel(st.words.data(), (size_t)n_words * 8);
    st.hdr.resize((size_t)st.n_slices);
    if (st.n_slices > 0) {
        PD_TRY(hipMemcpyAsync(st.words.data(), d_words.p, (size_t)n_words * 8, hipMemcpyDeviceToHost, stream));
        PD_TRY(hipMemcpyAsync(st.hdr.data(), d_hdr.p, (size_t)st.n_slices * 16, hipMemcpyDeviceToHost, stream));
    }
    PD_TRY(hipStreamSynchronize(stream));
    for (int64_t sl = 0; sl < st.n_slices; ++sl)
        if (st.hdr[(size_t)sl].chain_len > 0) {
            const SliceHdr& h = st.hdr[(size_t)sl];
            st.fix.push_back(FixEntry{h.row_base, (int32_t)(sl - h.chain_len), h.chain_len, 0});
        }
    times.download += secs_since(t0);
    return true;
}

// ---- device layout of a planned slice stream (pack_device_stream, hispmv_plan.cpp) ------------------------------------------
// One 256-thread workgroup per slice, 4 consecutive elements per thread.
__global__ __launch_bounds__(256) void layout_slices_kernel(const uint64_t* __restrict__ words, long long n_slices, int G, const int4* __restrict__ groups,
                                                            int window_floats, int n_waves, uint8_t* __restrict__ bytes, uint32_t* __restrict__ stray_cols) {
    __shared__ int wave_strays[4];
    const long long sl = blockIdx.x;
    const long long g = sl / G, s0 = g * G;
    const long long s1 = s0 + G < n_slices ? s0 + G : n_slices;
    const int n_here = (int)(s1 - s0);
    const int4 gd = groups[g];                       // {frag_begin, frag_count, offset in kSliceUnit, 1 compact | 2 stray slots}
    const bool compact = gd.w != 0;
    uint8_t* const base = bytes + (size_t)(unsigned)gd.z * kSliceUnit + (size_t)(sl - s0) * (compact ? kCompactSliceBytes : kWideSliceBytes);
    const int t = (int)threadIdx.x, i0 = 4 * t;
    const uint64_t* w = words + sl * kSliceElems + i0;
    uint64_t e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) e[k] = w[k];
    uint4 v;
    v.x = (uint32_t)e[0]; v.y = (uint32_t)e[1]; v.z = (uint32_t)e[2]; v.w = (uint32_t)e[3];
    ((uint4*)base)[t] = v;
    uint32_t m[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) m[k] = (uint32_t)(e[k] >> 32);
    if (!compact) {
        uint4 q; q.x = m[0]; q.y = m[1]; q.z = m[2]; q.w = m[3];
        ((uint4*)(base + kSliceElems * 4))[t] = q;
        return;
    }
    // strays of the slice in element order: count per thread, exclusive scan over the workgroup
    int mine = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) mine += (m[k] & kGlobalColBit) ? 1 : 0;
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if ((t & 63) >= d) incl += o; }
    if ((t & 63) == 63) wave_strays[t >> 6] = incl;
    __syncthreads();
    int before = incl - mine;
    for (int q = 0; q < (t >> 6); ++q) before += wave_strays[q];
    const int rot = n_here > 0 ? (int)(((unsigned long long)g * 29ull) % (unsigned long long)n_here) : 0;      // the kernel's walk (slices_group)
    const int pos = (int)(((sl - s0) - rot + n_here) % n_here);
    const uint32_t area = (uint32_t)window_floats + (uint32_t)(pos % n_waves) * kStraySlots;
    uint16_t out[4];
    int k_stray = before;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t idx = m[k] & 0x7fffu;
        if (m[k] & kGlobalColBit) {
            if (stray_cols && k_stray < kStraySlots) stray_cols[(size_t)sl * kStraySlots + k_stray] = m[k] & ~(kRowEndBit | kGlobalColBit);
            idx = area + (uint32_t)k_stray++;
        }
        out[k] = (uint16_t)(idx | ((m[k] & kRowEndBit) ? kCompactEndBit : 0u));
    }
    uint2 q;
    q.x = (uint32_t)out[0] | ((uint32_t)out[1] << 16); q.y = (uint32_t)out[2] | ((uint32_t)out[3] << 16);
    ((uint2*)(base + kSliceElems * 4))[t] = q;
}

int layout_on_device(const uint64_t* d_words, int64_t n_slices, int group_slices, const int32_t* d_groups, int window_floats, int n_waves,
                     uint8_t* d_bytes, uint32_t* d_stray_cols, void* stream) {
    if (n_slices <= 0) return 0;
    hipLaunchKernelGGL(layout_slices_kernel, dim3((unsigned)n_slices), dim3(256), 0, (hipStream_t)stream, d_words, (long long)n_slices, group_slices,
                       (const int4*)d_groups, window_floats, n_waves, d_bytes, d_stray_cols);
    return (int)hipGetLastError();
}

}  // namespace hispmv
