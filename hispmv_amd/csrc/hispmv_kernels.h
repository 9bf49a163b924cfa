// hispmv_kernels.h -- launchers of the gfx950 kernels (hispmv_kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#include <cstdint>

namespace hispmv {

struct SpmvDeviceMatrix {
    const uint64_t* words = nullptr;    // n_slices * kSliceElems packed elements
    const int4* hdr = nullptr;          // n_slices x {row_base, chain_len, x_base, x_span}
    const int4* fix_short = nullptr;    // {row, first_slice, len, 0}, len <= kFixShortMax
    const int4* fix_long = nullptr;     // same, len > kFixShortMax
    float* carry = nullptr;             // n_slices: partial sum each slice hands to the next
    int64_t n_slices = 0;
    int32_t n_fix_short = 0, n_fix_long = 0;
    int32_t rows = 0, cols = 0;
};

constexpr int kFixShortMax = 32;

// y = alpha*A*x + beta*bias.  Two launches on `stream`: the slice kernel, then (if any row is
// shared between slices) the carry fix-up.  Returns the first HIP error.
hipError_t launch_spmv(const SpmvDeviceMatrix& m, const float* x, const float* bias, float* y,
                       float alpha, float beta, hipStream_t stream);

// Dense overlay: y = alpha*W*x + beta*bias, W row-major rows x cols.
hipError_t launch_gemv(const float* W, int32_t rows, int32_t cols, const float* x, const float* bias,
                       float* y, float alpha, float beta, hipStream_t stream);

}  // namespace hispmv
