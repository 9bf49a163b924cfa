// hispmv_prep_device.h -- COO -> CSR -> slice stream on the MI355X (hispmv_prep_device.hip); same results as
// coo_to_csr + build_stream of hispmv_prep.h, byte for byte.
#pragma once
#include <string>

#include "hispmv_prep.h"

namespace hispmv {

struct DevicePrepTimes { double upload = 0, csr_device = 0, offsets_host = 0, stream_device = 0, download = 0; };

// Runs on the current HIP device (default stream).  false + `err` on a HIP failure or an index outside the matrix.
bool prep_on_device(int32_t rows, int32_t cols, int64_t nnz, const int32_t* r, const int32_t* c, const float* v,
                    Csr& csr, SliceStream& st, DevicePrepTimes& times, std::string& err);

// The device layout of a planned slice stream (pack_device_stream of hispmv_plan.h, byte for byte) written on the device from the
// host words (8 B per element, metas already rewritten to window indices by make_plan): compact groups to 6 B per element with
// their strays numbered per slice and the strays' columns written behind the headers, wide groups copied.  `d_groups` is the
// device copy of DeviceStream::groups; `d_stray_cols` (n_slices x kStraySlots, pre-filled with 0xffffffff) may be null when
// no group has stray slots.  Asynchronous on `stream`; returns the launch error.
int layout_on_device(const uint64_t* d_words, int64_t n_slices, int group_slices, const int32_t* d_groups, int window_floats, int n_waves,
                     uint8_t* d_bytes, uint32_t* d_stray_cols, void* stream);

}  // namespace hispmv
