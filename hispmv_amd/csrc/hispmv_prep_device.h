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

}  // namespace hispmv
