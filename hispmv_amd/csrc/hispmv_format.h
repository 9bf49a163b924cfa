// hispmv_format.h -- constants of the slice-stream format shared by the host packer
// (hispmv_prep.cpp) and the device kernels (hispmv_kernels.hip).
#pragma once
#include <cstdint>

namespace hispmv {

constexpr int kSliceElems = 1024;             // elements per wavefront slice (8 KiB of stream)
constexpr int kStepElems = 128;               // elements one wave-wide 16-byte load covers (2 per lane)
constexpr int kSliceSteps = kSliceElems / kStepElems;
constexpr uint32_t kRowEndBit = 0x80000000u;  // meta bit 31; bits 30:0 = column

}  // namespace hispmv
