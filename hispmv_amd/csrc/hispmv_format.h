// hispmv_format.h -- constants of the slice-stream format shared by the host packer
// (hispmv_prep.cpp) and the device kernels (hispmv_kernels.hip).
#pragma once
#include <cstdint>

namespace hispmv {

constexpr int kSliceElems = 1024;             // elements per wavefront slice (8 KiB of stream)
constexpr int kStepElems = 128;               // elements one wave-wide 16-byte load covers (2 per lane)
constexpr int kSliceSteps = kSliceElems / kStepElems;
constexpr uint32_t kRowEndBit = 0x80000000u;  // meta bit 31; bits 30:0 = column
constexpr uint32_t kGlobalColBit = 0x40000000u;  // device stream of an LDS-staged group: bits 29:0 are a column of x
                                                 // (gathered through L2), not an index into the group's window

}  // namespace hispmv
