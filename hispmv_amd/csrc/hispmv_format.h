// hispmv_format.h -- constants of the slice-stream format shared by the host packer
// (hispmv_prep.cpp), the device packer (hispmv_plan.cpp) and the device kernels (hispmv_kernels.hip).
#pragma once
#include <cstdint>
#include <memory>
#include <utility>

namespace hispmv {

constexpr int kSliceElems = 1024;             // elements per wavefront slice
constexpr uint32_t kRowEndBit = 0x80000000u;  // host word / wide device meta: bit 31; bits 30:0 = column
constexpr uint32_t kGlobalColBit = 0x40000000u;  // wide device meta of an LDS-staged group: bits 29:0 are a column of x
                                                 // (gathered through L2), not an index into the group's window

// Device layout of one slice (structure of arrays, element order = the host stream's CSR order):
//   [values: 1024 x fp32][meta: 1024 x u16 (compact) or 1024 x u32 (wide)]
// compact meta = rowEnd<<15 | index into the workgroup's LDS window of x (< 32768): 6 bytes per element instead of the
// 8 of the reference's word (spmv-helper.h:45-60) -- every slice of a group whose elements all lie inside its window;
// wide meta = rowEnd<<31 | (kGlobalColBit | column, or window index, or plain column for groups without a window).
// A wavefront reads a slice in kSliceSteps steps of kStepElems elements, kLaneElems consecutive elements per lane and step
// (one global_load_dwordx4 of values + one dwordx2 / dwordx4 of meta per lane and step, fully coalesced).
constexpr int kLaneElems = 4;
constexpr int kStepElems = 64 * kLaneElems;               // 256
constexpr int kSliceSteps = kSliceElems / kStepElems;     // 4
constexpr uint32_t kCompactEndBit = 0x8000u;
constexpr int kCompactMaxIndex = 32768;                   // LDS window floats addressable by a compact meta
constexpr int kSliceUnit = 2048;                          // slice sizes and offsets are multiples of this many bytes
constexpr int kCompactSliceBytes = kSliceElems * 6;       // 6144 = 3 units
constexpr int kWideSliceBytes = kSliceElems * 8;          // 8192 = 4 units
constexpr int kStraySlots = 64;                           // stray slots of a compact slice (hispmv_plan.h): x values one wavefront keeps behind the window
constexpr int kFixShortMax = 32;                          // a row cut over at most this many slices is finished by one thread of the tail launch, a longer chain by a wavefront

// Allocator that leaves trivially constructible elements uninitialised: a packed stream of hundreds of MB is
// written exactly once by the packers' parallel loops -- value-initialising it first walked every page on ONE core (1.8 of the
// packer's 2.2 s in the build container, where a page fault is expensive), now the copying threads touch the pages.
template <class T> struct DefaultInitAllocator : std::allocator<T> {
    template <class U> struct rebind { using other = DefaultInitAllocator<U>; };
    template <class U> void construct(U* p) noexcept { ::new ((void*)p) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};


}  // namespace hispmv
