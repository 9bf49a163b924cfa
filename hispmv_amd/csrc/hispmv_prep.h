// hispmv_prep.h -- host preprocessor: MatrixMarket / COO -> CSR -> equal-nnz wavefront
// slice stream ("HSS").  MI355X counterpart of the reference's HiSpmvHandle
// preprocessing (common/src/spmv-helper.cpp:139-715), redesigned rather than ported:
//
//   reference (FPGA)                               here (gfx950)
//   ---------------------------------------------  -------------------------------------------
//   64-bit self-describing word per PE slot         same idea: 64-bit word per element,
//   {valid,row15,tileEnd,shared,col14,fp32}          {rowEnd:1, col:31 | fp32 value}
//   (spmv-helper.h:45-60)
//   rows owned cyclically by 8*num_ch_A PEs,        elements kept in CSR order and cut into
//   balanced by padding + "shared rows" split        slices of exactly SLICE_ELEMS; a row that
//   over all PEs (balanceWorkload :265-347)          crosses a cut is "shared" between the
//                                                    wavefronts on both sides (carry chain)
//   2-D tiling so x fits BRAM / y fits URAM          no tiling: x is gathered through L2/LDS,
//   (tileAndPad :242-263)                            y rows are written once by their owner
//
// Pure host code (no HIP); compiled with g++ -fopenmp and linked into libhispmv.so.
#pragma once
#include <cstdint>
#include <string>
#include "hispmv_format.h"
#include <vector>

namespace hispmv {

struct SliceHdr {          // 16 B per slice
    int32_t row_base;      // number of row ends before this slice = first row that ends here
    int32_t chain_len;     // >0: first row ending here began chain_len slices earlier
    int32_t x_base;        // min column referenced in the slice
    int32_t x_span;        // max column - min column + 1
};

struct FixEntry {          // one per row whose elements span more than one slice
    int32_t row;           // the row (owned by the slice holding its last element)
    int32_t first_slice;   // first slice holding part of the row
    int32_t len;           // number of preceding slices whose carry belongs to the row
    int32_t pad;
};

struct Coo {
    int32_t rows = 0, cols = 0;
    std::vector<int32_t> r, c;
    std::vector<float> v;
};

// The big arrays (one entry per nonzero) use the default-initialising allocator: they are written exactly once, by a
// device-to-host copy or a parallel loop -- value-initialising them first walked 490 MB of fresh pages on ONE core before
// every download of soc-Pokec's shape.
using IndexVec = std::vector<int32_t, DefaultInitAllocator<int32_t>>;
using ValueVec = std::vector<float, DefaultInitAllocator<float>>;
using WordVec = std::vector<uint64_t, DefaultInitAllocator<uint64_t>>;

struct Csr {
    int32_t rows = 0, cols = 0;
    std::vector<int64_t> row_ptr;   // rows + 1
    IndexVec col;
    ValueVec val;
    int64_t nnz() const { return row_ptr.empty() ? 0 : row_ptr.back(); }
};

struct SliceStream {
    int32_t rows = 0, cols = 0;
    int64_t nnz = 0;          // real nonzeros
    int64_t n_elems = 0;      // nnz + one filler per empty row (before tail padding)
    int64_t n_slices = 0;
    WordVec words;                      // n_slices * kSliceElems
    std::vector<SliceHdr> hdr;          // n_slices
    std::vector<FixEntry> fix;          // rows split across slices, ascending row
    int64_t bytes() const { return (int64_t)words.size() * 8 + (int64_t)hdr.size() * 16 + (int64_t)fix.size() * 16; }
};

enum MtxFlavor {
    kFlavorCommon = 0,   // HiSpmvHandle::loadMtx semantics (spmv-helper.cpp:34-136)
    kFlavorCpu = 1       // cpu/ driver semantics (helper_functions.cpp:91-146)
};

// OpenMP threads of the host preprocessor = the CPUs the process may use (cgroup quota, affinity mask; HISPMV_HOST_THREADS /
// OMP_NUM_THREADS override), decided once.  Every parallel region of the library carries num_threads(host_threads()); the
// process-global OpenMP setting is never changed.
int host_threads();
// Touches every page of [p, p + bytes) from all OpenMP threads (before a buffer is filled by a single-threaded copy).
void prefault_parallel(void* p, size_t bytes);

// MatrixMarket coordinate reader.  Throws std::runtime_error with the reference's
// failure classes (not a MatrixMarket file / unsupported type) -- spmv-helper.cpp:50-71.
Coo read_mtx(const std::string& path, MtxFlavor flavor);

// COO -> CSR: stable counting sort by row, then stable sort by column inside each row;
// duplicates are kept as separate entries (reference: spmv-helper.cpp:139-227 keeps
// them too).  Throws std::out_of_range on an index outside [0,rows) x [0,cols).
Csr coo_to_csr(int32_t rows, int32_t cols, int64_t nnz, const int32_t* r, const int32_t* c, const float* v);

// Stable per-row sort by column of a CSR whose rows may be unsorted (rows already ascending are left alone).
void sort_rows_by_column(Csr& m);

// Element offset of every row in the slice stream (fillers for empty rows, row-aligned slices): rows + 1 entries.
std::vector<int64_t> stream_row_offsets(int32_t rows, const int64_t* row_ptr);

// CSR -> slice stream.
SliceStream build_stream(const Csr& m);

// Pack one element (shared by the packer and the decoder in tests).
inline uint64_t pack_elem(float val, int32_t col, bool row_end) {
    uint32_t vb; __builtin_memcpy(&vb, &val, 4);
    uint32_t meta = (uint32_t)col | (row_end ? kRowEndBit : 0u);
    return ((uint64_t)meta << 32) | vb;
}

}  // namespace hispmv
