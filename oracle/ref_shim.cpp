// ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.  C entry points around the REFERENCE's own
// MatrixMarket -> CSC -> CSR loader (cpu/src/helper_functions.cpp:148-241), which the recipe
// in oracle/Makefile compiles from where it lies under /root/reference into
// oracle/_ref/libref_cpu.so.  No reference source is copied: this file only includes the
// reference's header and calls its functions.  Used to pin orc_read_mtx_cpu (the restatement)
// and the product's CSR indices bit-exactly, and by tests/golden/make_golden.py.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "helper_functions.h"   // found through -I$(REF)/cpu/src

template <class T>
static T* dup_vec(const std::vector<T>& v) {
    T* p = (T*)std::malloc((v.empty() ? 1 : v.size()) * sizeof(T));
    if (!v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

extern "C" __attribute__((visibility("default")))
int ref_read_mtx_csr(const char* path, int* rows, int* cols, int* nnz, int** row_ptr, int** col_idx, float** vals) {
    std::vector<float> cscValues, csrValues;
    std::vector<int> cscRowIndices, cscColOffsets, csrCol, csrRow;
    readMatrixCSC((char*)path, cscValues, cscRowIndices, cscColOffsets, *rows, *cols, *nnz);
    convertCSCtoCSR(cscValues, cscRowIndices, cscColOffsets, csrValues, csrCol, csrRow, *rows, *cols, *nnz);
    *row_ptr = dup_vec(csrRow); *col_idx = dup_vec(csrCol); *vals = dup_vec(csrValues);
    return 0;
}
extern "C" __attribute__((visibility("default"))) void ref_free(void* p) { std::free(p); }
