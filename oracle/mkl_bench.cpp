// mkl_bench.cpp -- TEST INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg).
//
// The reference's timed CPU path -- mkl_sparse_s_mv (cpu/src/main.cpp:26-49) and cblas_sgemv (:74-96) -- called the
// way the reference calls it, from a process that holds NO other OpenMP runtime: this library is deliberately built
// without -fopenmp (liboracle.so links libgomp, whose start-up binds the initial thread to the first place when
// OMP_PROC_BIND is set; MKL's own runtime, libiomp5, then inherits a one-core mask and all its threads share that
// core).  With only libiomp5 in the process, OMP_PLACES=cores OMP_PROC_BIND=close act as in cpu/env.sh:2-4.
// MKL is a closed third-party library: bound through dlopen (libmkl_rt.so), three sparse entry points + cblas_sgemv.
// The matrix is copied into buffers first-touched by `threads` std::threads (row blocks of equal nnz).
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <thread>
#include <vector>

#define MB_API extern "C" __attribute__((visibility("default")))

struct mkl_descr { int type, mode, diag; };
typedef int (*mkl_create_csr_t)(void**, int, int, int, int*, int*, int*, float*);
typedef int (*mkl_mv_t)(int, float, void*, mkl_descr, const float*, float, float*);
typedef int (*mkl_destroy_t)(void*);
typedef void (*mkl_set_threads_t)(int);
typedef int (*mkl_get_threads_t)(void);
typedef void (*cblas_sgemv_t)(int, int, int, int, float, const float*, int, const float*, int, float, float*, int);

static void* open_mkl() {
    static void* h = nullptr; static bool tried = false;
    if (tried) return h;
    tried = true;
    const char* names[] = {"libmkl_rt.so", "libmkl_rt.so.2", "libmkl_rt.so.1",
                           "/opt/conda/lib/libmkl_rt.so", "/opt/conda/lib/libmkl_rt.so.2", "/opt/conda/lib/libmkl_rt.so.1"};
    for (auto n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    return h;
}
MB_API int mb_mkl_available() { return open_mkl() != nullptr; }

template <class F> static void par(int threads, F&& f) {
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back([&, t]() { f(t); });
    for (auto& q : th) q.join();
}

MB_API double mb_spmv(int rows, int cols, const int* rp, const int* ci, const float* va, int threads, double budget_s,
                      int max_reps, int* reps_out, int* threads_out) {
    void* h = open_mkl();
    if (!h) return -1.0;
    auto create = (mkl_create_csr_t)dlsym(h, "mkl_sparse_s_create_csr");
    auto mv = (mkl_mv_t)dlsym(h, "mkl_sparse_s_mv");
    auto destroy = (mkl_destroy_t)dlsym(h, "mkl_sparse_destroy");
    auto setthr = (mkl_set_threads_t)dlsym(h, "MKL_Set_Num_Threads");
    auto getthr = (mkl_get_threads_t)dlsym(h, "MKL_Get_Max_Threads");
    if (!create || !mv || !destroy) return -2.0;
    if (threads < 1) threads = 1;
    if (setthr) setthr(threads);                  // cpu/src/main.cpp:136 (mkl_set_num_threads(24) there)
    if (threads_out) *threads_out = getthr ? getthr() : threads;
    const int64_t nnz = rp[rows];
    int* rp2 = (int*)std::malloc(((size_t)rows + 1) * sizeof(int));
    int* ci2 = (int*)std::malloc(std::max<size_t>(1, (size_t)nnz) * sizeof(int));
    float* va2 = (float*)std::malloc(std::max<size_t>(1, (size_t)nnz) * sizeof(float));
    float* x = (float*)std::malloc((size_t)cols * sizeof(float));
    float* y = (float*)std::malloc((size_t)rows * sizeof(float));
    std::vector<int> cut((size_t)threads + 1, rows);
    cut[0] = 0;
    for (int t = 1; t < threads; ++t) cut[t] = (int)(std::lower_bound(rp, rp + rows + 1, (int)(nnz * t / threads)) - rp);
    for (int t = 1; t <= threads; ++t) cut[t] = std::max(cut[t], cut[t - 1]);
    cut[threads] = rows;
    par(threads, [&](int t) {
        for (int i = cut[t]; i < cut[t + 1]; ++i) {
            rp2[i] = rp[i];
            for (int k = rp[i]; k < rp[i + 1]; ++k) { ci2[k] = ci[k]; va2[k] = va[k]; }
            y[i] = -2.0f * (i + 1) / float(i + 2);                                         // main.cpp:175-178
        }
        const int64_t c0 = (int64_t)cols * t / threads, c1 = (int64_t)cols * (t + 1) / threads;
        for (int64_t j = c0; j < c1; ++j) x[j] = float(j + 1) / float(j + 2);             // main.cpp:173
    });
    rp2[rows] = rp[rows];
    void* A = nullptr;
    double per_rep = -3.0;
    int reps = 0;
    if (create(&A, 0 /*SPARSE_INDEX_BASE_ZERO*/, rows, cols, rp2, rp2 + 1, ci2, va2) == 0) {   // main.cpp:30-32
        mkl_descr d{20 /*SPARSE_MATRIX_TYPE_GENERAL*/, 0, 0};
        const float alpha = 0.85f;                                                             // main.cpp:147
        mv(10 /*NON_TRANSPOSE*/, alpha, A, d, x, 0.0f, y);                                     // warm-up; beta = 0: see header of section 5 in hispmv_oracle.cpp
        auto t0 = std::chrono::steady_clock::now();
        double el = 0;
        while ((reps < 2 || el < budget_s) && reps < max_reps) {
            mv(10, alpha, A, d, x, 0.0f, y);                                                   // main.cpp:38-41
            ++reps;
            el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        per_rep = el / reps;
        destroy(A);
    }
    if (reps_out) *reps_out = reps;
    std::free(rp2); std::free(ci2); std::free(va2); std::free(x); std::free(y);
    return per_rep;
}

MB_API double mb_gemv(int rows, int cols, int threads, double budget_s, int max_reps, int* reps_out) {
    void* h = open_mkl();
    if (!h) return -1.0;
    auto sgemv = (cblas_sgemv_t)dlsym(h, "cblas_sgemv");
    auto setthr = (mkl_set_threads_t)dlsym(h, "MKL_Set_Num_Threads");
    if (!sgemv) return -2.0;
    if (threads < 1) threads = 1;
    if (setthr) setthr(threads);
    float* A = (float*)std::malloc((size_t)rows * cols * sizeof(float));
    float* x = (float*)std::malloc((size_t)cols * sizeof(float));
    float* y = (float*)std::malloc((size_t)rows * sizeof(float));
    par(threads, [&](int t) {
        const int r0 = (int)((int64_t)rows * t / threads), r1 = (int)((int64_t)rows * (t + 1) / threads);
        for (int i = r0; i < r1; ++i) {
            float* Ai = A + (size_t)i * cols;
            for (int j = 0; j < cols; ++j) Ai[j] = float(i + 1) / float(j + 2);            // main.cpp:207-213
            y[i] = -2.0f * (i + 1) / float(i + 2);
        }
    });
    for (int j = 0; j < cols; ++j) x[j] = float(j + 1) / float(j + 2);
    int reps = 0;
    sgemv(101 /*CblasRowMajor*/, 111 /*CblasNoTrans*/, rows, cols, 0.85f, A, cols, x, 1, 0.0f, y, 1);   // main.cpp:85
    auto t0 = std::chrono::steady_clock::now();
    double el = 0;
    while ((reps < 2 || el < budget_s) && reps < max_reps) {
        sgemv(101, 111, rows, cols, 0.85f, A, cols, x, 1, 0.0f, y, 1);
        ++reps;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    if (reps_out) *reps_out = reps;
    std::free(A); std::free(x); std::free(y);
    return el / reps;
}
