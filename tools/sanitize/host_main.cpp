// Sanitizer driver for the host side (tools/sanitize_host.sh): random matrices of five kinds through
// coo_to_csr -> build_stream -> make_plan -> device layout, the transposed tile stream, unsorted CSR rows, and the
// MatrixMarket reader on the files given as arguments.
#include <cstdio>
#include <random>

#include "hispmv_plan.h"
#include "hispmv_prep.h"
#include "hispmv_tts.h"
using namespace hispmv;

int main(int argc, char** argv) {
    for (int i = 1; i < argc; ++i)
        for (int fl = 0; fl < 2; ++fl) {
            try { Coo c = read_mtx(argv[i], (MtxFlavor)fl); std::printf("%s flavor %d: %d x %d, %zu entries\n", argv[i], fl, c.rows, c.cols, c.r.size()); }
            catch (const std::exception& e) { std::printf("%s flavor %d: %s\n", argv[i], fl, e.what()); }
        }
    std::mt19937 g(5);
    for (int t = 0; t < 60; ++t) {
        const int rows = 1 + g() % 60000, cols = 1 + g() % 90000, kind = t % 5;
        const long nnz = g() % 400000;
        std::vector<int32_t> r(nnz), c(nnz);
        std::vector<float> v(nnz);
        for (long i = 0; i < nnz; ++i) {
            r[i] = g() % rows;
            if (kind == 1 && i < nnz / 3) r[i] = rows / 2;                       // one heavy row
            if (kind == 2) r[i] = (r[i] / 3) * 3 % rows;                         // empty rows
            c[i] = kind == 3 ? (int)(((long)r[i] * cols / rows + g() % 200) % cols) : g() % cols;
            if (kind == 4) c[i] = (int)(((long)r[i] * cols / rows + (g() % 16 ? g() % 100 : g() % cols)) % cols);   // band + strays
            v[i] = (float)(g() % 100) / 50.f - 1.f;
        }
        Csr m = coo_to_csr(rows, cols, nnz, r.data(), c.data(), v.data());
        SliceStream st = build_stream(m);
        for (int cus : {256, 8, 1}) { SliceStream copy = st; LaunchPlan p = make_plan(copy, cus); DeviceStream d = pack_device_stream(copy, p); (void)d; }
        TtsStream ts = build_tts(m, t % 3 == 0 ? 5000 : 0);                      // the second device format
        (void)ts;
        if (t % 7 == 0) {                                                        // CSR handed over with unsorted rows
            Csr u = m;
            for (int32_t i = 0; i + 1 <= u.rows; ++i) {
                const int64_t s0 = u.row_ptr[i], e0 = u.row_ptr[(size_t)i + 1];
                if (e0 - s0 > 1) { std::swap(u.col[(size_t)s0], u.col[(size_t)e0 - 1]); std::swap(u.val[(size_t)s0], u.val[(size_t)e0 - 1]); }
            }
            sort_rows_by_column(u);
            SliceStream su = build_stream(u);
            (void)su;
        }
    }
    {   // a stencil matrix large enough for LDS-window plans
        const int rows = 300000, per = 40;
        std::vector<int32_t> r, c;
        std::vector<float> v;
        for (int i = 0; i < rows; ++i)
            for (int k = 0; k < per; ++k) { r.push_back(i); c.push_back((i + (k * 37) % 3000) % rows); v.push_back(1.f); }
        Csr m = coo_to_csr(rows, rows, (long)r.size(), r.data(), c.data(), v.data());
        SliceStream st = build_stream(m);
        LaunchPlan p = make_plan(st, 256);
        std::printf("stencil: slices %lld, %d threads, %d slices per workgroup, window %d floats\n", (long long)st.n_slices, p.block_threads, p.group_slices, p.lds_floats);
    }
    std::puts("sanitize_host: done");
}
