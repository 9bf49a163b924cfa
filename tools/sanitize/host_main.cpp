// Sanitizer driver for the host side (tools/sanitize_host.sh): random matrices of five kinds through
// coo_to_csr -> build_stream -> make_plan -> device layout, the transposed tile stream, the format / tiling choice
// (choose_format with its band-tile, column-tile and tile-stream branches forced by thresholds), unsorted CSR rows, and the
// MatrixMarket reader on the files given as arguments.
#include <cstdio>
#include <random>

#include "hispmv_choose.h"
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
        for (int cus : {256, 8, 1}) {
            SliceStream copy = st; LaunchPlan p = make_plan(copy, cus); DeviceStream d = pack_device_stream(copy, p); (void)d;
            if (unplanned_words(copy, p) != st.words) { std::puts("unplanned_words: not the inverse of make_plan"); return 1; }
        }
        TtsStream ts = build_tts(m, t % 3 == 0 ? 5000 : 0);                      // the second device format
        (void)ts;
        {   // the loader's decision, every branch reachable at this size: tile streams from 1 K entries, tiny column tiles, each geometry
            FormatOptions o;
            o.tts_min_nnz = 1024; o.col_tile_bytes = (t % 2) ? 4096 : (4 << 20); o.format_mode = t % 4 == 3 ? 1 : 2; o.tts_geometry = t % 6; o.decide_only = t % 3 == 0;
            Csr copy = m;
            FormatChoice ch = choose_format(std::move(copy), nullptr, t % 2 ? 256 : 8, o);
            if (ch.parts.empty()) { std::puts("choose_format: no parts"); return 1; }
        }
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
        const WordVec before = st.words;
        LaunchPlan p = make_plan(st, 256);
        if (unplanned_words(st, p) != before) { std::puts("unplanned_words: not the inverse of make_plan (stencil)"); return 1; }
        {   // the loader's decision with the batch layout of short groups (6 M entries: 23 slices per workgroup on 256 CUs)
            std::vector<int32_t> rh, ch2; std::vector<float> vh;
            for (int i = 0; i < rows; ++i) for (int k = 0; k < per / 2; ++k) { rh.push_back(i); ch2.push_back((i + (k * 37) % 3000) % rows); vh.push_back(1.f); }
            Csr mc = coo_to_csr(rows, rows, (long)rh.size(), rh.data(), ch2.data(), vh.data());
            FormatOptions o;
            FormatChoice ch = choose_format(std::move(mc), nullptr, 256, o);
            std::printf("stencil through choose_format: %d threads, %d slices per workgroup, batch layout %s (%d slices per workgroup)\n", ch.parts[0].plan.block_threads,
                        ch.parts[0].plan.group_slices, ch.parts[0].has_batch_layout ? "yes" : "no", ch.parts[0].has_batch_layout ? ch.parts[0].batch_plan.group_slices : 0);
        }
        {   // ... and a wide unstructured band of 5 M entries: the band-tile branch
            const int br = 250000, bper = 20, half = 30000;
            std::vector<int32_t> r2, c2; std::vector<float> v2;
            std::mt19937 g2(9);
            for (int i = 0; i < br; ++i) for (int k = 0; k < bper; ++k) { r2.push_back(i); c2.push_back((int)std::min<long>(br - 1, std::max<long>(0, (long)i - half + (long)(g2() % (2 * half))))); v2.push_back(1.f); }
            Csr mb = coo_to_csr(br, br, (long)r2.size(), r2.data(), c2.data(), v2.data());
            FormatChoice ch = choose_format(std::move(mb), nullptr, 256, FormatOptions());
            std::printf("wide band: format %d, tile kind %d, %zu parts\n", ch.format, ch.tile_kind, ch.parts.size());
        }
        {   // ... and 1.3 M scattered entries: the sampled "no window can pay" path (no slice stream), the tile-stream packer in the
            // standard geometry, gap-coded column parts and the shape knob of the column parts
            const int sr = 400000;
            std::vector<int32_t> r3, c3; std::vector<float> v3;
            std::mt19937 g3(11);
            for (int i = 0; i < sr; ++i) for (int k = 0; k < 3 + (int)(g3() % 2); ++k) { r3.push_back(i); c3.push_back((int)(g3() % 900000)); v3.push_back(0.5f); }
            for (int variant = 0; variant < 3; ++variant) {
                Csr ms = coo_to_csr(sr, 900000, (long)r3.size(), r3.data(), c3.data(), v3.data());
                FormatOptions o;
                if (variant == 1) o.tts_geometry = 5;
                if (variant == 2) { o.tts_geometry = 1; o.tall_rows = 8192; o.tall_slots = 28672; o.tall_tiles = 64; o.tall_zero_fill = 0; o.tall_parts = 4; }
                FormatChoice ch = choose_format(std::move(ms), nullptr, 256, o);
                std::printf("scattered (variant %d): format %d, %zu parts, %.1f lines per gather\n", variant, ch.format, ch.parts.size(), ch.tts_lines_per_gather);
            }
        }
        std::printf("stencil: slices %lld, %d threads, %d slices per workgroup, window %d floats\n", (long long)st.n_slices, p.block_threads, p.group_slices, p.lds_floats);
    }
    {   // the order of the step kernel's queue (order_step_queue): all three modes, empty classes, every item exactly once
        std::vector<double> sl, tl;
        for (int i = 0; i < 700; ++i) sl.push_back(10.0 + (i * 37 % 41));
        for (int i = 0; i < 300; ++i) tl.push_back(i < 120 ? 100.0 + i % 7 : 15.0 + i % 13);
        for (int mode = 0; mode < 3; ++mode) {
            for (int empty = 0; empty < 3; ++empty) {
                const std::vector<double> a = empty == 1 ? std::vector<double>() : sl, b = empty == 2 ? std::vector<double>() : tl;
                const auto order = order_step_queue(a, b, 64, mode);
                std::vector<int> seen_a(a.size(), 0), seen_b(b.size(), 0);
                for (const auto& pr : order) (pr.first ? seen_b : seen_a)[(size_t)pr.second]++;
                bool ok = order.size() == a.size() + b.size();
                for (int q : seen_a) ok = ok && q == 1;
                for (int q : seen_b) ok = ok && q == 1;
                if (!ok) { std::puts("order_step_queue: an item is missing or doubled"); return 1; }
            }
        }
        std::puts("step queue: 3 modes x 3 class mixes, every item once");
    }
    std::puts("sanitize_host: done");
}
