// hispmv_batch.cpp -- hispmv_spmv_device_batch (include/hispmv.h): n independent SpMVs in as few launches as possible.  Builds
// the launch tables of a call signature once (slice grids per workgroup size, tile-stream grids per geometry, one dense grid,
// ONE tail launch), spreads the main launches over lanes (the caller's stream + side streams) and replays two-stream calls as HIP
// graphs whose alpha is patched in place.  No reference counterpart: the reference runs one matrix at a time
// (pyhispmv/src/fpga_handle.cpp:286-321).  Host-side HIP runtime calls only.
#include "hispmv_ctx.h"

using namespace hispmv;

namespace hispmv {

void free_batch_plans(hispmv_ctx* c) {
    for (auto& p : c->batch_plans) {
        for (auto& g : p.slot) {
            if (g.launched && g.done) (void)hipEventSynchronize(g.done);
            if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
            if (g.done) { (void)hipEventDestroy(g.done); g.done = nullptr; }
            g.launched = false;
        }
        if (p.graph_src) { (void)hipGraphDestroy(p.graph_src); p.graph_src = nullptr; }
        for (auto& l : p.launches) { dev_free(l.d_table); dev_free(l.d_table2); dev_free(l.d_items); dev_free(l.d_sync); }
    }
    c->batch_plans.clear();
}

}  // namespace hispmv

// Builds the launches of a batch call: every column tile of every sparse handle is one "part"; parts with the same
// workgroup size share a grid (largest first, so that the small ones fill the tail), ONE fix-up launch finishes the cut
// rows of all parts (each on its own output: y for tile 0, the handle's partial vector for tile t > 0), ONE merge launch
// adds the partial vectors of the column-tiled matrices to their y.
static int build_batch_plan_with(hispmv_ctx* c, hispmv_ctx::BatchPlan& plan, int32_t n, const int32_t* idx, const float* const* d_x,
                                 const float* const* bias, float* const* d_y, float beta, bool long_groups, bool& used_long_groups, bool& step_taken);
// The batch layouts (groups up to four times as long, hispmv_choose.h) pay under the step kernel and cost under the grids on two lanes
// (the set 0.289 -> 0.302 ms): a call that turns out not to qualify for the step kernel is planned again from the parts' first layouts.
static int build_batch_plan(hispmv_ctx* c, hispmv_ctx::BatchPlan& plan, int32_t n, const int32_t* idx, const float* const* d_x,
                            const float* const* bias, float* const* d_y, float beta) {
    bool used = false, step = false;
    int rc = build_batch_plan_with(c, plan, n, idx, d_x, bias, d_y, beta, true, used, step);
    if (rc == HISPMV_OK && used && !step) {
        for (auto& l : plan.launches) { dev_free(l.d_table); dev_free(l.d_table2); dev_free(l.d_items); dev_free(l.d_sync); }
        plan.launches.clear();
        rc = build_batch_plan_with(c, plan, n, idx, d_x, bias, d_y, beta, false, used, step);
    }
    return rc;
}
static int build_batch_plan_with(hispmv_ctx* c, hispmv_ctx::BatchPlan& plan, int32_t n, const int32_t* idx, const float* const* d_x,
                                 const float* const* bias, float* const* d_y, float beta, bool long_groups, bool& used_long_groups, bool& step_taken) {
    used_long_groups = false; step_taken = false;
    struct Ref { int i; size_t t; };
    struct Item { std::vector<Ref> refs; int threads; int64_t slices; bool strays = false; };      // strays: a part with stray slots (a grid class of its own)
    std::vector<Item> items;
    std::vector<Ref> refs;
    // (a call that shares the chip between its matrices -- two lanes: at least batch_streams_min_bytes of streams -- takes a part's
    // BATCH layout where it has one: groups twice as long, hispmv_choose.h)
    const bool shared_chip = c->batch_streams > 1 && plan.stream_bytes >= c->batch_streams_min_bytes && !std::getenv("HISPMV_NO_BATCH_LAYOUT");
    auto dev_of = [&](const Ref& r) -> SpmvDeviceMatrix& {
        Matrix::Part& p = c->mats[idx[r.i]]->parts[r.t];
        if (shared_chip && long_groups && p.has_batch_dev) { used_long_groups = true; return p.batch_dev; }
        return p.dev;
    };
    auto out_of = [&](const Ref& r) -> float* {
        Matrix& m = *c->mats[idx[r.i]];
        return r.t == 0 ? d_y[r.i] : m.d_ypart + (r.t - 1) * (size_t)kMaxBatch * m.rows;
    };
    const bool pin = !std::getenv("HISPMV_NO_XCD_PIN");
    auto upload_table0 = [&](hispmv_ctx::BatchLaunch& l, const void* host, size_t bytes) -> int {
        HIP_TRY(c, hipMalloc(&l.d_table, bytes));
        HIP_TRY(c, hipMemcpy(l.d_table, host, bytes, hipMemcpyHostToDevice));
        return HISPMV_OK;
    };
    {   // dense overlay handles: one grid for all of them, the largest first (the small ones fill its tail; launched one
        // after the other, the 512..2048-wide GeMVs of cpu/run_gemv.sh cost a launch latency each: 77 us for the five
        // sizes against 56 us of streaming)
        std::vector<int> dense;
        for (int i = 0; i < n; ++i) if (c->mats[idx[i]]->dense) dense.push_back(i);
        std::stable_sort(dense.begin(), dense.end(), [&](int a, int b) {
            const Matrix& ma = *c->mats[idx[a]]; const Matrix& mb = *c->mats[idx[b]];
            return (int64_t)ma.rows * ma.cols > (int64_t)mb.rows * mb.cols;
        });
        for (size_t k0 = 0; k0 < dense.size(); k0 += kMultiMax) {
            hispmv_ctx::BatchLaunch l;
            l.kind = 4;
            for (size_t k = k0; k < std::min(dense.size(), k0 + (size_t)kMultiMax); ++k) {
                const int i = dense[k];
                const Matrix& m = *c->mats[idx[i]];
                l.gemv.push_back(GemvEntry{m.d_dense, d_x[i], bias[i], d_y[i], m.rows, m.cols, beta, 0});
            }
            plan.launches.push_back(std::move(l));
            const int rc0 = upload_table0(plan.launches.back(), plan.launches.back().gemv.data(), plan.launches.back().gemv.size() * sizeof(GemvEntry));
            if (rc0 != HISPMV_OK) return rc0;
        }
    }
    // (+ 1 << 24: matrices whose x the kernel keeps in the LDS -- another LDS size, another kernel instantiation)
    auto tts_class = [](const Matrix& m) {
        return m.parts[0].tdev.staging_floats + (m.parts.size() == 1 && tts_x_in_lds(m.parts[0].tdev, 1) ? (1 << 24) : 0) + (m.parts[0].tdev.zero_fill << 25);
    };
    std::vector<int> tts_classes;                // staging size = the geometry (hispmv_tts.h): small 13 K, standard 28 K, tall 23 K, paired 11 K
    for (int i = 0; i < n; ++i) {
        const Matrix& m = *c->mats[idx[i]];
        if (m.dense || m.format != 1) continue;
        const int cls = tts_class(m);
        if (std::find(tts_classes.begin(), tts_classes.end(), cls) == tts_classes.end()) tts_classes.push_back(cls);
    }
    std::sort(tts_classes.begin(), tts_classes.end());
    for (const int geometry : tts_classes) {
        // transposed tile streams: their row tiles share one grid per geometry (the launch's workgroup size and LDS size
        // are those of its entries: tiles of different geometries must not ride together)
        hispmv_ctx::BatchLaunch l;
        l.kind = 3;
        auto flush = [&]() -> int {
            if (l.tts.empty()) return HISPMV_OK;
            plan.launches.push_back(std::move(l));
            const int rc0 = upload_table0(plan.launches.back(), plan.launches.back().tts.data(), plan.launches.back().tts.size() * sizeof(TtsEntry));
            l = hispmv_ctx::BatchLaunch{};
            l.kind = 3;
            return rc0;
        };
        std::vector<int> order;                  // matrices with the longest tiles first (a CU holds one tile at a time)
        for (int i = 0; i < n; ++i) {
            const Matrix& m = *c->mats[idx[i]];
            if (m.dense || m.format != 1) continue;
            if (tts_class(m) == geometry) order.push_back(i);
        }
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
            const Matrix& ma = *c->mats[idx[a]]; const Matrix& mb = *c->mats[idx[b]];
            return ma.n_elems / std::max(1, ma.parts[0].tdev.n_tiles) > mb.n_elems / std::max(1, mb.parts[0].tdev.n_tiles);
        });
        for (int i : order) {
            Matrix& m = *c->mats[idx[i]];
            if (l.tts.size() + m.parts.size() > (size_t)kMultiMax) { const int rc0 = flush(); if (rc0 != HISPMV_OK) return rc0; }
            // the column parts of a tall-geometry matrix: one item, pinned to XCD subsets (part 0 writes y with the bias,
            // part t > 0 alpha*A_t*x into the handle's partial vector: the merge launch adds it)
            const bool pinned = pin && (m.parts.size() == 2 || m.parts.size() == 4);
            for (size_t t = 0; t < m.parts.size(); ++t) {
                l.tts.push_back(TtsEntry{m.parts[t].tdev, d_x[i], t == 0 ? bias[i] : nullptr, out_of(Ref{i, t}), t == 0 ? beta : 0.0f, 0});
                if (t == 0) l.weight += m.n_slices * (int64_t)kWideSliceBytes * 5 / 2;
                if (!pinned) l.item_tiles.push_back(1);
            }
            if (pinned) l.item_tiles.push_back((uint8_t)m.parts.size());
        }
        const int rc0 = flush();
        if (rc0 != HISPMV_OK) return rc0;
    }
    for (int i = 0; i < n; ++i) {
        const Matrix& m = *c->mats[idx[i]];
        if (m.dense || m.format == 1) continue;
        if (m.l2_tiles && pin) {                 // the L2-sized column tiles of a matrix: one item, pinned to XCD subsets
            Item it{{}, m.parts[0].dev.block_threads, 0};
            for (size_t t = 0; t < m.parts.size(); ++t) { it.refs.push_back(Ref{i, t}); it.slices += m.parts[t].dev.n_slices; }
            items.push_back(std::move(it));
        } else {
            for (size_t t = 0; t < m.parts.size(); ++t) { const SpmvDeviceMatrix& d = dev_of(Ref{i, t}); items.push_back(Item{{Ref{i, t}}, d.block_threads, d.n_slices, d.has_strays}); }
        }
    }
    std::stable_sort(items.begin(), items.end(), [&](const Item& a, const Item& b) {
        if (a.threads != b.threads) return a.threads > b.threads;
        if (a.strays != b.strays) return !a.strays;             // (parts with stray slots: their own grid, the kernel instantiation that fetches them)
        return a.slices > b.slices;
    });
    const size_t first_slice_launch = plan.launches.size();
    for (const Item& it : items) for (const Ref& r : it.refs) refs.push_back(r);
    auto upload_table = [&](hispmv_ctx::BatchLaunch& l, const void* host, size_t bytes) -> int {
        HIP_TRY(c, hipMalloc(&l.d_table, bytes));
        HIP_TRY(c, hipMemcpy(l.d_table, host, bytes, hipMemcpyHostToDevice));
        return HISPMV_OK;
    };
    int rc;
    for (size_t k = 0; k < items.size();) {                     // slice kernels, one launch per class
        const int threads = items[k].threads;
        hispmv_ctx::BatchLaunch l;
        l.kind = 0;
        std::vector<MultiEntry> entries;
        const bool strays = items[k].strays;
        while (k < items.size() && items[k].threads == threads && items[k].strays == strays && entries.size() + items[k].refs.size() <= (size_t)kMultiMax) {
            for (const Ref& r : items[k].refs) {
                SpmvDeviceMatrix& d = dev_of(r);
                MultiEntry e{};
                e.words = d.words; e.hdr = d.hdr; e.groups = d.groups; e.frags = d.frags;
                e.x = d_x[r.i]; e.bias = r.t == 0 ? bias[r.i] : nullptr; e.y = out_of(r); e.carry = d.carry;
                e.n_slices = d.n_slices; e.group_slices = d.group_slices; e.lds_floats = d.lds_floats; e.ytile_floats = d.ytile_floats;
                e.cols = d.cols; e.rows = d.rows;
                e.beta = r.t == 0 ? beta : 0.0f;
                entries.push_back(e);
                l.parts.push_back(&d);
            }
            l.item_tiles.push_back((uint8_t)items[k].refs.size());
            ++k;
        }
        l.multi = entries;
        plan.launches.push_back(std::move(l));
        if ((rc = upload_table(plan.launches.back(), entries.data(), entries.size() * sizeof(MultiEntry))) != HISPMV_OK) return rc;
    }
    // Launch order = stream assignment (main launch k goes to lane k mod lanes, hispmv_spmv_device_batch).  HISPMV_BATCH_ORDER=
    // small_first: the slice grids of SMALL workgroups first (256 threads, then 512, 1024), ahead of the tile streams: a
    // 256-thread workgroup (4 wavefronts, a few KiB of LDS) fits on a CU NEXT TO a 1024-thread slice workgroup (91 VGPRs: five
    // wavefronts per SIMD; windows of <= 115 KiB), so launched together the two grids share CUs -- the small matrices' L2
    // gathers ride under the HBM-bound stream of the large ones -- while a tile (145 KiB, 16 x 126 VGPRs) shares with nobody.
    if (c->batch_order == 1) {
        std::vector<hispmv_ctx::BatchLaunch> slices(std::make_move_iterator(plan.launches.begin() + (long)first_slice_launch),
                                                    std::make_move_iterator(plan.launches.end()));
        plan.launches.erase(plan.launches.begin() + (long)first_slice_launch, plan.launches.end());
        std::reverse(slices.begin(), slices.end());
        // small slice grids, then the large ones, then whatever was there before (dense, tile streams)
        std::vector<hispmv_ctx::BatchLaunch> rest = std::move(plan.launches);
        plan.launches.clear();
        for (auto& l : slices) plan.launches.push_back(std::move(l));
        for (auto& l : rest) plan.launches.push_back(std::move(l));
    }
    // The step kernel: when the call shares the chip between its matrices and every main launch is a slice grid (1024- or 256-thread
    // plans, no XCD-pinned column tiles) or a tile-stream grid of the standard geometry, the main launches collapse into ONE launch
    // whose items -- a 1024-thread group, four 256-thread groups, a tile -- are drawn from a queue by one persistent workgroup per CU
    // (hispmv_kernels.hip: spmv_step_kernel; per-CU occupancy before / after: profiles/r4_experiments/step_kernel/).
    if (c->step_kernel && shared_chip && !c->cu_split && !c->batch_graphs) {
        bool ok = true;
        size_t lds = 0;
        bool strays = false;
        for (const auto& l : plan.launches) {
            if (l.kind == 4) ok = false;
            if (l.kind == 0) {
                for (uint8_t t : l.item_tiles) ok = ok && t == 1;
                for (const SpmvDeviceMatrix* d : l.parts) {
                    ok = ok && (d->block_threads == 1024 || d->block_threads == 256);
                    const size_t one = ((size_t)d->lds_floats + (size_t)d->ytile_floats * (d->block_threads / 64)) * sizeof(float);
                    lds = std::max(lds, d->block_threads == 256 ? 4 * one : one);
                    strays = strays || d->has_strays;
                }
            }
            if (l.kind == 3) {
                for (uint8_t t : l.item_tiles) ok = ok && t == 1;
                for (const TtsEntry& e : l.tts) {
                    ok = ok && e.m.zero_fill == 0 && e.m.threads == 1024 && !tts_x_in_lds(e.m, 1);
                    lds = std::max(lds, tts_tile_lds_bytes(e.m));
                }
            }
        }
        ok = ok && lds <= 160 * 1024 - 256;
        if (ok) {
            struct QItem { uint32_t a, b; double cost; int cls; };      // cls 0 = slice items, 1 = tiles
            std::vector<QItem> q[2];
            std::vector<MultiEntry> slice_table;
            std::vector<TtsEntry> tts_table;
            for (const auto& l : plan.launches) {
                if (l.kind == 0) {
                    for (size_t e = 0; e < l.multi.size(); ++e) {
                        const SpmvDeviceMatrix& d = *l.parts[e];
                        const uint32_t entry = (uint32_t)slice_table.size();
                        slice_table.push_back(l.multi[e]);
                        const int64_t ng = d.n_slices > 0 ? d.n_groups : 0;
                        // cost of a group in "microseconds of a CU": 0.27 us per compact 6 KiB slice read from an LDS window (measured: 32.7 ms of
                        // CU time for the 1024-thread groups of the set), 2.5-fold for groups that gather through L2; + the window staging
                        const bool window = d.lds_floats > 0;
                        const double per_slice = window ? 0.27 : 0.68;
                        if (d.block_threads == 1024) {
                            for (int64_t g = 0; g < ng; ++g) {
                                const int64_t n_here = std::min<int64_t>(d.group_slices, d.n_slices - g * d.group_slices);
                                q[0].push_back(QItem{0u | (entry << 8), (uint32_t)g, 1.0 + per_slice * (double)n_here + (double)d.lds_floats * 4.0 / 40000.0, 0});
                            }
                        } else {
                            for (int64_t g = 0; g < ng; g += 4) {
                                const int64_t n_here = std::min<int64_t>(d.group_slices, d.n_slices - g * d.group_slices);
                                // (four groups side by side, each gathering through L2 or from a small window: 10 - 20 us measured; priced so that
                                // they sort AHEAD of the 11 - 16 us groups of the smallest 1024-thread plans -- at the very end of the queue, where
                                // only a few CUs come free at a time, they stretched the step by 10 us)
                                static const double f256 = std::getenv("HISPMV_STEP_COST256") ? std::atof(std::getenv("HISPMV_STEP_COST256")) : 8.0;
                                q[0].push_back(QItem{1u | (entry << 8), (uint32_t)g, 1.0 + f256 * (double)n_here + (double)d.lds_floats * 4.0 / 40000.0, 0});
                            }
                        }
                    }
                }
                if (l.kind == 3) {
                    for (const TtsEntry& te : l.tts) {
                        const uint32_t entry = (uint32_t)tts_table.size();
                        tts_table.push_back(te);
                        // the slices of every tile, from the device tables (the host copies are gone after the upload)
                        std::vector<int4> tiles((size_t)te.m.n_tiles);
                        HIP_TRY(c, hipMemcpy(tiles.data(), te.m.tiles, tiles.size() * sizeof(int4), hipMemcpyDeviceToHost));
                        int n_blocks = 0;
                        for (const int4& t : tiles) n_blocks = std::max(n_blocks, t.z + t.w);
                        std::vector<int4> blocks((size_t)n_blocks * 2);
                        if (n_blocks > 0) HIP_TRY(c, hipMemcpy(blocks.data(), te.m.blocks, blocks.size() * sizeof(int4), hipMemcpyDeviceToHost));
                        // 0.85 us per 1024-element slice with 23 lines of x per gather (soc-Pokec), scaled by the line model of DESIGN.md 2.2
                        double lines = 23.0;
                        for (int i = 0; i < n; ++i) { const Matrix& m = *c->mats[idx[i]]; if (!m.dense && m.format == 1 && m.parts[0].tdev.words == te.m.words && m.tts_lines_per_gather > 0) lines = m.tts_lines_per_gather; }
                        static const double tile_scale = std::getenv("HISPMV_STEP_TILE_SCALE") ? std::atof(std::getenv("HISPMV_STEP_TILE_SCALE")) : 1.0;
                        const double per_slice = tile_scale * 0.85 * (50.0 + 2.8 * lines) / (50.0 + 2.8 * 23.0);
                        for (int t = 0; t < te.m.n_tiles; ++t) {
                            int64_t sl = 0;
                            for (int b = tiles[(size_t)t].z; b < tiles[(size_t)t].z + tiles[(size_t)t].w; ++b) sl += blocks[(size_t)b * 2].y;
                            q[1].push_back(QItem{2u | (entry << 8), (uint32_t)t, 2.0 + per_slice * (double)sl + 0.5 * (double)tiles[(size_t)t].w, 1});
                        }
                    }
                }
            }
            // (a call of tile streams alone keeps its grid: two power-law matrices cut into ~300 tiles of 35 - 110 us each leave the queue
            // nothing short to end with -- 0.179 against 0.165 ms, profiles/r4_experiments/step_kernel/powerlaw.json)
            if (slice_table.size() < (1u << 20) && tts_table.size() < (1u << 20) && !q[0].empty()) {
                const int W = std::max(1, c->n_cus);
                // the order of the queue: host-only code (hispmv_choose.cpp: order_step_queue; tests/test_step_queue.py)
                std::vector<double> cost[2];
                for (int k = 0; k < 2; ++k) for (const QItem& it : q[k]) cost[k].push_back(it.cost);
                std::vector<QItem> order;
                for (const auto& pr : order_step_queue(cost[0], cost[1], W, c->step_order)) order.push_back(q[pr.first][(size_t)pr.second]);
                std::vector<uint32_t> packed;
                packed.reserve(order.size() * 2);
                for (const QItem& it : order) { packed.push_back(it.a); packed.push_back(it.b); }
                hispmv_ctx::BatchLaunch L;
                L.kind = 6;
                L.n_items = (unsigned)order.size();
                L.step_workgroups = (int)std::min<size_t>((size_t)W, order.size());
                L.step_lds = lds;
                L.step_strays = strays;
                for (const auto& l : plan.launches) L.weight += l.weight;
                // the launches it replaces go; their device tables with them
                std::vector<hispmv_ctx::BatchLaunch> keep;
                for (auto& l : plan.launches) {
                    if (l.kind == 0 || l.kind == 3) { dev_free(l.d_table); dev_free(l.d_table2); }
                    else keep.push_back(std::move(l));
                }
                plan.launches.clear();
                plan.launches.push_back(std::move(L));
                for (auto& l : keep) plan.launches.push_back(std::move(l));
                hispmv_ctx::BatchLaunch& S = plan.launches.front();
                if (!slice_table.empty()) {
                    HIP_TRY(c, hipMalloc(&S.d_table, slice_table.size() * sizeof(MultiEntry)));
                    HIP_TRY(c, hipMemcpy(S.d_table, slice_table.data(), slice_table.size() * sizeof(MultiEntry), hipMemcpyHostToDevice));
                }
                if (!tts_table.empty()) {
                    HIP_TRY(c, hipMalloc(&S.d_table2, tts_table.size() * sizeof(TtsEntry)));
                    HIP_TRY(c, hipMemcpy(S.d_table2, tts_table.data(), tts_table.size() * sizeof(TtsEntry), hipMemcpyHostToDevice));
                }
                HIP_TRY(c, hipMalloc(&S.d_items, packed.size() * sizeof(uint32_t)));
                HIP_TRY(c, hipMemcpy(S.d_items, packed.data(), packed.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
                HIP_TRY(c, hipMalloc((void**)&S.d_sync, 2 * sizeof(unsigned)));
                HIP_TRY(c, hipMemset(S.d_sync, 0, 2 * sizeof(unsigned)));
                step_taken = true;
            }
        }
    }
    if (long_groups && used_long_groups && !step_taken) return HISPMV_OK;        // (planned again from the first layouts: build_batch_plan)
    std::vector<Ref> fixrefs = refs;                            // + tile streams that cut long rows into pieces
    for (int i = 0; i < n; ++i) {
        const Matrix& m = *c->mats[idx[i]];
        if (!m.dense && m.format == 1)
            for (size_t t = 0; t < m.parts.size(); ++t) if (m.parts[t].tdev.n_fix > 0) fixrefs.push_back(Ref{i, t});
    }
    // Lanes of the main launches: round-robin in launch order, and the lane with the most (time-weighted) bytes is the CALLER'S
    // stream with its launches ENQUEUED FIRST (HISPMV_BATCH_LANES=rr: plain round-robin), so that the tail launch follows the
    // last-finishing chain on the same queue instead of behind a cross-queue dependency.  Replayed as a graph the runtime
    // keeps the chain of the FIRST captured root on the launch stream's queue: with the 1024-thread slice grid captured first
    // (the structured set, where it finishes last) the tail starts 6.7 us behind its end instead of 12 (kernel timelines in
    // profiles/r3_experiments/step_structure.json): 0.288-0.294 against 0.291-0.299 ms per step; in the pessimistic family
    // the tile streams are the heavier chain and the order stays what it was.  (Measured and dropped: longest-processing-time
    // assignment -- the 256-thread grid then follows the 1024-thread one and the pessimistic family loses 1-4 %; the heavy
    // lane on the caller's stream but captured second: 0.302-0.303.)
    {
        plan.lanes = plan.stream_bytes >= c->batch_streams_min_bytes ? c->batch_streams : 1;
        std::vector<hispmv_ctx::BatchLaunch*> mains;
        for (auto& l : plan.launches) {
            if (l.kind == 0) for (const SpmvDeviceMatrix* d : l.parts) l.weight += d->n_slices * (int64_t)kWideSliceBytes;
            // (a tile stream runs at ~2.5 TB/s against ~6.5 for a slice stream: its bytes count 2.5-fold; set when the entries were made)
            if (l.kind == 4) for (const GemvEntry& e : l.gemv) l.weight += 4 * (int64_t)e.rows * e.cols;
            mains.push_back(&l);
        }
        plan.lanes = std::max(1, std::min<int>(plan.lanes, (int)mains.size()));
        // round-robin in launch order (tile streams, 1024-thread slices, 256-thread slices, ...) ...
        int k = 0;
        for (hispmv_ctx::BatchLaunch* l : mains) l->lane = k++ % plan.lanes;
        if (c->batch_lanes_heavy_first && plan.lanes >= 2) {
            // ... and the lane with the most (time-weighted) bytes becomes the caller's stream, its launches enqueued first
            std::vector<int64_t> load((size_t)plan.lanes, 0);
            for (hispmv_ctx::BatchLaunch* l : mains) load[(size_t)l->lane] += l->weight;
            const int heavy = (int)(std::max_element(load.begin(), load.end()) - load.begin());
            for (hispmv_ctx::BatchLaunch* l : mains) l->lane = l->lane == heavy ? 0 : l->lane == 0 ? heavy : l->lane;
            std::stable_sort(plan.launches.begin(), plan.launches.end(), [](const hispmv_ctx::BatchLaunch& x, const hispmv_ctx::BatchLaunch& y) { return x.lane < y.lane; });
        }
    }
    // The tail: ONE launch that finishes the cut rows and merges the partial vectors of column-tiled matrices (which then
    // apply their own fix-ups) when every tiled matrix of the call carries its row -> fix table and the tables fit one
    // launch; otherwise a fix-up launch and a merge launch.
    std::vector<int> tiled;
    bool fused = !std::getenv("HISPMV_NO_FUSED_TAIL");
    for (int i = 0; i < n; ++i) {
        const Matrix& m = *c->mats[idx[i]];
        if (m.dense || m.parts.size() < 2) continue;
        tiled.push_back(i);
        fused = fused && m.d_fix_of_row != nullptr;
    }
    if (fused) {
        std::vector<Ref> plain;                                 // parts whose cut rows the fix-up blocks finish
        for (const Ref& r : fixrefs) if (c->mats[idx[r.i]]->parts.size() < 2) plain.push_back(r);
        fused = plain.size() <= (size_t)kMultiMax && tiled.size() <= (size_t)kMultiMax;
        // HISPMV_LANE_TAILS=1 (experiment, r4_lane_tails.sh): one tail per LANE, enqueued on the lane's own stream right behind its main
        // launches -- no cross-stream join in front of it -- with the cut rows and merges of the matrices whose parts all ran in that
        // lane; what is left (a matrix with parts in two lanes) keeps the joined tail.
        static const bool lane_tails = std::getenv("HISPMV_LANE_TAILS") != nullptr;
        if (fused && lane_tails && plan.lanes > 1) {
            auto lane_of_out = [&](const float* out, const SpmvDeviceMatrix* dev) -> int {
                for (const auto& l : plan.launches) {
                    if (l.kind == 0) for (const SpmvDeviceMatrix* d : l.parts) if (d == dev) return l.lane;
                    if (l.kind == 3) for (const TtsEntry& e : l.tts) if (e.y == out) return l.lane;
                }
                return -1;
            };
            std::vector<std::vector<Ref>> lane_plain((size_t)plan.lanes);
            std::vector<std::vector<int>> lane_tiled((size_t)plan.lanes);
            std::vector<Ref> rest_plain; std::vector<int> rest_tiled;
            for (const Ref& r : plain) { const int ln = lane_of_out(out_of(r), &dev_of(r)); if (ln >= 0) lane_plain[(size_t)ln].push_back(r); else rest_plain.push_back(r); }
            for (int i : tiled) {
                Matrix& m = *c->mats[idx[i]];
                int ln = -2;
                for (size_t t = 0; t < m.parts.size(); ++t) { const int q = lane_of_out(out_of(Ref{i, t}), &m.parts[t].dev); ln = ln == -2 ? q : (ln == q ? ln : -1); }
                if (ln >= 0) lane_tiled[(size_t)ln].push_back(i); else rest_tiled.push_back(i);
            }
            auto make_tail = [&](const std::vector<Ref>& pl, const std::vector<int>& ti, int lane, bool in_lane) -> int {
                hispmv_ctx::BatchLaunch l;
                l.kind = 5; l.lane = lane; l.in_lane = in_lane;
                std::vector<MultiFixEntry> fix; std::vector<TailMergeEntry> mrg;
                bool any = !ti.empty();
                for (const Ref& r : pl) {
                    SpmvDeviceMatrix& d = dev_of(r);
                    fix.push_back(MultiFixEntry{d.fix_short, d.carry, out_of(r), d.n_fix_short, 0});
                    l.fix_counts.push_back(d.n_fix_short);
                    l.parts.push_back(&d); l.ys.push_back(out_of(r));
                    any = any || d.n_fix_short > 0 || d.n_fix_long > 0;
                }
                for (int i : ti) {
                    Matrix& m = *c->mats[idx[i]];
                    TailMergeEntry e{};
                    e.y = d_y[i]; e.parts = m.d_ypart; e.part_stride = (long long)kMaxBatch * m.rows; e.n_parts = (int32_t)m.parts.size() - 1; e.rows = m.rows;
                    e.fix_of_row = m.d_fix_of_row;
                    for (size_t t = 0; t < m.parts.size(); ++t) { e.fix[t] = m.parts[t].dev.fix_short; e.carry[t] = m.parts[t].dev.carry; }
                    mrg.push_back(e);
                    l.rows.push_back(m.rows);
                }
                if (!any) return HISPMV_OK;
                plan.launches.push_back(std::move(l));
                hispmv_ctx::BatchLaunch& L = plan.launches.back();
                int r2;
                if (!fix.empty() && (r2 = upload_table(L, fix.data(), fix.size() * sizeof(MultiFixEntry))) != HISPMV_OK) return r2;
                if (!mrg.empty()) {
                    HIP_TRY(c, hipMalloc(&L.d_table2, mrg.size() * sizeof(TailMergeEntry)));
                    HIP_TRY(c, hipMemcpy(L.d_table2, mrg.data(), mrg.size() * sizeof(TailMergeEntry), hipMemcpyHostToDevice));
                }
                return HISPMV_OK;
            };
            for (int ln = 0; ln < plan.lanes; ++ln)
                if ((rc = make_tail(lane_plain[(size_t)ln], lane_tiled[(size_t)ln], ln, true)) != HISPMV_OK) return rc;
            return make_tail(rest_plain, rest_tiled, 0, false);
        }
        if (fused) {
            hispmv_ctx::BatchLaunch l;
            l.kind = 5;
            std::vector<MultiFixEntry> fix;
            std::vector<TailMergeEntry> mrg;
            bool any = !tiled.empty();
            for (const Ref& r : plain) {
                SpmvDeviceMatrix& d = dev_of(r);
                fix.push_back(MultiFixEntry{d.fix_short, d.carry, out_of(r), d.n_fix_short, 0});
                l.fix_counts.push_back(d.n_fix_short);
                l.parts.push_back(&d); l.ys.push_back(out_of(r));               // (long chains: their own launches behind the tail)
                any = any || d.n_fix_short > 0 || d.n_fix_long > 0;
            }
            for (int i : tiled) {
                Matrix& m = *c->mats[idx[i]];
                TailMergeEntry e{};
                e.y = d_y[i]; e.parts = m.d_ypart; e.part_stride = (long long)kMaxBatch * m.rows; e.n_parts = (int32_t)m.parts.size() - 1; e.rows = m.rows;
                e.fix_of_row = m.d_fix_of_row;
                for (size_t t = 0; t < m.parts.size(); ++t) { e.fix[t] = m.parts[t].dev.fix_short; e.carry[t] = m.parts[t].dev.carry; }
                mrg.push_back(e);
                l.rows.push_back(m.rows);
            }
            if (!any) return HISPMV_OK;
            plan.launches.push_back(std::move(l));
            hispmv_ctx::BatchLaunch& L = plan.launches.back();
            if (!fix.empty() && (rc = upload_table(L, fix.data(), fix.size() * sizeof(MultiFixEntry))) != HISPMV_OK) return rc;
            if (!mrg.empty()) {
                HIP_TRY(c, hipMalloc(&L.d_table2, mrg.size() * sizeof(TailMergeEntry)));
                HIP_TRY(c, hipMemcpy(L.d_table2, mrg.data(), mrg.size() * sizeof(TailMergeEntry), hipMemcpyHostToDevice));
            }
            return HISPMV_OK;
        }
    }
    for (size_t k = 0; k < fixrefs.size(); k += kMultiMax) {    // fix-up of the cut rows
        hispmv_ctx::BatchLaunch l;
        l.kind = 1;
        std::vector<MultiFixEntry> fix;
        bool any = false;
        for (size_t q = k; q < std::min(fixrefs.size(), k + kMultiMax); ++q) {
            SpmvDeviceMatrix& d = dev_of(fixrefs[q]);
            fix.push_back(MultiFixEntry{d.fix_short, d.carry, out_of(fixrefs[q]), d.n_fix_short, 0});
            l.parts.push_back(&d);
            l.ys.push_back(out_of(fixrefs[q]));
            any = any || d.n_fix_short > 0 || d.n_fix_long > 0;
        }
        if (!any) continue;
        plan.launches.push_back(std::move(l));
        if ((rc = upload_table(plan.launches.back(), fix.data(), fix.size() * sizeof(MultiFixEntry))) != HISPMV_OK) return rc;
    }
    std::vector<MultiMergeEntry> merges;
    std::vector<int32_t> merge_rows;
    auto flush_merges = [&]() -> int {
        if (merges.empty()) return HISPMV_OK;
        hispmv_ctx::BatchLaunch l;
        l.kind = 2; l.rows = merge_rows;
        plan.launches.push_back(std::move(l));
        const int r2 = upload_table(plan.launches.back(), merges.data(), merges.size() * sizeof(MultiMergeEntry));
        merges.clear(); merge_rows.clear();
        return r2;
    };
    for (int i : tiled) {                                       // merge of the column-tile partial vectors
        Matrix& m = *c->mats[idx[i]];
        merges.push_back(MultiMergeEntry{d_y[i], m.d_ypart, (long long)kMaxBatch * m.rows, (int32_t)m.parts.size() - 1, m.rows});
        merge_rows.push_back(m.rows);
        if ((int)merges.size() == kMultiMax && (rc = flush_merges()) != HISPMV_OK) return rc;
    }
    return flush_merges();
}

HISPMV_API int hispmv_spmv_device_batch(hispmv_ctx* c, int32_t n, const int32_t* idx, const float* const* d_x,
                                        const float* const* d_bias, float* const* d_y, float alpha, float beta, void* stream) {
    if (!c) return HISPMV_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (stream) c->user_stream = (hipStream_t)stream;
    return spmv_batch_locked(c, n, idx, d_x, d_bias, d_y, alpha, beta, stream ? (hipStream_t)stream : c->stream);
}

namespace hispmv {
int spmv_batch_locked(hispmv_ctx* c, int32_t n, const int32_t* idx, const float* const* d_x, const float* const* d_bias,
                      float* const* d_y, float alpha, float beta, hipStream_t s) {
    if (n < 0 || (n > 0 && (!idx || !d_x || !d_y || (beta != 0.0f && !d_bias)))) return fail(c, HISPMV_EINVAL, "NULL argument");
    for (int i = 0; i < n; ++i) {
        if (idx[i] < 0 || idx[i] >= (int)c->mats.size()) return fail(c, HISPMV_EINVAL, "Matrix idx out of range");
        const Matrix& m = *c->mats[idx[i]];
        if (!m.loaded) return fail(c, HISPMV_ESTATE, "spmv_device_batch called before load_matrices");
        if (!d_x[i] || !d_y[i] || (beta != 0.0f && !d_bias[i])) return fail(c, HISPMV_EINVAL, "NULL device vector");
        for (int k = 0; k < i; ++k) {
            if (d_y[k] == d_y[i]) return fail(c, HISPMV_EINVAL, "two matrices of a batch write the same y");
            // the carry buffers of cut rows belong to the handle: one SpMV per handle at a time (hispmv.h, threading)
            if (idx[k] == idx[i] && !m.dense) return fail(c, HISPMV_EINVAL, "the same sparse handle twice in one batch");
        }
    }
    HIP_TRY(c, hipSetDevice(c->device));
    const float* const* bias = d_bias;
    std::vector<const float*> no_bias;
    if (!bias) { no_bias.assign((size_t)n, nullptr); bias = no_bias.data(); }
    // the launches of this call signature: built once, replayed afterwards (beta enters the tables; alpha is a kernel argument)
    std::vector<uint64_t> key{(uint64_t)n, (uint64_t)__builtin_bit_cast(uint32_t, beta)};
    for (int i = 0; i < n; ++i) {
        key.push_back((uint64_t)idx[i]); key.push_back((uint64_t)(uintptr_t)d_x[i]);
        key.push_back((uint64_t)(uintptr_t)(beta != 0.0f ? bias[i] : nullptr)); key.push_back((uint64_t)(uintptr_t)d_y[i]);
    }
    hispmv_ctx::BatchPlan* plan = nullptr;
    for (auto& p : c->batch_plans) if (p.key == key) { plan = &p; break; }
    if (!plan) {
        if (c->batch_plans.size() >= 16) free_batch_plans(c);      // callers that keep changing their vectors: start over
        c->batch_plans.emplace_back();
        c->batch_plans.back().key = key;
        for (int i = 0; i < n; ++i) {
            const Matrix& mi = *c->mats[idx[i]];
            c->batch_plans.back().stream_bytes += mi.dense ? 4 * (int64_t)mi.rows * mi.cols : 8 * mi.nnz;
        }
        const int rc = build_batch_plan(c, c->batch_plans.back(), n, idx, d_x, bias, d_y, beta);
        if (rc != HISPMV_OK) {                                      // nothing half-built stays behind
            for (auto& l : c->batch_plans.back().launches) { dev_free(l.d_table); dev_free(l.d_table2); dev_free(l.d_items); dev_free(l.d_sync); }
            c->batch_plans.pop_back();
            return rc;
        }
        plan = &c->batch_plans.back();
    }
    {
        int64_t step_items = 0;
        for (const auto& l : plan->launches) if (l.kind == 6) step_items += l.n_items;
        c->last_batch[0] = (int64_t)plan->launches.size(); c->last_batch[1] = step_items > 0; c->last_batch[2] = step_items;
        c->last_batch[3] = c->cu_split ? 3 : plan->lanes;
    }
    // main launches (kinds 0 and 3) are independent of each other: spread over the caller's stream and the side streams;
    // the fix-up and merge launches follow on the caller's stream behind a join
    const int lanes = c->cu_split ? 3 : plan->lanes;
    // the launches, as one function of the stream: main launches spread over the caller's stream and the side streams (forked
    // from / joined to it with events), fix-up and merge behind the join
    auto enqueue = [&]() -> int {
        if (lanes > 1) {
            HIP_TRY(c, hipEventRecord(c->ev_fork, s));
            for (int i = 0; i + 1 < lanes; ++i) HIP_TRY(c, hipStreamWaitEvent(c->side[i], c->ev_fork, 0));
        }
        bool joined = lanes <= 1;
        for (const auto& l : plan->launches) {
            hipError_t e = hipSuccess;
            const bool is_main = l.kind == 0 || l.kind == 3 || l.kind == 4 || l.kind == 6 || (l.kind == 5 && l.in_lane);      // (a lane's own tail rides on the lane's stream)
            hipStream_t ls = s;
            if (is_main && lanes > 1) ls = l.lane == 0 ? s : c->side[l.lane - 1];
            if (is_main && c->cu_split) ls = l.kind == 3 ? c->side[1] : c->side[0];
            if (!is_main && !joined) {
                for (int i = 0; i + 1 < lanes; ++i) { HIP_TRY(c, hipEventRecord(c->ev_join[i], c->side[i])); HIP_TRY(c, hipStreamWaitEvent(s, c->ev_join[i], 0)); }
                joined = true;
            }
            if (l.kind == 0) e = launch_spmv_multi(l.parts.data(), (int)l.parts.size(), l.item_tiles.data(), (int)l.item_tiles.size(), (const MultiEntry*)l.d_table, alpha, ls);
            else if (l.kind == 3) e = launch_tts_multi(l.tts.data(), (int)l.tts.size(), l.item_tiles.data(), (int)l.item_tiles.size(), (const TtsEntry*)l.d_table, alpha, ls);
            else if (l.kind == 6) e = launch_spmv_step((const MultiEntry*)l.d_table, (const TtsEntry*)l.d_table2, l.d_items, l.n_items, l.d_sync, l.step_workgroups, l.step_lds, l.step_strays, alpha, ls);
            else if (l.kind == 4) e = launch_gemv_multi(l.gemv.data(), (int)l.gemv.size(), (const GemvEntry*)l.d_table, alpha, ls);
            else if (l.kind == 1) e = launch_fixup_multi(l.parts.data(), l.ys.data(), (int)l.parts.size(), (const MultiFixEntry*)l.d_table, alpha, ls);
            else if (l.kind == 5) {
                e = launch_tail_multi(l.fix_counts.data(), (int)l.fix_counts.size(), (const MultiFixEntry*)l.d_table, l.rows.data(), (int)l.rows.size(),
                                      (const TailMergeEntry*)l.d_table2, alpha, ls);
                for (size_t q = 0; e == hipSuccess && q < l.parts.size(); ++q)       // rows that span more than 32 slices: a wavefront per row
                    if (l.parts[q]->n_fix_long > 0) e = launch_fixup_long(*l.parts[q], l.ys[q], alpha, ls);
            }
            else e = launch_merge_multi(l.rows.data(), (int)l.rows.size(), (const MultiMergeEntry*)l.d_table, ls);
            if (e != hipSuccess) return hip_fail(c, e, l.kind == 0 ? "launch_spmv_multi" : l.kind == 1 ? "launch_fixup_multi" : l.kind == 3 ? "launch_tts_multi" : l.kind == 4 ? "launch_gemv_multi" : l.kind == 6 ? "launch_spmv_step" : l.kind == 5 ? "launch_tail_multi" : "launch_merge_multi");
        }
        if (!joined)
            for (int i = 0; i + 1 < lanes; ++i) { HIP_TRY(c, hipEventRecord(c->ev_join[i], c->side[i])); HIP_TRY(c, hipStreamWaitEvent(s, c->ev_join[i], 0)); }
        return HISPMV_OK;
    };
    // A two-stream call is captured into a HIP graph the second time its signature is seen (the first run sets the
    // kernels' attributes) and replayed from then on: the set's step 0.315-0.320 -> 0.309-0.310 ms -- the fork/join events
    // of the two streams become graph edges.  HISPMV_BATCH_GRAPH=0 switches it off; a stream that is being captured by the
    // caller, or a capture the runtime refuses, falls back to plain launches.
    // (a stream the CALLER is capturing takes plain launches: they become nodes of the caller's graph; replaying the library's
    // own graph into a capture recorded nothing on this runtime)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool caller_captures = hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone;
    (void)hipGetLastError();
    if (c->batch_graphs && lanes > 1 && !caller_captures && !c->cu_split) {      // (a graph replay maps its branches to the runtime's own queues: the CU masks would be lost)      // (one-stream calls: a graph launch costs more than their 2-4 plain launches, the model layers 50 -> 54 us)
        using Slot = hispmv_ctx::BatchPlan::GraphSlot;
        auto drop_graphs = [&]() {
            for (Slot& g : plan->slot) {
                if (g.launched && g.done) (void)hipEventSynchronize(g.done);
                if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
                g.launched = false;
            }
            if (plan->graph_src) { (void)hipGraphDestroy(plan->graph_src); plan->graph_src = nullptr; }
        };
        auto launch_slot = [&](Slot& g) -> bool {
            if (hipGraphLaunch(g.exec, s) != hipSuccess) { (void)hipGetLastError(); return false; }
            if (!g.done && hipEventCreateWithFlags(&g.done, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); g.done = nullptr; }
            // (without the event the executable can never be patched safely: its alpha stays what it is, see below)
            g.launched = g.done && hipEventRecord(g.done, s) == hipSuccess;
            if (!g.launched) (void)hipGetLastError();
            g.last_use = ++plan->use_counter;
            return true;
        };
        if (plan->graph_src) {
            // 1. an executable that already carries this alpha
            Slot* pick = nullptr;
            for (Slot& g : plan->slot) if (g.exec && g.alpha == alpha) { pick = &g; break; }
            if (!pick) {
                // 2. another alpha on the same call: no capture -- a second executable of the captured graph the first time,
                //    afterwards the executable used longest ago, patched once ITS last launch has completed
                Slot* victim = nullptr;
                for (Slot& g : plan->slot) if (!g.exec) { victim = &g; break; }
                if (victim) {
                    if (hipGraphInstantiate(&victim->exec, plan->graph_src, nullptr, nullptr, 0) == hipSuccess) { c->graph_instantiations++; victim->launched = false; }
                    else { (void)hipGetLastError(); victim->exec = nullptr; victim = nullptr; }
                }
                if (!victim) victim = plan->slot[0].last_use <= plan->slot[1].last_use ? &plan->slot[0] : &plan->slot[1];
                bool ok = victim->exec != nullptr;
                if (ok && victim->launched) ok = victim->done && hipEventSynchronize(victim->done) == hipSuccess;
                if (ok) ok = graph_set_alpha(victim->exec, plan->graph_src, alpha) == hipSuccess;
                if (ok) { victim->alpha = alpha; c->graph_alpha_updates++; pick = victim; }
                else (void)hipGetLastError();
            }
            if (pick && launch_slot(*pick)) return HISPMV_OK;
            drop_graphs();
        }
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (plan->runs >= 1 && hipStreamIsCapturing(s, &st) == hipSuccess && st == hipStreamCaptureStatusNone &&
            hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            const int rc = enqueue();
            hipGraph_t g = nullptr;
            const hipError_t e_end = hipStreamEndCapture(s, &g);
            if (rc != HISPMV_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
            drop_graphs();
            hipError_t e = e_end;
            Slot& g0 = plan->slot[0];
            if (e == hipSuccess) e = hipGraphInstantiate(&g0.exec, g, nullptr, nullptr, 0);
            if (e == hipSuccess) c->graph_instantiations++;
            plan->graph_src = g;
            if (e == hipSuccess) { g0.alpha = alpha; g0.launched = false; if (launch_slot(g0)) return HISPMV_OK; }
            (void)hipGetLastError();
            drop_graphs();
            c->batch_graphs = false;               // this runtime / stream does not take it: plain launches from here on
        } else {
            (void)hipGetLastError();
        }
    }
    plan->runs++;
    return enqueue();
}
}  // namespace hispmv

HISPMV_API int hispmv_batch_call_info(hispmv_ctx* c, int64_t out[4]) {
    if (!c || !out) return HISPMV_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (c->last_batch[0] < 0) return fail(c, HISPMV_ESTATE, "no batch call yet");
    for (int i = 0; i < 4; ++i) out[i] = c->last_batch[i];
    return HISPMV_OK;
}

HISPMV_API int hispmv_batch_graph_stats(hispmv_ctx* c, int64_t out[2]) {
    if (!c || !out) return HISPMV_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    out[0] = c->graph_instantiations; out[1] = c->graph_alpha_updates;
    return HISPMV_OK;
}

