// hispmv_plan.cpp -- see hispmv_plan.h.  Host-only, OpenMP.
#include "hispmv_plan.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace hispmv {

namespace {

inline int32_t col_of(uint64_t w) { return (int32_t)((w >> 32) & 0x7fffffffu); }

struct Cfg { int threads, slices, cap, per_cu; };   // slices == 0: one chunk per resident workgroup

// Scratch of one thread: how often each 64-byte block of x is touched by a range of slices, and which blocks
// the group stages in LDS.
struct BlockSet {
    std::vector<uint32_t> cnt;       // dense path: elements per block of the span
    std::vector<int32_t> table;      // hash path: open addressing, -1 = empty
    std::vector<uint32_t> tcnt;      //            elements per slot
    std::vector<int32_t> used;       //            slots to clear
    std::vector<std::pair<uint32_t, int32_t>> ranked;   // {elements, block}
    std::vector<int32_t> blocks;     // result: staged blocks, ascending
    int64_t global_elems = 0;        // result: elements whose block is not staged
    int64_t runs = 0;                // result: runs of consecutive staged blocks (= fragments, before the length cap)

    // Chooses the blocks slices [s0, s1) stage when at most `limit` blocks fit: all of them, or -- when up to
    // kSpill times as many are touched -- the `limit` most used ones (the other elements gather through L2).
    // Returns the number of staged blocks, or -1 when the group touches too many blocks for a window to pay.
    static constexpr int kSpill = 4;
    int64_t collect(const SliceStream& st, int64_t s0, int64_t s1, int64_t limit) {
        blocks.clear(); global_elems = 0; runs = 0;
        int lo = INT32_MAX, hi = -1;
        for (int64_t s = s0; s < s1; ++s) {
            lo = std::min(lo, st.hdr[s].x_base);
            hi = std::max(hi, st.hdr[s].x_base + st.hdr[s].x_span - 1);
        }
        if (hi < 0) return 0;
        const int32_t b_lo = lo / kFragBlock, b_hi = hi / kFragBlock;
        const int64_t span_blocks = (int64_t)b_hi - b_lo + 1;
        const uint64_t* w = st.words.data() + s0 * kSliceElems;
        const int64_t n = (s1 - s0) * kSliceElems;
        const int64_t most = kSpill * std::max<int64_t>(limit, 64);
        ranked.clear();
        if (span_blocks <= 8 * most) {          // dense enough: one counter per block of the span
            cnt.assign((size_t)span_blocks, 0);
            for (int64_t i = 0; i < n; ++i) cnt[(size_t)(col_of(w[i]) / kFragBlock - b_lo)]++;
            for (int64_t b = 0; b < span_blocks; ++b)
                if (cnt[(size_t)b]) ranked.emplace_back(cnt[(size_t)b], (int32_t)(b_lo + b));
        } else {                                // scattered: hash set, gives up early
            size_t cap = 1024;
            while (cap < (size_t)(4 * most)) cap <<= 1;
            if (table.size() != cap) { table.assign(cap, -1); tcnt.assign(cap, 0); }
            used.clear();
            bool over = false;
            for (int64_t i = 0; i < n && !over; ++i) {
                const int32_t b = col_of(w[i]) / kFragBlock;
                size_t h = ((uint32_t)b * 2654435761u) & (cap - 1);
                while (table[h] != -1 && table[h] != b) h = (h + 1) & (cap - 1);
                if (table[h] == -1) {
                    table[h] = b; used.push_back((int32_t)h);
                    if ((int64_t)used.size() > most) over = true;
                }
                tcnt[h]++;
            }
            if (!over) for (int32_t h : used) ranked.emplace_back(tcnt[(size_t)h], table[(size_t)h]);
            for (int32_t h : used) { table[(size_t)h] = -1; tcnt[(size_t)h] = 0; }
            if (over) return -1;
        }
        if ((int64_t)ranked.size() > most) return -1;
        if ((int64_t)ranked.size() > limit) {   // keep the most used blocks (ties: lower block first)
            auto better = [](const std::pair<uint32_t, int32_t>& a, const std::pair<uint32_t, int32_t>& b) {
                return a.first != b.first ? a.first > b.first : a.second < b.second;
            };
            std::nth_element(ranked.begin(), ranked.begin() + limit, ranked.end(), better);
            for (size_t k = (size_t)limit; k < ranked.size(); ++k) global_elems += ranked[k].first;
            ranked.resize((size_t)limit);
        }
        // staging a block costs one L2 request, like gathering one element, plus its place in the window: when
        // blocks used once or twice make up a good part of the window (the stray couplings of an otherwise banded
        // group) they stay out; a few of them are not worth sending the group's slices down the two-way gather
        size_t rare = 0;
        for (const auto& r : ranked) rare += r.first < 3;
        if (rare * 4 > ranked.size()) {
            size_t keep = 0;
            for (size_t k = 0; k < ranked.size(); ++k) {
                if (ranked[k].first >= 3) ranked[keep++] = ranked[k];
                else global_elems += ranked[k].first;
            }
            ranked.resize(keep);
        }
        if (global_elems * 2 > n) return -1;              // the window would serve less than half of the gathers
        for (const auto& r : ranked) blocks.push_back(r.second);
        std::sort(blocks.begin(), blocks.end());
        for (size_t k = 0; k < blocks.size(); ++k) runs += k == 0 || blocks[k] != blocks[k - 1] + 1;
        return (int64_t)blocks.size();
    }
};

}  // namespace

int ytile_floats_for(const SliceStream& st) {
    int max_rows = 1;
    for (size_t sl = 0; sl < st.hdr.size(); ++sl)
        max_rows = std::max(max_rows, (sl + 1 < st.hdr.size() ? st.hdr[sl + 1].row_base : st.rows) - st.hdr[sl].row_base);
    return std::min(kSliceElems, (max_rows + 63) & ~63);
}

LaunchPlan make_plan(SliceStream& st, int n_cus, int only_cfg) {
    const int64_t n = st.n_slices;
    LaunchPlan best;
    best.ytile_floats = ytile_floats_for(st);
    const int ytile = best.ytile_floats;
    const Cfg cfgs[] = {
        // (caps are what is left of the CU's 160 KiB after the row-total tiles of the resident workgroups)
        {256, 8, 6 * 1024, 4},             // small windows: many small workgroups
        {512, 16, 12 * 1024, 2},
        {512, 0, 12 * 1024, 2},            // resident grid: 2 workgroups per CU, window staged once per workgroup
        {512, 16, kMaxLdsFloats, 1},       // mid-size matrices with a large window: 8 wavefronts per CU
        {1024, 64, kMaxLdsFloats, 1},
        {1024, 0, kMaxLdsFloats, 1},       // resident grid: 1 workgroup (16 wavefronts) per CU
    };
    const char* force = std::getenv("HISPMV_PLAN");          // experiments: "global" or an index into cfgs
    const int only = (force && *force >= '0' && *force <= '9') ? std::atoi(force) : -1;
    double best_cost = 1e300;
    bool have = false;
    Cfg chosen{256, 8, 0, 4};
    int64_t chosen_G = 8;
    double chosen_avg_window = 0, small_cost = 1e300;
    bool have_small = false;
    Cfg small{512, 16, 0, 2};
    int64_t small_G = 16;
    int cfg_index = -1;
    for (const Cfg& c0 : cfgs) {
        ++cfg_index;
        if (force && (only < 0 || only != cfg_index)) continue;
        if (only_cfg >= 0 && cfg_index != only_cfg) continue;
        // tiny matrices (< 4 MB of stream) are one latency chain long: staging a window adds two dependent
        // loads (fragment table, x) in front of it and buys nothing -- x is gathered through L2
        if (n < 512 && !force) break;
        Cfg c = c0;
        c.cap = std::min(c.cap, ((160 * 1024 - 4096) / c.per_cu - ytile * (c.threads / 64) * 4) / 4);   // 4 KiB: look-back mailbox, static LDS
        // (a window never grows into the index range of the wavefronts' stray areas: hispmv_plan.h, stray slots)
        c.cap = std::min(c.cap, kCompactMaxIndex - (c.threads / 64) * kStraySlots);
        c.cap &= ~(kFragBlock - 1);
        if (c.cap < 256) continue;
        int64_t G = c.slices;
        if (c.threads == 1024 && G != 0 && n / G < 512) continue;   // big workgroups only when there are plenty
        if (G == 0) {
            G = (n + (int64_t)n_cus * c.per_cu - 1) / ((int64_t)n_cus * c.per_cu);
            // experiment (r4_group_len.sh): HISPMV_PLAN_RESIDENT_DIV=d[,g] -- resident groups d times longer (a d-th of the CUs when the
            // matrix runs alone) for plans whose groups would be shorter than g slices: fewer window-staging phases in a batch call
            if (const char* e = std::getenv("HISPMV_PLAN_RESIDENT_DIV")) {
                int d = 1, below = 1 << 30;
                std::sscanf(e, "%d,%d", &d, &below);
                if (d > 1 && G < below) G = (n + (int64_t)(n_cus / d) * c.per_cu - 1) / ((int64_t)(n_cus / d) * c.per_cu);
            }
            static const int min_resident = std::getenv("HISPMV_PLAN_MIN_RESIDENT") ? std::atoi(std::getenv("HISPMV_PLAN_MIN_RESIDENT")) : 0;
            if (G < (min_resident > 0 ? min_resident : c.threads == 1024 ? 16 : 24)) continue;   // too little work to be worth a resident grid
        } else {
            if (n / G < 1024 && G > 4) G /= 2;                      // small matrices: more, smaller workgroups
            if (n / G < 512 && G > 4) G /= 2;
        }
        const int64_t ng = (n + G - 1) / G;
        const int64_t limit = c.cap / kFragBlock;
        int64_t staged_blocks = 0, global_elems = 0, runs = 0, two_way_elems = 0;
#pragma omp parallel num_threads(host_threads()) reduction(+ : staged_blocks, global_elems, runs, two_way_elems)
        {
            BlockSet bs;
#pragma omp for schedule(dynamic, 16)
            for (int64_t g = 0; g < ng; ++g) {
                const int64_t s0 = g * G, s1 = std::min<int64_t>(n, (g + 1) * G);
                const int64_t k = bs.collect(st, s0, s1, limit);
                if (k < 0) global_elems += (s1 - s0) * kSliceElems;
                else {
                    staged_blocks += k; global_elems += bs.global_elems; runs += bs.runs;
                    if (k > 0 && bs.global_elems > 0) two_way_elems += (s1 - s0) * kSliceElems;
                }
            }
        }
        // cost, with "every element gathers through L2" = 1: L2 requests (one per staged 64-byte block, one per
        // element outside the window) per element, x bytes staged into LDS relative to the stream bytes (L2 -> LDS
        // is ~5x cheaper per byte than the HBM stream), and a penalty for fewer than 16 resident wavefronts per CU
        const int waves = c.per_cu * c.threads / 64;
        // a workgroup stages its window BEFORE it works on its slices: with one or two workgroups per CU nothing hides
        // that phase (~3 us against ~0.4 us per slice; one L2-gathered element costs about 3.3 streamed ones): a window
        // next to a short group does not pay -- R-MAT scale 20 with 16 slices per workgroup, 124 KiB staged per 128 KiB
        // of stream and half of the gathers still through L2, ran 343 us against 180 us gathering everything through L2
        const double staged_ratio = (double)staged_blocks * kFragBlock * 4.0 / ((double)n * kSliceElems * 8.0);
        // Fragments are staged one after the other by a wavefront, two dependent loads each (table entry, x): a window
        // made of hundreds of short runs costs ~0.4 us per fragment and wavefront (ford2 as an unstructured band, 136
        // fragments per 4-slice group of a 256-thread workgroup: 28.8 us against ~10 gathering through L2).  And a group
        // with elements outside its window takes the two-way gather (window read + predicated L2 gather) for all of them.
        const double frag_chain = 0.3 * ((double)runs / (double)ng) / (double)(c.threads / 64);
        const double unhidden = ((c.per_cu <= 2 ? 2.25 : 0.5) + frag_chain) / (double)G;
        const double cost = ((double)staged_blocks + (double)global_elems + 0.15 * (double)two_way_elems) / ((double)n * kSliceElems) +
                            0.2 * staged_ratio + unhidden +
                            0.3 * std::max(0, 16 - waves) / 16.0;
        if (cost < best_cost && cost < 0.8) { best_cost = cost; have = true; chosen = c; chosen_G = G; chosen_avg_window = (double)staged_blocks * kFragBlock / (double)ng; }
        // (experiment, HISPMV_PLAN_CORESIDE_KIB: the best of the two-per-CU 512-thread plans, kept aside)
        if ((cfg_index == 1 || cfg_index == 2) && cost < small_cost && cost < 0.8) { small_cost = cost; have_small = true; small = c; small_G = G; }
    }
    // Experiment (selective co-residency): a matrix whose natural plan is a 1024-thread workgroup with a SMALL
    // window (<= HISPMV_PLAN_CORESIDE_KIB on average) takes the best 512-thread / two-per-CU plan instead (<= 48 KiB of window +
    // 32 KiB of row tiles), so that its workgroups fit on a CU NEXT TO an 8-wavefront tile of a paired tile stream (78 KiB);
    // matrices with large windows (PFlow_742, mouse_gene) keep their plans.  Off by default.
    static const int coreside_kib = std::getenv("HISPMV_PLAN_CORESIDE_KIB") ? std::atoi(std::getenv("HISPMV_PLAN_CORESIDE_KIB")) : 0;
    if (coreside_kib > 0 && have && have_small && chosen.threads == 1024 && chosen_avg_window * 4.0 <= coreside_kib * 1024.0) { chosen = small; chosen_G = small_G; }
    if (!have) {   // scattered columns: plain L2 gathers, small workgroups
        best.block_threads = 256; best.group_slices = (n / 8 < 1024) ? 4 : 8; best.lds_floats = 0; best.per_cu = 4;
        const int64_t ng = (n + best.group_slices - 1) / best.group_slices;
        best.groups.assign((size_t)std::max<int64_t>(ng, 1), GroupDesc{0, 0, 0, 0});
        return best;
    }

    // Build the chosen plan: fragment lists, and the element words of staged groups rewritten to LDS indices.
    const int64_t G = chosen_G, ng = (n + G - 1) / G, limit = chosen.cap / kFragBlock;
    best.block_threads = chosen.threads; best.group_slices = (int)G; best.per_cu = chosen.per_cu;
    best.groups.assign((size_t)ng, GroupDesc{0, 0, 0, 0});
    best.slice_spills.assign((size_t)n, 0);
    std::vector<std::vector<Frag>> gfrags((size_t)ng);
#pragma omp parallel num_threads(host_threads())
    {
        BlockSet bs;
        std::vector<int32_t> rank_of;    // dense path: block - b_lo -> rank
#pragma omp for schedule(dynamic, 16)
        for (int64_t g = 0; g < ng; ++g) {
            const int64_t s0 = g * G, s1 = std::min<int64_t>(n, (g + 1) * G);
            const int64_t k = bs.collect(st, s0, s1, limit);
            if (k <= 0) continue;
            const std::vector<int32_t>& blocks = bs.blocks;
            // fragments = runs of consecutive blocks, split at kFragMaxLen
            std::vector<Frag>& fr = gfrags[(size_t)g];
            int32_t off = 0;
            for (size_t i = 0; i < blocks.size();) {
                size_t j = i + 1;
                while (j < blocks.size() && blocks[j] == blocks[j - 1] + 1 && (int)(j - i) * kFragBlock < kFragMaxLen) ++j;
                fr.push_back(Frag{blocks[i] * kFragBlock, (int32_t)(j - i) * kFragBlock, off, 0});
                off += (int32_t)(j - i) * kFragBlock;
                i = j;
            }
            best.groups[(size_t)g].lds_floats = off;
            best.groups[(size_t)g].n_global = (int32_t)std::min<int64_t>(bs.global_elems, INT32_MAX);
            // rewrite the column field of the group's elements: column -> index in the staged window, or the
            // column itself with kGlobalColBit for the elements whose block is not staged
            uint64_t* w = st.words.data() + s0 * kSliceElems;
            const int64_t ne = (s1 - s0) * kSliceElems;
            for (int64_t i = 0; i < ne; ++i) {
                const int32_t col = col_of(w[i]);
                const int32_t b = col / kFragBlock;
                const auto it = std::lower_bound(blocks.begin(), blocks.end(), b);
                const bool inside = it != blocks.end() && *it == b;
                const uint32_t idx = inside ? (uint32_t)(it - blocks.begin()) * kFragBlock + (uint32_t)(col % kFragBlock)
                                            : (kGlobalColBit | (uint32_t)col);
                w[i] = (w[i] & 0x80000000ffffffffull) | ((uint64_t)idx << 32);
                if (!inside) { uint16_t& q = best.slice_spills[(size_t)(s0 + i / kSliceElems)]; if (q < 0xffff) ++q; }
            }
        }
    }
    int max_lds = 0;
    for (int64_t g = 0; g < ng; ++g) {
        GroupDesc& d = best.groups[(size_t)g];
        d.frag_begin = (int32_t)best.frags.size();
        d.frag_count = (int32_t)gfrags[(size_t)g].size();
        best.frags.insert(best.frags.end(), gfrags[(size_t)g].begin(), gfrags[(size_t)g].end());
        max_lds = std::max(max_lds, d.lds_floats);
        best.staged_floats += d.lds_floats;
        best.global_elems += d.n_global;
    }
    best.lds_floats = (max_lds + 3) & ~3;
    if (best.lds_floats == 0) best.lds_floats = kFragBlock;   // a plan with a window never reports 0
    return best;
}

namespace {
// stray slots: plans whose wavefronts walk several slices (a one-round plan may run the look-back variant, whose walk is not
// rotated: the packer could not know a slice's position), window + stray areas within what a compact meta can index and within
// the LDS share the plan was sized for (make_plan: 160 KiB less 4 KiB, per resident workgroup)
bool stray_slots_possible(const LaunchPlan& plan) {
    const char* env = std::getenv("HISPMV_STRAY_SLOTS");          // (read per call: tests switch it inside one process)
    const bool stray_env = !(env && std::atoi(env) == 0);
    const int n_waves = plan.block_threads / 64;
    return stray_env && !plan.slice_spills.empty() && plan.group_slices > n_waves && plan.lds_floats > 0 &&
           plan.lds_floats + n_waves * kStraySlots <= kCompactMaxIndex &&
           ((int64_t)plan.lds_floats + n_waves * kStraySlots + (int64_t)plan.ytile_floats * n_waves) * 4 <= (160 * 1024 - 4096) / std::max(1, plan.per_cu);
}
// a group with elements outside its window keeps 6-byte elements when every one of its slices has at most kStraySlots of them
bool stray_group_ok(const LaunchPlan& plan, int64_t g, int64_t n) {
    const GroupDesc& gd = plan.groups[(size_t)g];
    if (gd.frag_count <= 0 || gd.n_global <= 0) return false;
    const int64_t G = plan.group_slices, s0 = g * G, s1 = std::min(n, s0 + G);
    for (int64_t sl = s0; sl < s1; ++sl) if (plan.slice_spills[(size_t)sl] > kStraySlots) return false;
    return true;
}
}  // namespace

double stray_slot_coverage(const SliceStream& st, const LaunchPlan& plan) {
    if (!stray_slots_possible(plan)) return 0.0;
    int64_t all = 0, covered = 0;
    for (int64_t g = 0; g < (int64_t)plan.groups.size(); ++g) {
        const int64_t ngl = plan.groups[(size_t)g].frag_count > 0 ? plan.groups[(size_t)g].n_global : 0;
        all += ngl;
        if (ngl > 0 && stray_group_ok(plan, g, st.n_slices)) covered += ngl;
    }
    return all > 0 ? (double)covered / (double)all : 0.0;
}

DeviceStream pack_device_stream(const SliceStream& st, const LaunchPlan& plan, bool materialize) {
    DeviceStream d;
    const int64_t n = st.n_slices, G = plan.group_slices;
    const int64_t ng = std::max<int64_t>((n + G - 1) / G, 1);
    const int n_waves = plan.block_threads / 64;
    d.groups.assign((size_t)ng * 4, 0);
    std::vector<int64_t> off((size_t)ng + 1, 0);
    std::vector<uint8_t> compact((size_t)ng, 0);       // 1 compact, 3 compact with stray slots
    const bool stray_possible = stray_slots_possible(plan);
    bool any_stray = false;
    for (int64_t g = 0; g < ng; ++g) {
        const GroupDesc gd = (size_t)g < plan.groups.size() ? plan.groups[(size_t)g] : GroupDesc{0, 0, 0, 0};
        const int64_t s0 = g * G, s1 = std::min(n, s0 + G);
        compact[(size_t)g] = gd.frag_count > 0 && gd.n_global == 0 && gd.lds_floats <= kCompactMaxIndex;
        if (!compact[(size_t)g] && stray_possible && (size_t)g < plan.groups.size() && stray_group_ok(plan, g, n)) { compact[(size_t)g] = 3; any_stray = true; }
        off[(size_t)g + 1] = off[(size_t)g] + std::max<int64_t>(s1 - s0, 0) * (compact[(size_t)g] ? kCompactSliceBytes : kWideSliceBytes);
        d.groups[(size_t)g * 4 + 0] = gd.frag_begin; d.groups[(size_t)g * 4 + 1] = gd.frag_count;
        d.groups[(size_t)g * 4 + 2] = (int32_t)(off[(size_t)g] / kSliceUnit); d.groups[(size_t)g * 4 + 3] = compact[(size_t)g];
        if (compact[(size_t)g]) d.compact_slices += std::max<int64_t>(s1 - s0, 0);
        if (compact[(size_t)g] == 3) d.stray_slices += std::max<int64_t>(s1 - s0, 0);
    }
    if (off[(size_t)ng] / kSliceUnit > INT32_MAX) throw std::length_error("stream larger than 4 TiB");
    d.n_bytes = off[(size_t)ng];
    d.any_stray = any_stray;
    if (any_stray) d.stray_floats = n_waves * kStraySlots;
    if (!materialize) return d;
    d.bytes.resize((size_t)off[(size_t)ng]);
    if (any_stray) d.stray_cols.assign((size_t)n * kStraySlots, 0xffffffffu);
#pragma omp parallel for num_threads(host_threads()) schedule(dynamic, 4)
    for (int64_t g = 0; g < ng; ++g) {
        const int64_t s0 = g * G, s1 = std::min(n, s0 + G);
        const int64_t n_here = s1 - s0;
        const int64_t rot = n_here > 0 ? (int64_t)(((unsigned long long)g * 29ull) % (unsigned long long)n_here) : 0;   // the kernel's walk (slices_group)
        for (int64_t sl = s0; sl < s1; ++sl) {
            const uint64_t* w = st.words.data() + sl * kSliceElems;
            uint8_t* base = d.bytes.data() + off[(size_t)g] + (sl - s0) * (compact[(size_t)g] ? kCompactSliceBytes : kWideSliceBytes);
            uint32_t* vals = (uint32_t*)base;
            for (int i = 0; i < kSliceElems; ++i) vals[i] = (uint32_t)w[i];
            if (compact[(size_t)g]) {
                uint16_t* meta = (uint16_t*)(base + kSliceElems * 4);
                // position of the slice in its workgroup's walk -> the wavefront that takes it -> that wavefront's stray area
                const int64_t pos = ((sl - s0) - rot + n_here) % n_here;
                const uint32_t area = (uint32_t)plan.lds_floats + (uint32_t)(pos % n_waves) * kStraySlots;
                uint32_t k = 0;
                for (int i = 0; i < kSliceElems; ++i) {
                    const uint32_t m = (uint32_t)(w[i] >> 32);
                    uint32_t idx = m & 0x7fffu;
                    if (m & kGlobalColBit) {                     // a stray: its x value waits in the wavefront's stray area
                        d.stray_cols[(size_t)sl * kStraySlots + k] = m & ~(kRowEndBit | kGlobalColBit);
                        idx = area + k++;
                    }
                    meta[i] = (uint16_t)(idx | ((m & kRowEndBit) ? kCompactEndBit : 0u));
                }
            } else {
                uint32_t* meta = (uint32_t*)(base + kSliceElems * 4);
                for (int i = 0; i < kSliceElems; ++i) meta[i] = (uint32_t)(w[i] >> 32);
            }
        }
    }
    return d;
}

// The planned words of a part back to columns (make_plan rewrote the column field of staged groups to window indices).
WordVec unplanned_words(const SliceStream& st, const LaunchPlan& plan) {
    // (the copy and the rewrite in ONE parallel pass over an uninitialised buffer: `WordVec w = st.words` walked 130 MB on one core for
    // a matrix of TSOPF's size, twice with the stream copy of add_batch_layout -- 70 of the 185 ms a second layout cost)
    WordVec w(st.words.size());
    const int64_t G = plan.group_slices, n = st.n_slices;
    const int64_t n_groups = plan.groups.empty() ? (n + G - 1) / std::max<int64_t>(1, G) : (int64_t)plan.groups.size();
#pragma omp parallel for num_threads(host_threads()) schedule(dynamic, 16)
    for (int64_t g = 0; g < n_groups; ++g) {
        const int64_t s0 = g * G, s1 = std::min<int64_t>(n, s0 + G);
        const bool staged = g < (int64_t)plan.groups.size() && plan.groups[(size_t)g].frag_count > 0;
        if (!staged) {                                                  // not staged: the words still hold columns
            for (int64_t i = s0 * kSliceElems; i < s1 * kSliceElems; ++i) w[(size_t)i] = st.words[(size_t)i];
            continue;
        }
        const GroupDesc& gd = plan.groups[(size_t)g];
        const Frag* f0 = plan.frags.data() + gd.frag_begin;
        const Frag* f1 = f0 + gd.frag_count;
        for (int64_t i = s0 * kSliceElems; i < s1 * kSliceElems; ++i) {
            w[(size_t)i] = st.words[(size_t)i];
            const uint32_t m = (uint32_t)(w[(size_t)i] >> 32);
            uint32_t col;
            if (m & kGlobalColBit) col = m & ~(kRowEndBit | kGlobalColBit);
            else {
                const int32_t idx = (int32_t)(m & ~kRowEndBit);
                const Frag* it = std::upper_bound(f0, f1, idx, [](int32_t v, const Frag& f) { return v < f.lds_off; }) - 1;
                col = (uint32_t)(it->col_start + (idx - it->lds_off));
            }
            w[(size_t)i] = (w[(size_t)i] & 0x80000000ffffffffull) | ((uint64_t)col << 32);
        }
    }
    return w;
}


}  // namespace hispmv
