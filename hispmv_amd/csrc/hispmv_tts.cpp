// hispmv_tts.cpp -- packer of the transposed tile stream (hispmv_tts.h).  Host-only, OpenMP over the row tiles.
#include "hispmv_tts.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace hispmv {

namespace {

struct TileOut {
    std::vector<TtsBlock> blocks;            // slice_begin / chunk_begin relative to the tile
    std::vector<uint8_t> words;
    std::vector<int32_t> col_base;
    std::vector<uint16_t> flags, flags_hi;
    std::vector<int32_t> chunk_info;
    int64_t fillers = 0, pads = 0;
    double lines = 0; int64_t gathers = 0;
    int max_slots = 0;
};

struct Elem { int32_t col; int32_t row; float val; };     // row: tile-local

// One block: elements [e0, e1) of the tile's column-sorted list.
void pack_block(const std::vector<Elem>& el, size_t e0, size_t e1, int n_rows, TileOut& out, std::vector<int32_t>& cnt,
                std::vector<int32_t>& start, std::vector<uint32_t>& slot_of, const TtsGeometry& geo) {
    // row-major order of the block: rows ascending, inside a row the elements in column order (the order of `el`), a row
    // without elements in this block gets one filler slot -- or, with gap-coded row ends, nothing at all unless it has to break
    // a run of absent rows (a row end's code reaches at most 3 rows back)
    std::fill(cnt.begin(), cnt.begin() + n_rows, 0);
    for (size_t i = e0; i < e1; ++i) cnt[(size_t)el[i].row]++;
    int32_t pos = 0, fillers = 0;
    std::vector<uint8_t> owns((size_t)n_rows, 1), code((size_t)n_rows, 1);      // gap_rows: which rows own a slot, and their codes
    if (geo.gap_rows) {
        int prev = -1;
        for (int r = 0; r < n_rows; ++r) {
            const bool present = cnt[(size_t)r] > 0;
            const bool breaker = !present && r - prev == 3;         // third absent row in a run: a filler keeps the next code <= 3
            owns[(size_t)r] = present || breaker;
            if (owns[(size_t)r]) { code[(size_t)r] = (uint8_t)(r - prev); prev = r; }
        }
    }
    for (int r = 0; r < n_rows; ++r) {
        start[(size_t)r] = pos;
        if (!owns[(size_t)r]) continue;
        if (cnt[(size_t)r] == 0) { ++fillers; pos += 1; } else pos += cnt[(size_t)r];
    }
    const int32_t n_slots = pos;
    if (n_slots > geo.max_slots) throw std::logic_error("internal: TTS block exceeds its slot budget");
    const int32_t n_chunks = (n_slots + kTtsChunk - 1) / kTtsChunk;
    TtsBlock b{};
    b.slice_begin = (int32_t)out.col_base.size();
    b.chunk_begin = (int32_t)(out.chunk_info.size() / 2);
    b.n_chunks = n_chunks; b.n_slots = n_slots;
    // row-end flags and chunk table
    const size_t f0 = out.flags.size();
    out.flags.resize(f0 + (size_t)n_chunks * 64, 0);
    if (geo.gap_rows) out.flags_hi.resize(f0 + (size_t)n_chunks * 64, 0);
    std::vector<int32_t> ends_before((size_t)n_chunks + 1, 0);
    std::vector<int32_t> first_row((size_t)n_chunks, -1), last_before((size_t)n_chunks + 1, -1);      // gap_rows: first row ending in / last row ending before a chunk
    auto set_end = [&](int32_t slot, int r) {
        const int32_t c = slot / kTtsChunk, s = slot % kTtsChunk;
        const int j = s / 256, l = (s % 256) / 4, k = s % 4;
        const int cd = geo.gap_rows ? code[(size_t)r] : 1;
        if (cd & 1) out.flags[f0 + (size_t)c * 64 + (size_t)l] |= (uint16_t)(1u << (4 * j + k));
        if (cd & 2) out.flags_hi[f0 + (size_t)c * 64 + (size_t)l] |= (uint16_t)(1u << (4 * j + k));
        ends_before[(size_t)c + 1]++;
        if (first_row[(size_t)c] < 0) first_row[(size_t)c] = r;
        last_before[(size_t)c + 1] = r;
    };
    for (int r = 0; r < n_rows; ++r) if (owns[(size_t)r]) set_end(start[(size_t)r] + std::max(cnt[(size_t)r], 1) - 1, r);
    for (int c = 0; c < n_chunks; ++c) ends_before[(size_t)c + 1] += ends_before[(size_t)c];
    for (int c = 0; c < n_chunks; ++c) if (last_before[(size_t)c + 1] < 0) last_before[(size_t)c + 1] = last_before[(size_t)c];
    for (int c = 0; c < n_chunks; ++c) {
        if (!geo.gap_rows) {
            const int32_t row_base = ends_before[(size_t)c];        // first row that ends in this chunk (when one does)
            int32_t chain = 0;
            if (ends_before[(size_t)c + 1] > row_base && row_base < n_rows && start[(size_t)row_base] < c * kTtsChunk)
                chain = c - start[(size_t)row_base] / kTtsChunk;
            out.chunk_info.push_back(row_base);
            out.chunk_info.push_back(chain);
        } else {
            const int32_t fr = first_row[(size_t)c];               // first row that ends in this chunk (-1: none)
            int32_t chain = 0;
            if (fr >= 0 && start[(size_t)fr] < c * kTtsChunk) chain = c - start[(size_t)fr] / kTtsChunk;
            out.chunk_info.push_back(last_before[(size_t)c]);      // the last slot-owning row before the chunk: row ends add their codes to it
            out.chunk_info.push_back(chain | ((fr >= 0 ? (int32_t)code[(size_t)fr] : 0) << 16));
        }
    }
    // slots of the elements (column order -> row-major position); cnt[] becomes the running fill of each row
    std::fill(cnt.begin(), cnt.begin() + n_rows, 0);
    slot_of.resize(e1 - e0);
    for (size_t i = e0; i < e1; ++i) { const int r = el[i].row; slot_of[i - e0] = (uint32_t)(start[(size_t)r] + cnt[(size_t)r]++); }
    // phase A words: the elements in column order, then the fillers (value 0, the block's first column), in slices of
    // 1024 words whose columns lie within 65536 of the slice's base
    struct W { int32_t col; uint32_t slot; float val; };
    std::vector<W> ws;
    ws.reserve((e1 - e0) + (size_t)fillers);
    for (size_t i = e0; i < e1; ++i) ws.push_back(W{el[i].col, slot_of[i - e0], el[i].val});
    const int32_t first_col = e1 > e0 ? el[e0].col : 0;
    // (cnt[] now holds each row's element count again)  fillers follow the elements: column -1 = "the base of whatever
    // slice it lands in" (offset 0: a valid column, the gather is a broadcast)
    // (zero_fill geometry: nothing is streamed for an absent row -- its slot of the zero-filled staging reads 0.0)
    if (!geo.zero_fill)
        for (int r = 0; r < n_rows; ++r) if (cnt[(size_t)r] == 0 && owns[(size_t)r]) ws.push_back(W{-1, (uint32_t)start[(size_t)r], 0.0f});
    size_t i = 0;
    while (i < ws.size()) {
        const int32_t base = (ws[i].col >= 0 ? ws[i].col : first_col) & ~31;          // 128-byte aligned
        size_t j = i;
        while (j < ws.size() && j - i < (size_t)kTtsChunk && (ws[j].col < 0 || ws[j].col - base < 65536)) ++j;
        out.col_base.push_back(base);
        const size_t w0 = out.words.size();
        out.words.resize(w0 + (size_t)kTtsChunk * 8);
        uint32_t* vals = (uint32_t*)(out.words.data() + w0);
        uint32_t* meta = vals + kTtsChunk;
        // Word order inside a slice: the kernel reads 4 consecutive words per lane and step (one dwordx4 of values, one
        // of metas) and gathers with one instruction per word position k -- so that the 64 lanes of a gather read 64
        // CONSECUTIVE elements of the column order, sorted element 256 j + 64 k + l sits at word 256 j + 4 l + k.
        // (Phase A has no order of its own: every word carries its slot.)
        for (size_t q = 0; q < (size_t)kTtsChunk; ++q) {
            const size_t step = q / 256, k = (q % 256) / 64, l = q % 64, at = 256 * step + 4 * l + k;
            if (i + q < j) {
                std::memcpy(&vals[at], &ws[i + q].val, 4);
                meta[at] = ((uint32_t)(ws[i + q].col >= 0 ? ws[i + q].col - base : 0) << 16) | ws[i + q].slot;
            } else { vals[at] = 0; meta[at] = (uint32_t)geo.max_slots; ++out.pads; }
        }
        // diagnostic: distinct 128-byte lines per 64-lane gather
        for (size_t s0 = i; s0 < j; s0 += 64) {
            int last = -1, n_lines = 0;
            for (size_t q = s0; q < std::min(j, s0 + 64); ++q) { const int ln = (ws[q].col >= 0 ? ws[q].col : base) >> 5; if (ln != last) { ++n_lines; last = ln; } }
            out.lines += n_lines; out.gathers++;
        }
        i = j;
    }
    b.n_slices = (int32_t)out.col_base.size() - b.slice_begin;
    if (b.n_slices > kTtsMaxBlockSlices) throw std::logic_error("internal: TTS block exceeds its slice budget");
    out.blocks.push_back(b);
    out.fillers += fillers;
    out.max_slots = std::max(out.max_slots, n_slots);
}

}  // namespace

TtsStream build_tts(const Csr& m, int64_t target_tile_elems, TtsGeometry geo) {
    TtsStream S;
    S.geometry = geo;
    S.rows = m.rows; S.cols = m.cols; S.nnz = m.nnz();
    const int32_t R = m.rows;
    // row tiles: about `target` elements each (every row counts at least one), at most kTtsMaxRows rows.  More rows per
    // tile = more elements per cache line of x in a gather (the tile's elements per column); the target keeps >= ~256
    // tiles on large matrices so that every CU has one.
    const int64_t total = S.nnz + R;
    // (the floor keeps enough elements per tile for the lanes of a gather to share lines of x; when x is at most 256 KiB --
    // a wide, short layer of apps/model_test.py: 1024 x 8192 -- its lines stay in L1 / L2 whatever the tile, and a tile per
    // CU matters more: 6 K elements instead of 24 K, 256 tiles instead of 87 for that layer)
    int64_t floor_elems = m.cols <= 64 * 1024 ? 6 * kTtsChunk : 24 * kTtsChunk;
    if (const char* env = std::getenv("HISPMV_TTS_FLOOR")) floor_elems = std::max<int64_t>(1024, std::atoll(env));     // experiments
    int64_t target = target_tile_elems > 0 ? target_tile_elems : std::max<int64_t>(total / geo.tiles_wanted, std::min<int64_t>(floor_elems, geo.max_slots));
    struct Range { int32_t r0, r1; int64_t k0, k1; int32_t carry; };     // k0 >= 0: the piece [k0, k1) of row r0; carry >= 0: carry tile
    std::vector<Range> ranges;
    // a row longer than a tile and a quarter is cut into pieces of about one tile (at most 33: the fix-up kernel for short
    // chains adds them in order, one thread per row)
    const int64_t split_above = target + target / 4;
    constexpr int kMaxPieces = 33;
    for (int32_t r = 0; r < R;) {
        const int64_t len0 = m.row_ptr[(size_t)r + 1] - m.row_ptr[r];
        if (len0 > split_above) {
            const int pieces = (int)std::min<int64_t>(kMaxPieces, (len0 + target - 1) / target);
            const int64_t per = (len0 + pieces - 1) / pieces;
            S.fix.insert(S.fix.end(), {r, S.n_carry, pieces - 1, 0});
            for (int q = 0; q < pieces; ++q) {
                const int64_t k0 = m.row_ptr[r] + q * per, k1 = std::min<int64_t>(m.row_ptr[(size_t)r + 1], k0 + per);
                ranges.push_back(Range{r, r + 1, k0, k1, q + 1 < pieces ? S.n_carry++ : -1});
            }
            ++r;
            continue;
        }
        int32_t e = r;
        int64_t acc = 0;
        while (e < R && e - r < geo.max_rows) {
            const int64_t raw = m.row_ptr[(size_t)e + 1] - m.row_ptr[e];
            if (raw > split_above) break;                                    // (a long row starts its own tiles)
            const int64_t len = std::max<int64_t>(raw, 1);
            if (e > r && acc + len > target) break;
            acc += len; ++e;
        }
        ranges.push_back(Range{r, e, -1, -1, -1});
        r = e;
    }
    const size_t nt = ranges.size();
    std::vector<TileOut> outs(nt);
    int col_bits = 1;
    while (col_bits < 31 && (1ll << col_bits) < (long long)m.cols) ++col_bits;
#pragma omp parallel num_threads(host_threads())
    {
        std::vector<Elem> el, tmp;
        std::vector<int32_t> cnt((size_t)geo.max_rows), start((size_t)geo.max_rows), seen((size_t)geo.max_rows);
        std::vector<uint32_t> slot_of;
#pragma omp for schedule(dynamic, 1)
        for (long long t = 0; t < (long long)nt; ++t) {
            const Range rg = ranges[(size_t)t];
            const int n_rows = rg.r1 - rg.r0;
            el.clear();
            if (rg.k0 >= 0) {
                for (int64_t k = rg.k0; k < rg.k1; ++k) el.push_back(Elem{m.col[(size_t)k], 0, m.val[(size_t)k]});
            } else {
                for (int32_t r = rg.r0; r < rg.r1; ++r)
                    for (int64_t k = m.row_ptr[r]; k < m.row_ptr[(size_t)r + 1]; ++k) el.push_back(Elem{m.col[(size_t)k], r - rg.r0, m.val[(size_t)k]});
            }
            // stable sort by column (rows stay ascending inside a column): LSD radix sort, 11 bits per pass -- two thirds of the
            // packer's thread time went into std::stable_sort here
            if (el.size() < 2048) {
                std::stable_sort(el.begin(), el.end(), [](const Elem& a, const Elem& b) { return a.col < b.col; });
            } else {
                tmp.resize(el.size());
                Elem* src = el.data(); Elem* dst = tmp.data();
                for (int shift = 0; shift < col_bits; shift += 11) {
                    uint32_t hist[2049] = {0};
                    const size_t n = el.size();
                    for (size_t i = 0; i < n; ++i) hist[(((uint32_t)src[i].col >> shift) & 2047u) + 1]++;
                    for (int d = 0; d < 2048; ++d) hist[d + 1] += hist[d];
                    for (size_t i = 0; i < n; ++i) dst[hist[((uint32_t)src[i].col >> shift) & 2047u]++] = src[i];
                    std::swap(src, dst);
                }
                if (src != el.data()) el.swap(tmp);
            }
            // blocks: greedy over the column-sorted list -- a block closes when its elements plus one filler for every
            // row it has not seen would exceed the slot budget
            TileOut& out = outs[(size_t)t];
            std::fill(seen.begin(), seen.begin() + n_rows, -1);
            size_t b0 = 0;
            int block_id = 0, distinct = 0;
            int sl_count = 0, sl_fill = 0, sl_base = 0;        // slices the block's elements take so far; fill and base of the open one
            for (size_t i = 0; i <= el.size(); ++i) {
                bool close = i == el.size();
                int n_sl = sl_count, n_fill = sl_fill, n_base = sl_base;
                if (!close) {
                    const int r = el[i].row;
                    const int add_distinct = seen[(size_t)r] != block_id ? 1 : 0;
                    // the element opens a new slice when the open one is full or its 16-bit column offset would overflow
                    if (sl_count == 0 || sl_fill == kTtsChunk || el[i].col - sl_base >= 65536) { n_sl = sl_count + 1; n_fill = 1; n_base = el[i].col & ~31; }
                    else n_fill = sl_fill + 1;
                    // (gap-coded row ends: at most every third absent row needs a filler)
                    const int64_t fillers = geo.gap_rows ? (n_rows - distinct - add_distinct + 2) / 3 : n_rows - distinct - add_distinct;
                    const int64_t extra = geo.zero_fill ? 0 : fillers - (kTtsChunk - n_fill);   // filler words beyond the open slice's room
                    const int64_t total_slices = n_sl + (extra > 0 ? (extra + kTtsChunk - 1) / kTtsChunk : 0);
                    if ((int64_t)(i - b0 + 1) + fillers > geo.max_slots || total_slices > kTtsMaxBlockSlices) close = true;
                }
                if (close) {
                    pack_block(el, b0, i, n_rows, out, cnt, start, slot_of, geo);
                    if (i == el.size()) break;
                    b0 = i; ++block_id; distinct = 0;
                    n_sl = 1; n_fill = 1; n_base = el[i].col & ~31;
                }
                sl_count = n_sl; sl_fill = n_fill; sl_base = n_base;
                const int r = el[i].row;
                if (seen[(size_t)r] != block_id) { seen[(size_t)r] = block_id; ++distinct; }
            }
        }
    }
    // concatenate the tiles
    int64_t n_slices = 0, n_chunks = 0;
    std::vector<int64_t> tile_work(nt, 0);
    for (size_t t = 0; t < nt; ++t) {
        TileOut& o = outs[t];
        TtsTile tile{ranges[t].carry >= 0 ? -(ranges[t].carry + 1) : ranges[t].r0, ranges[t].r1 - ranges[t].r0, (int32_t)S.blocks.size(), (int32_t)o.blocks.size()};
        for (TtsBlock b : o.blocks) { b.slice_begin += (int32_t)n_slices; b.chunk_begin += (int32_t)n_chunks; S.blocks.push_back(b); }
        S.tiles.push_back(tile);
        n_slices += (int64_t)o.col_base.size(); n_chunks += (int64_t)o.chunk_info.size() / 2;
        if (n_slices > INT32_MAX / 2 || n_chunks > INT32_MAX / 2) throw std::length_error("TTS stream too large");
        S.n_fillers += o.fillers; S.n_pad_words += o.pads;
        S.max_rows = std::max(S.max_rows, tile.n_rows); S.max_slots = std::max(S.max_slots, o.max_slots);
        int64_t tile_slots = 0;
        for (const TtsBlock& b : o.blocks) tile_slots += b.n_slots;
        S.total_slots += tile_slots; S.max_tile_slots = std::max(S.max_tile_slots, tile_slots);
        tile_work[t] = tile_slots;
    }
    // The tile table is the launch order (workgroup t takes tiles[t]; a CU holds one tile at a time): longest first, so
    // that uneven tiles (row lengths far from uniform: 1.4 tiles per CU for Zipf lengths at soc-Pokec's shape) pack behind
    // each other instead of a long one starting last.  Tiles of equal work keep their row order.
    {
        std::vector<size_t> order(nt);
        for (size_t t = 0; t < nt; ++t) order[t] = t;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return tile_work[a] / 4096 > tile_work[b] / 4096; });
        std::vector<TtsTile> sorted(nt);
        for (size_t t = 0; t < nt; ++t) sorted[t] = S.tiles[order[t]];
        S.tiles.swap(sorted);
    }
    // concatenate the tiles' arrays: offsets first, then the copies in parallel (single-threaded this was three quarters of
    // the packer's time on soc-Pokec's shape: 1.3 of 1.7 s on 8 cores -- 260 MB of words through one core, and the frees)
    std::vector<size_t> w_off(nt + 1, 0), cb_off(nt + 1, 0), fl_off(nt + 1, 0), ci_off(nt + 1, 0);
    for (size_t t = 0; t < nt; ++t) {
        w_off[t + 1] = w_off[t] + outs[t].words.size(); cb_off[t + 1] = cb_off[t] + outs[t].col_base.size();
        fl_off[t + 1] = fl_off[t] + outs[t].flags.size(); ci_off[t + 1] = ci_off[t] + outs[t].chunk_info.size();
    }
    S.words.resize((size_t)n_slices * kTtsChunk * 8);
    S.col_base.resize(cb_off[nt]); S.flags.resize(fl_off[nt]); S.chunk_info.resize(ci_off[nt]);
    if (geo.gap_rows) S.flags_hi.resize(fl_off[nt]);
    double lines = 0; int64_t gathers = 0;
#pragma omp parallel for num_threads(host_threads()) schedule(dynamic, 1) reduction(+ : lines, gathers)
    for (long long tt = 0; tt < (long long)nt; ++tt) {
        const size_t t = (size_t)tt;
        TileOut& o = outs[t];
        if (!o.words.empty()) std::memcpy(S.words.data() + w_off[t], o.words.data(), o.words.size());
        if (!o.col_base.empty()) std::memcpy(S.col_base.data() + cb_off[t], o.col_base.data(), o.col_base.size() * sizeof(int32_t));
        if (!o.flags.empty()) std::memcpy(S.flags.data() + fl_off[t], o.flags.data(), o.flags.size() * sizeof(uint16_t));
        if (!o.flags_hi.empty()) std::memcpy(S.flags_hi.data() + fl_off[t], o.flags_hi.data(), o.flags_hi.size() * sizeof(uint16_t));
        if (!o.chunk_info.empty()) std::memcpy(S.chunk_info.data() + ci_off[t], o.chunk_info.data(), o.chunk_info.size() * sizeof(int32_t));
        lines += o.lines; gathers += o.gathers;
        o = TileOut{};
    }
    S.lines_per_gather = gathers ? lines / (double)gathers : 0.0;
    return S;
}

std::vector<int32_t> tts_column_cuts(const Csr& m, int parts) {
    std::vector<int32_t> cuts;
    if (parts < 2) return cuts;
    const int32_t n_bins = (m.cols + 31) / 32;
    std::vector<int64_t> hist((size_t)n_bins, 0);
    const int64_t nnz = m.nnz();
    for (int64_t k = 0; k < nnz; ++k) hist[(size_t)(m.col[(size_t)k] >> 5)]++;
    int64_t acc = 0;
    int next = 1;
    for (int32_t b = 0; b < n_bins && next < parts; ++b) {
        acc += hist[(size_t)b];
        while (next < parts && acc * parts >= nnz * next) { cuts.push_back((b + 1) * 32); ++next; }
    }
    while ((int)cuts.size() < parts - 1) cuts.push_back(m.cols);
    return cuts;
}

Csr csr_column_range(const Csr& m, int32_t c0, int32_t c1) {
    Csr t;
    t.rows = m.rows; t.cols = m.cols;
    t.row_ptr.assign((size_t)m.rows + 1, 0);
#pragma omp parallel for num_threads(host_threads()) schedule(static)
    for (int32_t i = 0; i < m.rows; ++i) {
        const int32_t* b = m.col.data() + m.row_ptr[i];
        const int32_t* e = m.col.data() + m.row_ptr[(size_t)i + 1];
        t.row_ptr[(size_t)i + 1] = std::lower_bound(b, e, c1) - std::lower_bound(b, e, c0);
    }
    for (int32_t i = 0; i < m.rows; ++i) t.row_ptr[(size_t)i + 1] += t.row_ptr[i];
    t.col.resize((size_t)t.row_ptr[m.rows]); t.val.resize((size_t)t.row_ptr[m.rows]);
#pragma omp parallel for num_threads(host_threads()) schedule(static)
    for (int32_t i = 0; i < m.rows; ++i) {
        const int32_t* b = m.col.data() + m.row_ptr[i];
        const int32_t* e = m.col.data() + m.row_ptr[(size_t)i + 1];
        const int64_t k0 = m.row_ptr[i] + (std::lower_bound(b, e, c0) - b);
        const int64_t n = t.row_ptr[(size_t)i + 1] - t.row_ptr[i];
        std::copy_n(m.col.data() + k0, n, t.col.data() + t.row_ptr[i]);
        std::copy_n(m.val.data() + k0, n, t.val.data() + t.row_ptr[i]);
    }
    return t;
}

}  // namespace hispmv
