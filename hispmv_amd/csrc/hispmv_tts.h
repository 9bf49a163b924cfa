// hispmv_tts.h -- the TRANSPOSED TILE STREAM: second device format, for matrices whose rows are short and whose columns
// are scattered (soc-Pokec, circuit matrices with wide jittered bands), where the slice stream gathers x with one L2
// request per element.
//
// Lineage: the reference tiles the matrix in two dimensions, keeps the row sums of a row tile in per-PE URAM
// accumulators across its column tiles and streams {row, col, value} words in an order chosen by the packer
// (tileAndPad spmv-helper.cpp:242-263, AccumBuffer base_functions.cpp:439-519, encode spmv-helper.h:45-60).  Here:
//   * a ROW TILE (<= kTtsMaxRows consecutive rows) belongs to one workgroup; its row sums live in LDS for the whole tile;
//   * the tile is cut into BLOCKS along the columns; a block holds <= kTtsMaxSlots element slots;
//   * inside a block the elements are streamed SORTED BY COLUMN -- the 64 lanes of a gather read neighbouring columns of
//     x, a few cache lines per wave instruction instead of 64 -- as words {fp32 value, column offset:16 | slot:16};
//     `slot` is the element's position in the block's ROW-MAJOR order;
//   * phase A (column order): gather, multiply, ds_write the product to staging[slot] in LDS -- the transposition;
//     phase B (row order): read the products back in row-major order and reduce them with the slice kernel's lane-local
//     combine + DPP segmented scan, row ends from a bit per slot; row totals are added to the tile's LDS accumulators by
//     the lane that owns the row end (plain read-modify-write: rows are distinct inside a wave, cut rows are finished
//     from a carry slot after a barrier).  No atomics: ds_add_f32 takes 193 cycles per wave instruction on gfx950
//     (tools/lds_atomic_bench.hip), a compare-and-swap loop collapses under contention.  Results are deterministic.
//   * every row of the tile owns >= 1 slot in every block (zero-valued filler), so rows need no index list;
//   * a row longer than a tile and a quarter (Zipf row lengths: 1.6 M entries against 126 K per tile) is cut into <= 33 pieces of
//     consecutive elements, each a tile of its own: the last piece is an ordinary one-row tile (y = alpha*sum + beta*bias),
//     the others are CARRY TILES (row0 < 0) that write their raw sum to carry[-row0 - 1]; the fix-up kernel of the slice
//     stream then adds alpha * (carry_0 + carry_1 + ...) to y[row] (entries {row, first carry, pieces - 1, 0} in `fix`).
// Host-only code (packer); the kernel is in hispmv_kernels.hip.
#pragma once
#include <cstdint>
#include <memory>
#include <utility>
#include <vector>

#include "hispmv_prep.h"

namespace hispmv {

constexpr int kTtsChunk = 1024;          // slots per row-major chunk / words per column-order slice
constexpr int kTtsMaxSlots = 28 * 1024;  // slots of a block (112 KiB of staging)
constexpr int kTtsMaxRows = 8 * 1024;    // rows of a tile (32 KiB of accumulators)
constexpr int kTtsThreads = 1024;        // one 16-wave workgroup per CU (two 8-wave workgroups with half the LDS each: measured slower,
                                         //   the tiles get shorter and every gather touches more lines of x)
constexpr int kTtsMaxBlockSlices = 48;  // column-order slices of a block (the 16-bit column offsets can cut slices short)
// A second geometry (HISPMV_TTS_SMALL=1; not used by default): half the LDS per workgroup, so that TWO 16-wavefront
// workgroups share a CU and one's column-order pass overlaps the other's row-order pass.  Tried for matrices whose
// gathers are cheap anyway (<= 8 lines per gather with the tall tiles; the tall geometry spends half of its wave cycles
// waiting there): PFlow_742 as an unstructured band 81.9 -> 93.4 us, Si41Ge41H72 39.5 -> 48.0 us, boyd2 18.7 -> 14.6 us.
constexpr int kTtsSmallSlots = 13 * 1024;
constexpr int kTtsSmallRows = 4 * 1024;
// A third geometry, the TALL one: 16 K rows and 23 K slots per tile (64 + 92 KiB of LDS), for matrices whose gathers touch
// many lines of x with the 8 K-row tiles (soc-Pokec: 23 lines per gather).  Twice the rows over HALF the columns (the
// matrix is cut into two column parts, each a tile stream of its own: part 0 writes y, part 1 a partial vector that the
// merge launch adds) keeps the elements per tile and the number of tiles, and halves the column range a tile's elements
// spread over: 9 lines per gather.  With that many rows a block of 23 K slots holds few elements per row, and one filler
// WORD per absent row would be half the stream: in this geometry (`zero_fill`) an absent row still owns a slot of the
// block's row-major order, but nothing is streamed for it -- the staging is zero when phase A starts (phase B writes
// zeros back over what it has read), so the slot reads 0.0.
constexpr int kTtsTallSlots = 23 * 1024;
constexpr int kTtsTallRows = 16 * 1024;
// A fourth, the PAIRED geometry: two 8-wavefront workgroups per CU, each with 8 K rows of accumulators and 11 K slots of
// staging (2 x 78 KiB), again over two column parts so that the number of row tiles stays what the rows allow (a CU then
// holds a tile of each part, or two of one).  Measured on soc-Pokec the tile stream is bound by two things that do not
// overlap inside ONE workgroup: the texture addresser takes the lanes of a scattered gather one per cycle (34.3 M cache
// accesses per launch for 32.4 M elements: 53 us of the 103 whatever the lines per gather are), and phase B is VALU work
// behind a barrier.  Two independent workgroups per CU are in different phases most of the time.
constexpr int kTtsPairedSlots = 11 * 1024;
constexpr int kTtsPairedThreads = 512;
struct TtsGeometry {
    int max_slots = kTtsMaxSlots, max_rows = kTtsMaxRows, tiles_wanted = 256;
    bool zero_fill = false;      // rows absent from a block own a slot but no stream word (the kernel keeps the staging zero-filled)
    // GAP-CODED row ends (round 4): rows absent from a block own NOTHING -- no word, no slot.  A row end carries a 2-bit code, the
    // distance (1..3) from the previous slot-owning row of the block to this one (a second plane of flag bits, `flags_hi`; a
    // zero-valued filler slot breaks a run of more than two absent rows), and the kernel takes a row end's accumulator from the
    // running sum of the codes instead of the running count of row ends.  What the tall geometry needed to pay off: its gathers
    // touch 13 lines instead of 23, but with one slot per row and block its row-order pass had 46 % more slots to reduce.
    bool gap_rows = false;
    int threads = kTtsThreads;   // workgroup size
};

struct TtsTile {           // 16 B per workgroup
    int32_t row0;          // first row; < 0: carry tile (one piece of a long row), its sum goes to carry[-row0 - 1]
    int32_t n_rows;
    int32_t block_begin;   // first block of the tile
    int32_t n_blocks;
};
struct TtsBlock {          // 32 B
    int32_t slice_begin;   // first column-order slice (1024 words each: values at words + slice*8192, metas 4096 B behind)
    int32_t n_slices;
    int32_t chunk_begin;   // first row-major chunk (flags: 64 x u16 per chunk; chunk table: {row_base, chain_len})
    int32_t n_chunks;
    int32_t n_slots;       // real slots (elements + fillers); the rest of the last chunk is padding without row ends
    int32_t pad0, pad1, pad2;
};

struct TtsStream {
    int32_t rows = 0, cols = 0;
    int64_t nnz = 0;
    std::vector<TtsTile> tiles;
    std::vector<TtsBlock> blocks;
    std::vector<uint8_t, DefaultInitAllocator<uint8_t>> words;   // per slice: 1024 x fp32 value, then 1024 x u32 meta (col_off << 16 | slot)
    std::vector<int32_t> col_base;       // per slice: column the 16-bit offsets are relative to
    std::vector<uint16_t> flags;         // per chunk: 64 x u16, bit 4j+k of lane l = row end at slot 256j + 4l + k (gap_rows: bit 0 of the code)
    std::vector<uint16_t> flags_hi;      // gap_rows only: bit 1 of the row ends' codes, same layout (code = distance to the previous slot-owning row)
    std::vector<int32_t> chunk_info;     // per chunk: {rows ending before the chunk (tile-local), chain_len: the first row
                                         //   ending in the chunk began this many chunks earlier}; gap_rows: {the last slot-owning row
                                         //   BEFORE the chunk (-1: none), chain_len | code of the chunk's first row end << 16}
    std::vector<int32_t> fix;            // per row cut into pieces: {row, first carry, number of carries, 0} (the slice stream's FixEntry)
    int32_t n_carry = 0;
    int64_t n_fillers = 0, n_pad_words = 0;
    double lines_per_gather = 0;         // distinct 128-byte lines of x per 64-lane gather (diagnostic / format choice)
    int max_rows = 0, max_slots = 0;
    TtsGeometry geometry;                // what it was packed for (padding words of phase A write to slot geometry.max_slots)
    int64_t total_slots = 0, max_tile_slots = 0;   // a tile is one workgroup's work: a tile far above the mean (one very long row) is the critical path
    int64_t bytes() const {
        return (int64_t)words.size() + (int64_t)col_base.size() * 4 + (int64_t)(flags.size() + flags_hi.size()) * 2 + (int64_t)chunk_info.size() * 4 +
               (int64_t)tiles.size() * 16 + (int64_t)blocks.size() * 32 + (int64_t)fix.size() * 4 + (int64_t)n_carry * 4;
    }
};

// Packs a CSR matrix (columns ascending per row).  `target_tile_elems`: elements per row tile (0 = chosen from the matrix
// and the geometry).
TtsStream build_tts(const Csr& m, int64_t target_tile_elems = 0, TtsGeometry geometry = TtsGeometry());

// The column parts of the tall geometry: `parts` - 1 cut columns (ascending, multiples of 32) that split the ELEMENTS of
// the matrix evenly (a histogram over the columns; a part is [cut[p-1], cut[p]) with cut[-1] = 0, cut[parts-1] = cols).
std::vector<int32_t> tts_column_cuts(const Csr& m, int parts);
// The tall geometry for a device with `n_cus` CUs and `parts` column parts (each part gets n_cus / parts tiles).
inline TtsGeometry tts_tall_geometry(int n_cus, int parts) {
    TtsGeometry g;
    g.max_slots = kTtsTallSlots; g.max_rows = kTtsTallRows; g.tiles_wanted = n_cus / parts > 0 ? n_cus / parts : 1; g.zero_fill = true;
    return g;
}
constexpr int kTtsTallParts = 2;
// The tall geometry with gap-coded row ends: the same tiles, no slot for an absent row, no zero-filled staging.
inline TtsGeometry tts_tallgap_geometry(int n_cus, int parts) {
    TtsGeometry g = tts_tall_geometry(n_cus, parts);
    g.zero_fill = false; g.gap_rows = true;
    return g;
}
inline TtsGeometry tts_paired_geometry(int n_cus) {
    TtsGeometry g;
    g.max_slots = kTtsPairedSlots; g.max_rows = kTtsMaxRows; g.tiles_wanted = n_cus; g.zero_fill = true; g.threads = kTtsPairedThreads;
    return g;
}
// Columns [c0, c1) of a CSR matrix as a matrix of its own (same rows and width: column ids stay global, x is shared).
Csr csr_column_range(const Csr& m, int32_t c0, int32_t c1);

}  // namespace hispmv
