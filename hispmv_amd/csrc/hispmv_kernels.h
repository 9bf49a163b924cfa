// hispmv_kernels.h -- launchers of the gfx950 kernels (hispmv_kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#include <cstdint>

namespace hispmv {

struct SpmvDeviceMatrix {
    const void* words = nullptr;        // the slices in their device layout (hispmv_format.h), group after group
    const int4* hdr = nullptr;          // n_slices x {row_base, chain_len, rows ending in the slice, 1 if elements lie outside the x window}
    const int4* fix_short = nullptr;    // {row, first_slice, len, 0}, len <= kFixShortMax
    const int4* fix_long = nullptr;     // same, len > kFixShortMax
    const int4* groups = nullptr;       // n_groups x {frag_begin, frag_count, offset of the group's first slice / kSliceUnit, 1 = compact}
    const int4* frags = nullptr;        // {col_start, len, lds_off, 0}
    float* carry = nullptr;             // n_slices: partial sum each slice hands to the next
    int64_t n_groups = 0;
    int32_t group_slices = 8;           // slices per workgroup
    int32_t block_threads = 256;        // workgroup size (64 * wavefronts)
    int32_t lds_floats = 0;             // dynamic LDS (floats) for the x window (+ the wavefronts' stray areas behind it); 0 = gather x from L2
    bool has_strays = false;            // some groups use stray slots (hispmv_plan.h): the columns of every slice's strays follow the headers
    int32_t ytile_floats = 1024;        // LDS floats per wavefront for the row totals of one slice (>= max rows ending in a slice)
    // single-launch carry hand-off between slices (look-back); when false the fix-up kernels run instead
    bool lookback = true;
    unsigned long long* gran = nullptr;     // n_slices x {fp32 carry, launch tag}
    unsigned long long* ticket = nullptr;   // group tickets, monotonic over launches
    int* err = nullptr;                     // set to 1 if a bounded wait expired
    unsigned long long launches = 0;        // launches so far (host side)
    unsigned long long ticket_launches = 0; // launches that drew tickets
    bool use_ticket = true;                 // false when the whole grid is co-resident (no ordering needed)
    int64_t n_slices = 0;
    int32_t n_fix_short = 0, n_fix_long = 0;
    int32_t rows = 0, cols = 0;
};

// Transposed tile stream (hispmv_tts.h): device tables of one matrix.
struct TtsDeviceMatrix {
    const void* words = nullptr;        // per column-order slice: 1024 x fp32, then 1024 x {col_off:16 | slot:16}
    const int32_t* col_base = nullptr;  // per slice
    const uint16_t* flags = nullptr;    // per row-major chunk: 64 x u16 row-end bits
    const uint16_t* flags_hi = nullptr; // gap-coded row ends (TtsGeometry::gap_rows): BOTH planes interleaved, one u32 per lane and chunk (bit 0 of the codes in
                                        // the lower half, bit 1 in the upper half) -- the kernel reads this array instead of `flags`; NULL otherwise
    const int2* chunk_info = nullptr;   // per chunk: {rows ending before it, chain_len}
    const int4* tiles = nullptr;        // {row0, n_rows, block_begin, n_blocks}; row0 < 0: carry tile, its sum -> carry[-row0 - 1]
    const int4* blocks = nullptr;       // 2 x int4 per block: {slice_begin, n_slices, chunk_begin, n_chunks}, {n_slots, 0, 0, 0}
    float* carry = nullptr;             // raw sums of the carry tiles (pieces of rows longer than two tiles)
    const int4* fix = nullptr;          // {row, first carry, carries, 0} per such row: y[row] += alpha * sum (spmv_fixup_short_kernel)
    int32_t n_fix = 0;
    int32_t zero_fill = 0;              // 1: rows absent from a block have no stream word, the staging is kept zero-filled (TtsGeometry::zero_fill);
                                        // 2: gap-coded row ends (TtsGeometry::gap_rows): absent rows own nothing, a row end's accumulator = running sum of the codes
    int32_t n_tiles = 0, rows = 0, cols = 0;
    int32_t n_carry = 0;                // carry tiles; the carry buffer holds kTtsMaxVectors x n_carry (vector v of a batched launch: carry + v * n_carry)
    int32_t acc_floats = 0, staging_floats = 0;    // LDS: accumulators (max rows of a tile), staging (max slots of a block + dummy)
    int32_t xlds_floats = 0;                       // > 0: x is short enough for the LDS (cols rounded up to 64; kTtsXldsMax)
    int32_t batch_stage_floats = 0;                // per-vector staging area of the NV-vector kernel: the LARGEST block of this matrix in whole chunks + 64
    int32_t threads = 512;                         // workgroup size (hispmv_tts.h: kTtsThreads)
};
struct TtsEntry {                       // multi-matrix launch: one per matrix
    TtsDeviceMatrix m;
    const float* x; const float* bias; float* y;
    float beta; int32_t pad;
};

struct GemvEntry {                      // multi-matrix launch of the dense overlay: one per matrix
    const float* W; const float* x; const float* bias; float* y;
    int32_t rows, cols;
    float beta; int32_t pad;
};

struct LookbackArgs {
    unsigned long long* gran;
    unsigned long long* ticket;
    int* err;
    unsigned long long ticket_base;
    unsigned epoch;
    int use_ticket;
};

// Multi-matrix launch (hispmv_spmv_device_batch): device table entry per matrix part ...
struct MultiEntry {
    const void* words; const int4* hdr; const int4* groups; const int4* frags;
    const float* x; const float* bias; float* y; float* carry;
    long long n_slices;
    int32_t group_slices, lds_floats, ytile_floats, cols, rows;
    float beta;                     // per entry: 0 for the column tiles t > 0 of a matrix (they write partial vectors, no bias)
};
struct MultiFixEntry { const int4* fix; const float* carry; float* y; int32_t n, pad; };
struct MultiMergeEntry { float* y; const float* parts; long long part_stride; int32_t n_parts, rows; };
// The tail of a batch call in ONE launch (spmv_tail_multi_kernel): the fix-up of the cut rows of every matrix part, and -- for
// matrices with column parts -- the merge of their partial vectors, which then applies the fix-ups of ITS rows itself:
//   y[i] = (y[i] + alpha * chain_0(i)) + (part_1[i] + alpha * chain_1(i)) + ...      (the bits of fix-up launch + merge launch)
// fix_of_row: (n_parts + 1) x rows int32, index into that part's fix list or -1.
constexpr int kTailMaxParts = 9;      // tile 0 + up to 8 column tiles
struct TailMergeEntry {
    float* y; const float* parts; long long part_stride; int32_t n_parts, rows;
    const int32_t* fix_of_row;
    const int4* fix[kTailMaxParts];
    const float* carry[kTailMaxParts];
};
// ... and, as a kernel argument, where each entry's workgroups begin in the grid (begin[n] = grid size)
// An ITEM of a multi launch is one table entry, or -- `tiles[k]` = 2, 4 or 8 -- the column tiles of one matrix that
// gather x through L2: consecutive table entries from `first[k]`, pinned to disjoint XCD subsets.  Workgroups are dealt
// round-robin over the 8 XCDs (MI355X_MICROARCH.md, workgroup dispatch), so blocks with the same index mod 8 share an
// L2; the item's range starts at a multiple of 8 and block q of it belongs to tile (q % 8) / (8 / tiles): every XCD then
// serves ONE tile and its L2 holds that tile's part of x, while all tiles run in the same round.
constexpr int kMultiMax = 32;
struct MultiPrefix {
    int32_t n, pad;
    long long begin[kMultiMax + 1];
    uint8_t first[kMultiMax];
    uint8_t tiles[kMultiMax];
};

constexpr int kMaxBatch = 4;                // vectors one pass of the batched slice kernel takes (carry holds kMaxBatch * n_slices)

// Once per process/device before the first launch (raises the dynamic-LDS limit of the slice kernels).
hipError_t prepare_spmv_kernels();

// y = alpha*A*x + beta*bias.  Two launches on `stream`: the slice kernel, then (if any row is
// shared between slices) the carry fix-up.  Returns the first HIP error.
hipError_t launch_spmv(SpmvDeviceMatrix& m, const float* x, const float* bias, float* y,
                       float alpha, float beta, hipStream_t stream, bool fixup_only = false);

// Batched SpMV (linear with several vectors): how many vectors (4, 2 or 1 = use launch_spmv) one pass can take for
// this matrix, and the launch: vector v is x + v*cols -> y + v*rows; bias_stride 0 (shared) or rows (per vector).
// Fix-up carry variant whatever m.lookback says: per vector bitwise equal to launch_spmv with lookback = false.
int spmv_batch_width(const SpmvDeviceMatrix& m, int64_t vecs);
hipError_t launch_spmv_batched(SpmvDeviceMatrix& m, int nv, const float* x, const float* bias, int bias_stride, float* y,
                               float alpha, float beta, hipStream_t stream);

// Multi-matrix launch: `n` (<= kMultiMax) matrix parts with the same workgroup size in one grid, each writing carry[]
// for rows cut by slice boundaries; `d_table` is the device copy of their MultiEntry descriptors (the caller owns and
// caches it); every entry carries its own beta (0 = no bias read).  launch_fixup_multi then finishes the cut rows of `n` parts
// (any workgroup sizes) in one launch; parts with fix_long rows get their own extra launch.
// `item_tiles` (n_items entries summing to n; NULL = all 1) groups consecutive parts into XCD-pinned column-tile sets.
hipError_t launch_spmv_multi(const SpmvDeviceMatrix* const* parts, int n, const uint8_t* item_tiles, int n_items,
                             const MultiEntry* d_table, float alpha, hipStream_t stream);
// y += parts[0] + parts[1] + ... (column tiles t > 0 of one matrix, each of `rows` floats, `part_stride` apart); nv vectors
// of a batched pass: vector v is y + v*y_stride, its partial vectors parts + v*vec_stride.
hipError_t launch_merge_parts(float* y, const float* parts, int n_parts, int64_t part_stride, int32_t rows, int nv,
                              int64_t y_stride, int64_t vec_stride, hipStream_t stream);
// The same for `n` matrices in one launch (d_table: device copy of their MultiMergeEntry descriptors).
hipError_t launch_merge_multi(const int32_t* rows, int n, const MultiMergeEntry* d_table, hipStream_t stream);
hipError_t launch_fixup_multi(const SpmvDeviceMatrix* const* parts, float* const* ys, int n, const MultiFixEntry* d_fix_table,
                              float alpha, hipStream_t stream);
// The long chains of one part (a wavefront per row that spans more than kFixShortMax slices).
hipError_t launch_fixup_long(const SpmvDeviceMatrix& m, float* y, float alpha, hipStream_t stream);
// Fix-up (n_fix parts, short chains only) and merge (n_merge matrices with column parts) in one launch; tables on the device.
hipError_t launch_tail_multi(const int32_t* fix_counts, int n_fix, const MultiFixEntry* d_fix_table, const int32_t* merge_rows, int n_merge,
                             const TailMergeEntry* d_merge_table, float alpha, hipStream_t stream);

// Transposed tile stream: y = alpha*A*x + beta*bias in ONE launch (one workgroup of 1024 threads per row tile; no carry
// buffers, no fix-up launch); launch_tts_multi: the tiles of `n` matrices in one grid (d_table: device copy of TtsEntry).
hipError_t launch_tts(const TtsDeviceMatrix& m, const float* x, const float* bias, float* y, float alpha, float beta, hipStream_t stream);
// `nv` <= kTtsMaxVectors vectors in one launch (x + v*cols -> y + v*rows, shared bias): per vector bitwise equal to launch_tts.
constexpr int kTtsMaxVectors = 8;
constexpr int kTtsXldsMax = 16 * 1024;      // floats of x the tile-stream kernels keep in the LDS (64 KiB)
bool tts_x_in_lds(const TtsDeviceMatrix& m, int nv);
// 4 / 2: that many vectors can share each pass over the words (their accumulators and staging areas fit the LDS together); 1: not
int tts_batch_width(const TtsDeviceMatrix& m, int64_t vecs);
hipError_t launch_tts_batched(const TtsDeviceMatrix& m, int nv, const float* x, const float* bias, float* y, float alpha, float beta, hipStream_t stream);
// (launch_tts finishes the rows it cut into pieces with a second tiny launch; in a multi-matrix call they ride in launch_fixup_multi)
// `item_parts` (n_items entries summing to n; NULL = all 1): 2 = the two column parts of one matrix, consecutive entries,
// pinned to XCDs 0-3 / 4-7 so that an XCD's L2 holds one part's half of x.
hipError_t launch_tts_multi(const TtsEntry* entries, int n, const uint8_t* item_parts, int n_items, const TtsEntry* d_table, float alpha,
                            hipStream_t stream);

// The step kernel (hispmv_kernels.hip: spmv_step_kernel): every slice group and every tile of a batch call as ITEMS of one queue,
// drawn by `workgroups` persistent 1024-thread workgroups.  d_items: n_items x {kind | table entry << 8, index}:
// kind 0 = group `index` of slice_table[entry] (a 1024-thread plan), 1 = groups index .. index + 3 of a 256-thread plan, 2 = tile
// `index` of tts_table[entry] (standard geometry, x gathered through the cache).  d_sync: two zeroed words the kernel rearms itself.
// `strays`: some slice part has stray slots.
struct StepArgs { const MultiEntry* slice_table; const TtsEntry* tts_table; const int2* items; unsigned* sync; unsigned n_items; float alpha; int pad, ticket_word; };
hipError_t launch_spmv_step(const MultiEntry* d_slice_table, const TtsEntry* d_tts_table, const void* d_items, unsigned n_items,
                            unsigned* d_sync, int workgroups, size_t lds_bytes, bool strays, float alpha, hipStream_t stream);
size_t tts_tile_lds_bytes(const TtsDeviceMatrix& m);

// Dense overlay: y = alpha*W*x + beta*bias, W row-major rows x cols.
hipError_t launch_gemv(const float* W, int32_t rows, int32_t cols, const float* x, const float* bias,
                       float* y, float alpha, float beta, hipStream_t stream);
// the row blocks of `n` (<= kMultiMax) dense matrices in one grid (d_table: device copy of the entries); per matrix
// bitwise equal to launch_gemv
hipError_t launch_gemv_multi(const GemvEntry* entries, int n, const GemvEntry* d_table, float alpha, hipStream_t stream);
// `vecs` vectors (x + v*cols -> y + v*rows, shared bias), 8/4/2/1 per pass over W; per vector bitwise equal to launch_gemv.
hipError_t launch_gemv_batched(const float* W, int32_t rows, int32_t cols, int64_t vecs, const float* x, const float* bias,
                               float* y, float alpha, float beta, hipStream_t stream);

// Patches alpha into every kernel node of an instantiated batch-call graph (`graph`: the captured graph it came from).
hipError_t graph_set_alpha(hipGraphExec_t exec, hipGraph_t graph, float alpha);

// Multi-GPU boundary rows (hispmv.h: hispmv_boundary_pack / hispmv_boundary_apply).
// n_floats floats (rounded up to whole float4s: both blocks are sized in multiples of 64 floats) from host-mapped memory to device memory
hipError_t launch_fetch_vectors(const float* src, float* dst, int64_t n_floats, hipStream_t stream);
hipError_t launch_boundary_pack(const float* const* last, const float* mask, float* send, int n, hipStream_t stream);
hipError_t launch_boundary_apply(float* const* first, const float* recv, const float* weights, int n, int world,
                                 hipStream_t stream);

}  // namespace hispmv
