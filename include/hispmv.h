/* hispmv.h -- C ABI of libhispmv.so, the MI355X (gfx950) drop-in for the SpMV hot path of
 * mfkiwl/HiSpMV:  y = alpha * A * x + beta * bias,  A sparse (slice stream) or dense (GeMV overlay).
 *
 * Every entry point replaces one piece of the reference's pybind11 class FpgaHandle
 * (pyhispmv/include/fpga_handle.h:9-74, pyhispmv/src/fpga_handle.cpp, bound in
 * pyhispmv/src/pyhispmv_bindings.cpp:3-39).  The reference-side binding a maintainer
 * would write against this header is shown in INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes only; all functions return 0 (or a handle index
 * >= 0) on success, HISPMV_FULL (-1) where the reference returns -1, and another negative
 * HISPMV_E* code otherwise -- never exit(), never a C++ exception across the boundary
 * (the reference calls std::exit on device errors, fpga_handle.cpp:58-64,82-88,267-270).
 * hispmv_last_error() gives the message.  Calls on one context are serialised internally.
 * Host pointers are borrowed for the duration of the call only.
 */
#ifndef HISPMV_H
#define HISPMV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HISPMV_OK          0
#define HISPMV_FULL       -1   /* matrix does not fit the arena (fpga_handle.cpp:192-195,235-238) */
#define HISPMV_EINVAL     -2   /* bad argument (negative device id, empty path, bad shape, index out of range) */
#define HISPMV_EDEVICE    -3   /* HIP runtime error / no gfx950 device */
#define HISPMV_ESTATE     -4   /* call out of order (run before load/select) */
#define HISPMV_ENOTDENSE  -5   /* dense handle requested on a context created without dense_overlay (spmv-helper.cpp:718) */
#define HISPMV_EIO        -6   /* MatrixMarket file unreadable / malformed */
#define HISPMV_ENOMEM     -7

typedef struct hispmv_ctx hispmv_ctx;

/* ---- FpgaHandle::FpgaHandle (fpga_handle.cpp:40-154; bindings :8-12) -------------------------
 * xclbin_path is accepted for signature compatibility; it must be non-empty (fpga_handle.cpp:70-71)
 * but is not opened.  device_id is the HIP device ordinal; negative is an error (:51-52).
 * The hardware tuple (num_ch_A.. row_dist_net) is recorded: dense_overlay gates
 * hispmv_create_dense_handle exactly as in the reference, num_ch_A sizes the default arena
 * (num_ch_A x 256 MiB, fpga_handle.h:12) unless HISPMV_ARENA_BYTES / hispmv_set_arena_bytes overrides it. */
int hispmv_create(hispmv_ctx** out, const char* xclbin_path, int device_id,
                  int num_ch_A, int num_ch_B, int num_ch_C, int urams_per_pe, int fp_acc_latency,
                  int dense_overlay, int pre_accumulator, int row_dist_net);

/* The reference never frees its handles (fpga_handle.cpp:177,220); we do. */
void hispmv_destroy(hispmv_ctx* ctx);

/* Message of the last failing call on ctx (or of the last failing hispmv_create when ctx is NULL). */
const char* hispmv_last_error(const hispmv_ctx* ctx);

/* Arena budget in bytes shared by all handles of the context ("-1 when full" contract). */
int hispmv_set_arena_bytes(hispmv_ctx* ctx, int64_t bytes);
int64_t hispmv_arena_bytes_used(const hispmv_ctx* ctx);

/* ---- FpgaHandle::createSparseMtxHandle (fpga_handle.cpp:156-207; bindings :20-23) --------------
 * COO triplets, unsorted and duplicated entries allowed (duplicates are summed by the
 * multiply, not coalesced -- spmv-helper.cpp:139-227).  Returns the handle index (0,1,2.. in
 * creation order), HISPMV_FULL, or an error. */
int hispmv_create_sparse_handle(hispmv_ctx* ctx, const int32_t* coo_rows, const int32_t* coo_cols,
                                const float* coo_values, int64_t nnz, int32_t rows, int32_t cols);

/* HiSpmvHandle::prepareSparseMtxForFPGA(mtx_file) (spmv-helper.cpp:642-646): MatrixMarket
 * coordinate file.  flavor 0 = common/ loader semantics (spmv-helper.cpp:34-136), 1 = cpu/ loader
 * semantics (cpu/src/helper_functions.cpp:91-146). */
int hispmv_create_sparse_handle_from_mtx(hispmv_ctx* ctx, const char* mtx_path, int flavor);

/* CSR input (what cpu/src/main.cpp:26-33 hands MKL): row_ptr[rows+1] starting at 0 and non-decreasing, columns inside
 * [0, cols).  Rows whose columns are not ascending (scipy: has_sorted_indices == False) are sorted by column on the
 * way in (stably: duplicates keep their order); NULL col_idx / values with nnz > 0 -> HISPMV_EINVAL. */
int hispmv_create_sparse_handle_from_csr(hispmv_ctx* ctx, const int32_t* row_ptr, const int32_t* col_idx,
                                         const float* values, int32_t rows, int32_t cols);

/* ---- FpgaHandle::createDenseMtxHandle (fpga_handle.cpp:209-250; bindings :15-18) ---------------
 * Row-major rows x cols fp32. */
int hispmv_create_dense_handle(hispmv_ctx* ctx, const float* flattened_dense_values, int32_t rows, int32_t cols);

/* ---- FpgaHandle::loadMatrices (fpga_handle.cpp:252-264) -----------------------------------------
 * Uploads every handle created so far to HBM.  Idempotent (the reference corrupts its offsets
 * when called twice, :259-261 -- not replicated). */
int hispmv_load_matrices(hispmv_ctx* ctx);

/* ---- FpgaHandle::selectMatrix (fpga_handle.cpp:266-283) -----------------------------------------
 * Out of range -> HISPMV_EINVAL (reference: exit, :267-270). */
int hispmv_select_matrix(hispmv_ctx* ctx, uint32_t matrix_idx);

/* ---- FpgaHandle::runKernel (fpga_handle.cpp:286-321) --------------------------------------------
 * y = alpha * A * x + beta * bias for the selected matrix; x[cols], bias[rows] read-only,
 * y[rows] written; blocking.  bias is not read when beta == 0 (BLAS/MKL convention, cpu/src/main.cpp:39). */
int hispmv_run_kernel(hispmv_ctx* ctx, const float* x, const float* bias, float* y, float alpha, float beta);

/* ---- FpgaHandle::runLinear (fpga_handle.cpp:323-388) --------------------------------------------
 * alpha = beta = 1; num_vecs = x_len / cols; y_out[num_vecs * rows]; same bias for every vector;
 * independent of the select_matrix state. */
int hispmv_linear(hispmv_ctx* ctx, int matrix_idx, const float* x, int64_t x_len, const float* bias, float* y_out);

/* ---- device-resident entry points (no counterpart in the reference, whose vectors always cross
 * PCIe -- fpga_handle.cpp:306-320).  d_* are device pointers; the launch is asynchronous on
 * `stream` (a hipStream_t).  ONE rule for every entry point that takes a stream (hispmv_spmv_device, hispmv_spmv_device_batch,
 * hispmv_boundary_pack, hispmv_boundary_apply): NULL = the context's own stream (created non-blocking: it does NOT synchronise
 * with HIP's null stream), anything else = the caller's stream.  Work issued with NULL is ordered with other NULL work of the
 * same context; hispmv_synchronize waits for it.  Used by bench.py and the multi-GPU driver. */
int hispmv_spmv_device(hispmv_ctx* ctx, int matrix_idx, const float* d_x, const float* d_bias, float* d_y,
                       float alpha, float beta, void* stream);
/* Waits for the context's stream and for the last caller-supplied stream a launch of this context went to, then reports
 * a device-side error of those launches (HISPMV_EDEVICE: an in-kernel bounded wait expired).  Launches sent to OTHER
 * caller streams before that must be synchronised by the caller first. */
int hispmv_synchronize(hispmv_ctx* ctx);

/* Whole-kernel device time of the last hispmv_spmv_device/run_kernel/linear launch sequence,
 * measured with HIP events on the launch stream (milliseconds); negative if unavailable. */
float hispmv_last_kernel_ms(hispmv_ctx* ctx);

/* Diagnostics of the HIP-graph replay of hispmv_spmv_device_batch (HISPMV_BATCH_GRAPH=1; plain launches by default): out = {graphs instantiated, alpha patches applied to an
 * instantiated graph}.  A call signature (handles, vectors, beta) is captured and instantiated ONCE; calls that differ only
 * in alpha patch the graph's kernel nodes (hipGraphExecKernelNodeSetParams) instead of instantiating again. */
int hispmv_batch_graph_stats(hispmv_ctx* ctx, int64_t out[2]);
/* How the LAST hispmv_spmv_device_batch call of this context was issued (diagnostics; bench.py names the kernels of its step from
 * it): out = {launches of the call, 1 if its slice groups and tiles ran as items of the step kernel's queue (one persistent
 * workgroup per CU, hispmv_kernels.hip: spmv_step_kernel), items of that queue, HIP streams the main launches were spread over}.
 * HISPMV_ESTATE before the first batch call. */
int hispmv_batch_call_info(hispmv_ctx* ctx, int64_t out[4]);

/* n independent SpMVs y_i = alpha*A_i*x_i + beta*bias_i on loaded handles idx[i] in as few launches as possible: the
 * workgroups of all matrices with the same workgroup size share ONE grid (plus one fix-up launch), so small matrices no
 * longer pay 6-20 us of launch latency each (no reference counterpart: the reference runs one matrix at a time,
 * fpga_handle.cpp:286-321).  idx, d_x, d_bias, d_y are HOST arrays of n entries (device pointers inside); the y_i must
 * be distinct and a sparse handle may appear only once (its carry buffers belong to the handle).  Rows cut by slice
 * boundaries always take the fix-up variant here, so a result may differ in the last
 * bit from hispmv_spmv_device on a matrix whose single launch uses the look-back variant.  Asynchronous on `stream`. */
int hispmv_spmv_device_batch(hispmv_ctx* ctx, int32_t n, const int32_t* idx, const float* const* d_x,
                             const float* const* d_bias, float* const* d_y, float alpha, float beta, void* stream);

/* Time `reps` back-to-back launches of matrix_idx on the context stream with HIP events
 * (kernel-only, the reference's convention: spmv-helper.cpp:1030-1035).  Returns ms per launch. */
float hispmv_time_device(hispmv_ctx* ctx, int matrix_idx, const float* d_x, const float* d_bias, float* d_y,
                         float alpha, float beta, int reps);

/* ---- multi-GPU boundary rows (no reference counterpart: the reference is single-device) ----
 * A matrix whose element sequence is split over ranks has at most one row cut at each rank boundary.  Per step
 * every rank publishes, for each of its n matrices, the last entry of its local y when that row continues on the
 * next rank (its "tail"), the tails are all-gathered (torch.distributed / RCCL), and each rank adds the chain of
 * tails that feeds its first row.  These two launches are the device side of that step on `stream`:
 *   pack:  send[i] = mask[i] * *last[i]                                   (last[i] may be NULL: 0)
 *   apply: *first[i] += sum_r recv[r*n + i] * weights[i*world + r]        (first[i] NULL: nothing), r ascending
 * All pointers are device pointers (last/first: device arrays of n device pointers).  `stream` as everywhere in this header:
 * NULL = the context's stream, so SpMVs and boundary kernels issued with NULL run on one queue, in order. */
int hispmv_boundary_pack(hispmv_ctx* ctx, const float* const* d_last, const float* d_mask, float* d_send, int32_t n, void* stream);
int hispmv_boundary_apply(hispmv_ctx* ctx, float* const* d_first, const float* d_recv, const float* d_weights, int32_t n, int32_t world,
                          void* stream);

/* ---- getters (HiSpmvHandle getters, spmv-helper.cpp:752-810) ------------------------------------ */
typedef struct hispmv_matrix_info {
    int32_t rows, cols;
    int64_t nnz;            /* getNNZ */
    int32_t is_dense;       /* isDense */
    int32_t loaded;
    int64_t n_slices;       /* wavefront slices (the analogue of getRunLength's beats) */
    int64_t n_elems;        /* stream elements before tail padding (nnz + empty-row fillers) */
    int64_t n_split_rows;   /* rows shared between slices (the analogue of the shared-row list) */
    int64_t device_bytes;   /* bytes this handle takes in the arena */
    double prep_seconds;    /* host preprocessing time ("Pre-processing Time") */
    int32_t block_threads;  /* launch plan chosen at load time: workgroup size, */
    int32_t group_slices;   /*   slices per workgroup (format 1: K-slots per block of the tile geometry, 28 or 13), */
    int32_t lds_bytes;      /*   LDS bytes of the x window (0 = x gathered through L2) */
    int32_t col_tiles;      /* number of column tiles (1 = untiled) */
    int32_t carry_lookback; /* 1 = rows shared between slices are merged inside the launch (look-back), 0 = fix-up launch */
    int32_t col_tile_width; /* columns per tile when col_tiles > 1, else 0 */
    int32_t col_tile_base;  /* tile t covers columns [base + t*width, base + (t+1)*width) of the range holding 99.8 % of the
                               elements; the first tile also takes every column below, the last every column above */
    int32_t compact_slices; /* slices stored with 6-byte elements (fp32 value + 16-bit {rowEnd, index into the LDS window of x -- or into the
                               owning wavefront's stray area behind it, for up to 64 elements per slice whose column lies outside the window});
                               the others take 8 bytes per element (32-bit meta) */
    int32_t format;         /* 0 = slice stream (rows in order, segmented scan); 1 = transposed tile stream (scattered short-row matrices:
                               row tiles with LDS accumulators, elements streamed sorted by column, transposed through LDS; n_slices then
                               counts its 1024-word slices, n_split_rows the rows cut into pieces: longer than a tile and a quarter) */
    float tts_lines_per_gather;  /* format 1: distinct 128-byte lines of x per 64-lane gather (64 = no lane shares a line) */
    int32_t tile_kind;      /* col_tiles > 1: 1 = tiles are column ranges (col_tile_base / col_tile_width above); 2 = BAND tiles: the same
                               base / width describe ranges of the OFFSET col - row*cols/rows from the scaled diagonal (a banded matrix
                               whose band is wider than an LDS window, cut along the diagonal); 3 = STRAY SPLIT: part 0 holds the elements that lie
                               inside the x window of their workgroup (6-byte elements from LDS), part 1 the few per cent that do not (gathered
                               through L2 into a partial vector the tail launch adds); 0 = untiled */
    int32_t batch_group_slices; /* > 0: the handle also holds a BATCH LAYOUT of its slices -- groups of this many slices (twice group_slices,
                               half as many workgroups) that hispmv_spmv_device_batch uses when a call shares the chip between its matrices
                               (two launch lanes); single launches keep group_slices.  0 = none.  Same results bit for bit. */
} hispmv_matrix_info;
int hispmv_get_matrix_info(const hispmv_ctx* ctx, int matrix_idx, hispmv_matrix_info* out);
int hispmv_num_matrices(const hispmv_ctx* ctx);

/* ---- host-only preprocessor access (no device needed): lets tests check the CSR indices and the
 * packed stream against the oracle on a CPU-only box.  Mirrors HiSpmvHandle::getPreparedMtx
 * (spmv-helper.cpp:800-802). ------------------------------------------------------------------- */
typedef struct hispmv_prep hispmv_prep;
int hispmv_prep_from_coo(hispmv_prep** out, const int32_t* coo_rows, const int32_t* coo_cols,
                         const float* coo_values, int64_t nnz, int32_t rows, int32_t cols);
int hispmv_prep_from_mtx(hispmv_prep** out, const char* mtx_path, int flavor);
/* The same object with COO -> CSR (stable radix sort) and CSR -> slice stream computed ON THE DEVICE `device_id`
 * (retires the hotspot of common/src/spmv-helper.cpp:139-227): CSR, words, headers and split-row list
 * are byte-identical to hispmv_prep_from_coo's.  seconds (may be NULL) = {upload, COO->CSR on the device, row offsets on
 * the host, stream on the device, downloads}.  create_sparse_handle uses this path from 2 M entries
 * (HISPMV_PREP=host|device|auto). */
int hispmv_prep_from_coo_device(hispmv_prep** out, int device_id, const int32_t* coo_rows, const int32_t* coo_cols,
                                const float* coo_values, int64_t nnz, int32_t rows, int32_t cols, double seconds[5]);
void hispmv_prep_free(hispmv_prep* p);
const char* hispmv_prep_last_error(void);
/* dims[0..7] = rows, cols, nnz, n_elems, n_slices, slice_elems, n_split_rows, stream_bytes */
int hispmv_prep_dims(const hispmv_prep* p, int64_t dims[8]);
const int64_t* hispmv_prep_csr_row_ptr(const hispmv_prep* p);
const int32_t* hispmv_prep_csr_col(const hispmv_prep* p);
const float* hispmv_prep_csr_val(const hispmv_prep* p);
const uint64_t* hispmv_prep_words(const hispmv_prep* p);     /* n_slices * slice_elems 64-bit words */
const int32_t* hispmv_prep_slice_hdr(const hispmv_prep* p);   /* n_slices x {row_base, chain_len, x_base, x_span} */
const int32_t* hispmv_prep_fix(const hispmv_prep* p);         /* n_split_rows x {row, first_slice, len, 0} */

/* Launch plan the loader would choose for this stream on a device with n_cus compute units (host-only, no
 * device needed): plan[0..5] = workgroup threads, slices per workgroup, x-window LDS floats, row-total LDS
 * floats per wavefront, workgroups, total dynamic LDS bytes per workgroup. */
int hispmv_prep_plan(const hispmv_prep* p, int n_cus, int64_t plan[6]);

/* The FORMAT AND TILING the loader would choose for this matrix on a device with n_cus compute units -- the MI355X analogue of
 * the reference's per-matrix configuration search (automation_tool/src/dse.py:23-95) -- computed by the same host-only function
 * hispmv_create_sparse_handle* calls (hispmv_amd/csrc/hispmv_choose.cpp; no device needed).  out[0..13] = format (0 slice
 * stream, 1 transposed tile stream), tile kind (0 untiled, 1 column tiles, 2 band tiles), parts, tile width, tile base, 1 if the
 * tiles gather x through L2 (XCD-pinned in a batch call), workgroup threads / slices per workgroup (tile stream: K-slots per
 * block) / LDS window floats of part 0, slices, stream elements, rows cut between slices (pieces), 1000 x lines per gather of a
 * tile stream, elements that gather x through L2.  Honours the HISPMV_FORMAT / _BAND_TILES / _TTS_GEOMETRY / _COL_TILE_BYTES /
 * _TTS_MIN_NNZ switches like the loader. */
int hispmv_prep_choose_format(const hispmv_prep* p, int n_cus, int64_t out[16]);

/* The order of the step kernel's queue (hispmv_spmv_device_batch, hispmv_batch_call_info), host-only (hispmv_choose.cpp:
 * order_step_queue, the function the batch planner calls): n_slice slice items and n_tile tiles with a cost each (microseconds of a
 * CU), n_wg workgroups; mode 0 = long tiles (> a quarter of the step) alternating with the longest slice items, then longest first
 * (default), 1 = longest first, 2 = tiles then slice items as given (3, 4, 16*t + s: experiments -- two / three slice items per long tile,
 * cycles of t long tiles and s slice items).  out_class[i] (0 slice item, 1 tile) and out_index[i] name the
 * item at queue position i (n_slice + n_tile positions).  No reference counterpart (the reference runs one matrix at a time). */
int hispmv_prep_step_queue(const double* slice_costs, int32_t n_slice, const double* tile_costs, int32_t n_tile, int32_t n_wg, int32_t mode,
                           int32_t* out_class, int32_t* out_index);

/* inside[nnz] (CSR order): 1 for the entries whose 64-byte block of x is held by the x window of their workgroup under the launch
 * plan for n_cus compute units, 0 for the entries that gather through L2 -- the criterion by which the loader splits a matrix with a
 * few per cent of stray couplings into a windowed part and a stray part (hispmv_matrix_info.tile_kind 3). */
int hispmv_prep_window_membership(const hispmv_prep* p, int n_cus, uint8_t* inside);

/* Applies that plan to the prepared stream IN PLACE (the words of LDS-staged groups then carry window
 * indices instead of columns -- or 0x40000000 | column for the elements of the group whose 64-byte block of x
 * is not in the window) and exposes its tables: groups = n x {frag_begin, frag_count, lds_floats, elements
 * outside the window}, frags = m x {col_start, len, lds_off, 0}.  counts[0..1] = n, m. */
int hispmv_prep_apply_plan(hispmv_prep* p, int n_cus, int64_t counts[2]);
const int32_t* hispmv_prep_groups(const hispmv_prep* p);
/* The planned stream in its DEVICE LAYOUT (hispmv_amd/csrc/hispmv_format.h; call hispmv_prep_apply_plan first): counts = {bytes, groups,
 * compact slices, slices of groups with stray slots, LDS floats of the wavefronts' stray areas, LDS floats of the x window}.  Arrays:
 * 0 = the slices, group after group (compact: 1024 x fp32 then 1024 x u16 {rowEnd:1 | LDS index:15}; wide: 1024 x u32 metas);
 * 1 = groups x {frag_begin, frag_count, offset of the group's first slice / 2048, 1 = compact | 2 = stray slots}; 2 = slices x 64 stray
 * columns (0xffffffff = unused; empty when no group has stray slots).  For tests: the packer without a device. */
int hispmv_prep_device_stream(hispmv_prep* p, int64_t counts[6]);
const void* hispmv_prep_device_array(const hispmv_prep* p, int which);
/* The same layout written on the DEVICE from the planned host words (the kernel hispmv_load_matrices runs with HISPMV_LAYOUT=device):
 * after hispmv_prep_apply_plan + hispmv_prep_device_stream; bytes_out takes counts[0] bytes, stray_cols_out slices x 64 u32 (may be
 * NULL when counts[4] == 0).  For tests (device == host, byte for byte); HISPMV_EDEVICE without a GPU. */
int hispmv_prep_device_stream_on_device(hispmv_prep* p, int device_id, uint8_t* bytes_out, uint32_t* stray_cols_out);
const int32_t* hispmv_prep_frags(const hispmv_prep* p);

/* Number of device / pinned-memory frees the runtime rejected since the library was loaded (a pointer released twice
 * or never allocated); 0 in a correct run.  For tests. */
int64_t hispmv_free_failures(void);

/* The transposed tile stream of the prepared matrix (the second device format: hispmv_matrix_info.format == 1), packed
 * on the host: counts = {tiles, blocks, column-order slices of 1024 words, row-major chunks of 1024 slots, fillers,
 * padding words, max rows of a tile, max slots of a block}; arrays: 0 words (per slice 1024 x fp32 then 1024 x
 * {col_off:16 | slot:16}), 1 col_base (int32 per slice), 2 flags (64 x u16 per chunk), 3 chunk_info ({rows ending
 * before, chain_len} per chunk), 4 tiles ({row0, n_rows, block_begin, n_blocks}; row0 < 0: carry tile), 5 blocks (8 x int32: slice_begin,
 * n_slices, chunk_begin, n_chunks, n_slots, 0, 0, 0).  target_tile_elems 0 = the loader's choice; small_geometry: 0 = tiles of
 * <= 8192 rows and blocks of <= 28 K slots (one workgroup per CU), 1 = <= 4096 rows / 13 K slots (two per CU: what the
 * loader takes when a gather of the tall geometry touches <= 8 lines; hispmv_matrix_info.group_slices = 28 or 13), 2 + q =
 * column part q (0 or 1) of the TALL geometry as the loader builds it for 256 CUs: the matrix cut at the column that halves
 * its elements, each half packed into tiles of <= 16384 rows and blocks of <= 23 K slots in which rows absent from a block
 * own a slot but no word (hispmv_matrix_info.group_slices = 23, col_tiles = 2: part 0 gives alpha*A_0*x + beta*bias, part 1
 * the partial vector alpha*A_1*x that the merge launch adds), 8 + q = column part q of the same geometry with GAP-CODED row
 * ends (HISPMV_TTS_GEOMETRY=tallgap): absent rows own no slot, a row end's 2-bit code = bit of array 2 | bit of array 7 << 1
 * (1 end, next row present; 2 end, one absent row follows; 3 end, two follow), chunk_info = {last slot-owning row before
 * the chunk, chain_len | first code << 16}; array 7 = flags_hi (64 x u16 per chunk; null for the other geometries). */
int hispmv_prep_build_tts(hispmv_prep* p, int64_t target_tile_elems, int small_geometry, int64_t counts[8], double* lines_per_gather);
const void* hispmv_prep_tts_array(const hispmv_prep* p, int which);
/* Rows longer than two tiles are cut into pieces, each a tile of its own; all but a row's last piece are carry tiles
 * (tiles[.].row0 = -(carry index + 1)) whose sums a fix-up launch adds: counts = {rows cut, carry tiles}; array 6 of
 * hispmv_prep_tts_array = {row, first carry, carries, 0} per row cut (int32 x 4). */
int hispmv_prep_tts_pieces(const hispmv_prep* p, int64_t counts[2]);

/* OpenMP threads the host preprocessor uses: the CPUs this process may use (cgroup cpu.max quota, e.g. 16 of the 256 a GPU
 * box shows), set once at the first call of the library unless OMP_NUM_THREADS is given. */
int hispmv_host_threads(void);

/* Library identification: "hispmv-amd <version> gfx950". */
const char* hispmv_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HISPMV_H */
