/*
 * uspmv.h -- C ABI of the MI355X-native SELL-C-sigma SpMV / SpMMV engine (libuspmv.so).
 *
 * Drop-in boundary for the scs/crs kernel path of RRZE-HPC/Ultimate-SpMV.  The reference has no
 * FFI; its seam is the C++ function-pointer type SpmvKernel::OnePrecFuncPtr / MultiPrecFuncPtr
 * (code/classes_structs.hpp:283-333) and the free functions of code/interface.hpp.  Every entry
 * point below names the reference interface it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - plain pointers and sizes only; indices are 32-bit (`IT = int`, code/main.cpp:1711-1718),
 *     sizes 64-bit (`ST = long`, code/classes_structs.hpp:31);
 *   - every function returns a uspmv_status (0 = success) and NEVER calls exit(); the text of
 *     the last failure of the calling thread is available from uspmv_last_error();
 *   - pointers prefixed d_ are device (HBM) pointers of the current HIP device; all others are
 *     host pointers;  `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - device entry points are asynchronous with respect to the host (the reference's launchers
 *     call cudaDeviceSynchronize, code/classes_structs.hpp:1032-1034; callers that need that
 *     behaviour call uspmv_stream_synchronize);
 *   - there is NO CPU fallback: device entry points fail with USPMV_ERR_NO_DEVICE / USPMV_ERR_HIP
 *     when no gfx950 device is usable.
 */
#ifndef USPMV_H
#define USPMV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    USPMV_OK = 0,
    USPMV_ERR_INVALID = 1,     /* bad argument (NULL, negative size, unknown enum ...)        */
    USPMV_ERR_IO = 2,          /* file cannot be opened / parsed                               */
    USPMV_ERR_UNSUPPORTED = 3, /* valid request the engine does not implement                  */
    USPMV_ERR_OVERFLOW = 4,    /* a 32-bit index would overflow (code/utilities.hpp:1959-1962) */
    USPMV_ERR_NO_DEVICE = 5,   /* no HIP device visible                                        */
    USPMV_ERR_HIP = 6,         /* a HIP runtime call failed                                    */
    USPMV_ERR_ALLOC = 7,
    USPMV_ERR_COMM = 8         /* a peer rank failed, never arrived or disagrees (host communicator / transport) */
} uspmv_status;

typedef enum { USPMV_F64 = 0, USPMV_F32 = 1 } uspmv_dtype;            /* -dp / -sp              */
typedef enum { USPMV_COLWISE = 0, USPMV_ROWWISE = 1 } uspmv_layout;   /* Makefile:26-31         */
typedef enum { USPMV_SEG_ROWS = 0, USPMV_SEG_NNZ = 1 } uspmv_seg;     /* -seg_rows / -seg_nnz   */

typedef struct uspmv_coo uspmv_coo_t;     /* host COO,           MtxData  code/classes_structs.hpp:1169-1238 */
typedef struct uspmv_scs uspmv_scs_t;     /* host SELL-C-sigma,  ScsData  code/classes_structs.hpp:1313-1339 */
typedef struct uspmv_dmat uspmv_dmat_t;   /* device-resident SCS (what assign_spmv_kernel_gpu_data
                                             uploads, code/utilities.hpp:3721-3811)                         */
typedef struct uspmv_halo uspmv_halo_t;   /* per-rank halo description (ContextData,
                                             code/classes_structs.hpp:156-184)                             */

const char *uspmv_status_string(int status);
const char *uspmv_last_error(void);
const char *uspmv_version(void);
/* debugging aid: SIGSEGV / SIGBUS / SIGABRT print the native call stack (module+offset) to stderr before the default action */
int uspmv_debug_backtrace_on_crash(int on);

/* ------------------------------------------------------------------ L1: COO / MatrixMarket */
/* read_mtx (code/utilities.hpp:2148-2309) + mm_read_unsymmetric_sparse (code/mmio.h:132-263):
 * real / integer / pattern (= 0.01) coordinate files, general or symmetric (expanded), entries
 * stable-sorted by row. */
int uspmv_read_mtx(const char *path, uspmv_coo_t **out);
/* Binary cache of a COO handle (no reference counterpart): what uspmv_read_mtx returned --
 * expanded, row-sorted -- written / read back verbatim, so that later runs skip the text parse. */
int uspmv_coo_save(const uspmv_coo_t *m, const char *path);
int uspmv_coo_load(const char *path, uspmv_coo_t **out);
/* MatrixMarket writer (the reference only reads; used to produce inputs): coordinate real general, or -- symmetric != 0 -- the entries
 * with column <= row under a "symmetric" banner, which uspmv_read_mtx / read_mtx expand again.  17 significant digits. */
int uspmv_coo_write_mtx(const uspmv_coo_t *m, const char *path, int symmetric);
/* -equilibrate 1 of a one-precision run: equilibrate_matrix (code/utilities.hpp:2667-2685), in place.  (With
 * ap[dp_sp] the reference indexes two empty vectors here, code/main.cpp:1143-1153 -- undefined behaviour, not offered.) */
int uspmv_coo_equilibrate(uspmv_coo_t *m);
/* MtxData filled by a host application (API_doc.md:7-9).  Arrays are copied. */
int uspmv_coo_create(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t *I, const int32_t *J,
                     const double *values, uspmv_coo_t **out);
int uspmv_coo_dims(const uspmv_coo_t *m, int64_t *n_rows, int64_t *n_cols, int64_t *nnz);
/* borrowed pointers, valid until uspmv_coo_free */
int uspmv_coo_arrays(const uspmv_coo_t *m, const int32_t **I, const int32_t **J, const double **values);
void uspmv_coo_free(uspmv_coo_t *m);

/* Deterministic synthetic matrices for the BASELINE configurations whose SuiteSparse files are
 * not available offline: 27-point stencil with `dof` unknowns per grid node on
 * an nx*ny*nz grid, rows [row_begin,row_end) only (so that every rank of a distributed run
 * generates just its block; column indices stay global).  Symmetric pattern and values
 * v(i,j) = hash(min,max,seed) in [-1,1), diagonal 27*dof + u.  magnitude_decades > 0 spreads the
 * off-diagonal magnitudes log-uniformly over that many decades (HV15R-class, for -ap splits). */
int uspmv_gen_stencil27(int64_t nx, int64_t ny, int64_t nz, int dof, uint64_t seed, double magnitude_decades,
                        int64_t row_begin, int64_t row_end, uspmv_coo_t **out);
/* entries per row of that matrix for rows [row_begin,row_end) -- with uspmv_seg_from_row_counts the ranks of a distributed
 * run agree on the partition without anybody generating the whole matrix */
int uspmv_gen_stencil27_row_counts(int64_t nx, int64_t ny, int64_t nz, int dof, int64_t row_begin, int64_t row_end, int32_t *out);
/* Banded-random matrix of SURVEY.md 8(d) (HV15R-class: n = 2 017 169, 140 entries per row, band +-50 000, magnitudes
 * over 10 decades): the diagonal plus nnz_per_row - 1 hashed distinct columns in [i - band, i + band], general
 * pattern, columns ascending inside a row; the irregular counterpart of the stencil generator. */
int uspmv_gen_banded_random(int64_t n, int nnz_per_row, int64_t band, uint64_t seed, double magnitude_decades,
                            int64_t row_begin, int64_t row_end, uspmv_coo_t **out);
/* KKT-structured matrix of the nlpkkt class ([H A^T; A 0], PDE-constrained optimisation on an N^3 grid with boundary control;
 * n = 2 N^3 + 6 N^2 as in SuiteSparse nlpkkt200 / nlpkkt240): unknowns = states, multipliers, boundary controls; interior state
 * rows 28 entries, multiplier rows 25, control rows 5-7; a row's columns live in two index ranges N^3 apart.  Symmetric pattern
 * and values, columns ascending inside a row; rows [row_begin, row_end) with local row ids and global column ids. */
int uspmv_gen_kkt(int64_t N, uint64_t seed, int64_t row_begin, int64_t row_end, uspmv_coo_t **out);
int uspmv_gen_kkt_row_counts(int64_t N, int64_t row_begin, int64_t row_end, int32_t *out);

/* ------------------------------------------------------------------ L2: format conversion */
/* convert_to_scs (code/utilities.hpp:1842-2104; library twin code/interface.hpp:401-656).
 * fixed_permutation may be NULL (sigma-window sort by descending row length, std::sort tie order
 * reproduced) or an old->new row map of n_rows entries (:1911-1928).  dtype selects VT. */
int uspmv_convert_to_scs(const uspmv_coo_t *m, int64_t C, int64_t sigma, int dtype,
                         const int32_t *fixed_permutation, uspmv_scs_t **out);
/* meta[8] = C, sigma, n_rows, n_cols, n_rows_padded, n_chunks, n_elements, nnz */
int uspmv_scs_meta(const uspmv_scs_t *s, int64_t meta[8]);
int uspmv_scs_dtype(const uspmv_scs_t *s, int *dtype);
/* borrowed pointers (chunk_ptrs[n_chunks+1], chunk_lengths[n_chunks], col_idxs/values[n_elements],
 * old_to_new_idx/new_to_old_idx[n_rows]); any argument may be NULL */
int uspmv_scs_arrays(const uspmv_scs_t *s, const int32_t **chunk_ptrs, const int32_t **chunk_lengths,
                     const int32_t **col_idxs, const void **values, const int32_t **old_to_new_idx,
                     const int32_t **new_to_old_idx);
/* mutable view of col_idxs (halo set-up rewrites it in place like the reference does) */
int uspmv_scs_col_idxs_mut(uspmv_scs_t *s, int32_t **col_idxs);
/* permute_scs_cols (code/utilities.hpp:1802-1831): col = perm[col] where col < n_rows */
int uspmv_permute_scs_cols(uspmv_scs_t *s, const int32_t *perm);
void uspmv_scs_free(uspmv_scs_t *s);
/* apply_permutation (code/utilities.hpp:1768-1782): out[i] = in[perm[i]], host vectors */
int uspmv_apply_permutation(void *out, const void *in, const int32_t *perm, int64_t n, int dtype);
/* partition_precisions, ap[dp_sp] non-equilibrated branch (code/utilities.hpp:2899-2911):
 * |v| >= threshold_1 -> dp, else sp (values rounded to float), COO order kept. */
int uspmv_partition_precisions(const uspmv_coo_t *m, double threshold_1, uspmv_coo_t **dp, uspmv_coo_t **sp);

/* ------------------------------------------------------------------ L3: device kernels    */
int uspmv_device_count(int *count);
int uspmv_set_device(int device);
int uspmv_stream_synchronize(void *stream);

/* H2D staging of one SCS struct (assign_spmv_kernel_gpu_data, code/utilities.hpp:3721-3811). */
int uspmv_dmat_upload(const uspmv_scs_t *s, uspmv_dmat_t **out);
/* GPU-side conversion: the result of uspmv_convert_to_scs [+ uspmv_permute_scs_cols with the
 * struct's own old_to_new_idx when permute_cols != 0] + uspmv_dmat_upload, bit for bit, but only the
 * O(n_rows) layout (row lengths, sigma-window std::sort, chunk lengths / pointers, permutations) is
 * computed on the host; the O(nnz) scatter of code/utilities.hpp:2013-2036 runs on the device straight
 * from the uploaded COO arrays.  *layout is a host struct WITHOUT entries: meta data, chunk arrays and
 * the permutations are there, uspmv_scs_arrays returns NULL for col_idxs / values, and the functions that
 * need host entries (uspmv_dmat_upload / _optimize*, uspmv_halo_discover, ...) refuse it.  The COO entries
 * must be sorted by row (uspmv_read_mtx and the generators produce that order). */
int uspmv_convert_to_scs_device(const uspmv_coo_t *m, int64_t C, int64_t sigma, int dtype, const int32_t *fixed_permutation,
                                int permute_cols, uspmv_scs_t **layout, uspmv_dmat_t **out);
/* The same from DEVICE-resident COO arrays (d_I, d_J: int32, d_V: double; entries sorted by row, order inside a row = summation order):
 * nothing but O(n_rows) integers ever leaves the device.  Row populations by a run-boundary kernel (code/utilities.hpp:1901-1903), chunk
 * lengths by a per-chunk maximum and chunk_ptrs by a device scan (:1949-1966), the scatter as above (:2013-2036).  The sigma-window
 * ordering (:1930-1941) in one of two ways:
 *   USPMV_SORT_HOST          the reference's std::sort (same pair type, same comparator) on the host over the row COUNTS only -- 4 bytes
 *                            per row down, 4 up; every array bit-identical to uspmv_convert_to_scs incl. the tie order of the unstable sort;
 *   USPMV_SORT_DEVICE_STABLE a stable rank per window on the device (sigma <= 8192): rows of equal length keep their original order.
 *                            chunk_lengths, chunk_ptrs and y in ORIGINAL row order are bit-identical to the reference's; the order of
 *                            equal-length rows inside a window, hence old_to_new_idx and the row order of y_permuted, differs.
 * d_fixed_permutation (device, n_rows entries, may be NULL) as fixed_permutation above.  d_old_to_new / d_new_to_old (device, n_rows
 * int32 each, may be NULL) receive the struct's permutations for uspmv_apply_permutation_dev; *layout (may be NULL) a host struct
 * without entries as above.  Works on `stream`; returns when the handle is complete. */
/* diagnosis: device addresses of {tile-local-column indices, x-line lists, line pointers, index offsets, values, col_idxs, chunk_ptrs,
 * chunk_lengths} of the struct the SpMV kernels read (tools/placement_probe.py) */
int uspmv_dmat_plan_addresses(const uspmv_dmat_t *m, uint64_t addr[8]);
/* meta[4] = C, n_chunks, n_elements, dtype of a device handle (one without a host struct has nothing else to ask) */
int uspmv_dmat_meta(const uspmv_dmat_t *m, int64_t meta[4]);
enum { USPMV_SORT_HOST = 0, USPMV_SORT_DEVICE_STABLE = 1 };
int uspmv_convert_to_scs_device_from_arrays(const int32_t *d_I, const int32_t *d_J, const double *d_V, int64_t n_rows, int64_t n_cols, int64_t nnz,
                                            int64_t C, int64_t sigma, int dtype, const int32_t *d_fixed_permutation, int permute_cols,
                                            int sort_mode, void *stream, uspmv_scs_t **layout, int32_t *d_old_to_new, int32_t *d_new_to_old,
                                            uspmv_dmat_t **out);
/* copies of the device arrays of a handle (any pointer may be NULL): n_chunks+1, n_chunks, n_elements, n_elements */
int uspmv_dmat_download(const uspmv_dmat_t *m, int32_t *chunk_ptrs, int32_t *chunk_lengths, int32_t *col_idxs, void *values);
/* Wrap arrays that already live in HBM (owned by the caller, e.g. a framework allocator). */
int uspmv_dmat_wrap(int64_t C, int64_t n_chunks, int64_t n_elements, int dtype, const int32_t *d_chunk_ptrs,
                    const int32_t *d_chunk_lengths, const int32_t *d_col_idxs, const void *d_values,
                    uspmv_dmat_t **out);
void uspmv_dmat_free(uspmv_dmat_t *m);
/* Optional, MI355X-specific: derive a tile-local-column plan from the host struct the handle was made
 * from: per 256-row tile the list of 16-element x lines it touches plus 16-bit LDS-local
 * column indices.  uspmv_spmv then stages the x lines of a tile in LDS and streams 2-byte instead of
 * 4-byte indices (results unchanged, bit for bit).  Tiles touching more than max_lines lines (0 = default
 * 512) keep the gather path.  n_tiles / n_staged report the outcome (may be NULL).
 * With max_lines = 0 the planner has three fallbacks when the line plan leaves a tenth of the tiles or more unstaged: the same
 * plan over single x ELEMENTS (uspmv_dmat_plan_granularity = 1); that plan on rows dealt to the tiles by the matrix graph (a private copy of the values
 * in HBM, y stored through a row map -- for matrices whose rows and columns are numbered alike but not coherently); the column-window sweep for wide
 * irregular rows.  Every one of them walks each row's slots in the reference's order (scs_impl_cpu, code/kernels.hpp:218-258): same bits. */
int uspmv_dmat_optimize(uspmv_dmat_t *m, const uspmv_scs_t *s, int max_lines, int64_t *n_tiles, int64_t *n_staged);
/* Same for an ap[dp_sp] pair (structs with identical row layout): one shared line list per tile, 16-bit
 * indices for both structs; uspmv_spmv_ap then streams 10 + 6 instead of 12 + 8 bytes per non-zero. */
int uspmv_dmat_optimize_ap(uspmv_dmat_t *dp, uspmv_dmat_t *sp, const uspmv_scs_t *s_dp, const uspmv_scs_t *s_sp,
                           int max_lines, int64_t *n_tiles, int64_t *n_staged);
/* SpMMV counterpart of uspmv_dmat_optimize (no reference counterpart): plan for block vectors of
 * block_vec_size columns -- per 64-row tile the X rows it touches are staged in LDS once and the
 * kernel reads them with 2-byte local indices.  Used by uspmv_spmmv for b*sizeof(VT) in {16,32,64,128}
 * whose tiles fit, C = 32 or 64; other widths and chunk heights keep the gather kernels
 * (*n_staged = 0).  Results are bit-identical with and without the plan. */
int uspmv_dmat_optimize_block(uspmv_dmat_t *m, const uspmv_scs_t *s, int block_vec_size, int64_t *n_tiles, int64_t *n_staged);

/* Column-window sweep plan (no reference counterpart): for matrices with wide, irregular rows, where a
 * 256-row tile touches more x lines than LDS holds.  x is cut into windows of 2^wlog elements; a tile of tile_rows
 * (256 | 512 | 1024) rows whose rows all visit the windows in non-decreasing slot order (column-sorted rows do) is
 * processed by one workgroup that stages window after window in LDS and runs every row's entries of the staged
 * window -- same slot-ordered FMA chain, bit-identical y.  The plan holds a private, padding-free copy of the entries
 * (sizeof(VT) + 2 bytes per non-zero).  Tiles that do not qualify keep the gather kernel inside the same uspmv_spmv call.
 * wlog / tile_rows 0 = defaults (64 KiB windows; 2048 rows, 1024 for an ap pair).  uspmv_dmat_optimize[_ap] tries this by itself when the
 * tile-local-column plan stages less than half of the tiles.  n_tiles / n_sweep report the outcome (may be NULL). */
int uspmv_dmat_optimize_sweep(uspmv_dmat_t *m, const uspmv_scs_t *s, int wlog, int tile_rows, int64_t *n_tiles, int64_t *n_sweep);
int uspmv_dmat_optimize_sweep_ap(uspmv_dmat_t *dp, uspmv_dmat_t *sp, const uspmv_scs_t *s_dp, const uspmv_scs_t *s_sp, int wlog,
                                 int tile_rows, int64_t *n_tiles, int64_t *n_sweep);
/* Which single-vector plan uspmv_spmv / uspmv_spmv_ap will use: kind 0 none (gather kernel), 1 tile-local-column, 2 column-window
 * sweep; tiles of that plan and how many of them it covers (any pointer may be NULL). */
/* The column-window sweep plan built on the DEVICE from the handle's own arrays (csrc/sweep_plan_kernels.hip): what
 * uspmv_dmat_optimize_device[_ap] falls through to when the tile-local-column plan stages fewer than half of the tiles, so that
 * handles without a host struct (uspmv_dmat_wrap, the launchers of include/uspmv_launchers.hpp) reach scs_spmv_sweep.  sp: the sp
 * part of an ap[dp_sp] pair or NULL.  The arrays equal those of uspmv_dmat_optimize_sweep[_ap] bit for bit. */
int uspmv_dmat_optimize_sweep_device(uspmv_dmat_t *m, uspmv_dmat_t *sp, int wlog, int tile_rows, int64_t *n_tiles, int64_t *n_sweep);
/* FNV-1a digests of the sweep plan's device arrays and meta[8] = present, rows per tile, log2 window, sweep tiles, tiles, chunks
 * left to the gather kernel, elements of the dp / sp stream (tests) */
int uspmv_dmat_sweep_plan_digest(const uspmv_dmat_t *m, uint64_t digest[16], int64_t meta[8]);
int uspmv_dmat_plan_info(const uspmv_dmat_t *m, int *kind, int64_t *n_tiles, int64_t *n_planned);
/* block-vector plans of the handle: meta[10] = one-list-per-tile plan present, phased plan present, line plan present (column-major
 * block vectors staged by 128-byte lines, no re-layout pass), tiles, phases of the phased plan, phases of the line plan, X rows the
 * line plan stages, one-byte indices, index part built on the device, most X rows of a phase */
int uspmv_dmat_block_plan_info(const uspmv_dmat_t *m, int64_t meta[10]);
/* X rows the phased block plan stages per product (sum of its phases' lists; 0 without such a plan) */
int uspmv_dmat_block_plan_staged(const uspmv_dmat_t *m, int64_t *rows_staged);
/* x elements per entry of the tile-local-column plan's lists: 16 (128-byte lines, the default), 1 (single elements: the plan uspmv_dmat_optimize falls
 * to when a tile's columns are scattered over too many lines -- row numberings that are only locally coherent), 0 without such a plan.  Same kernel
 * arithmetic as scs_impl_cpu (code/kernels.hpp:218-258) either way. */
int uspmv_dmat_plan_granularity(const uspmv_dmat_t *m, int *elements_per_list_entry);
/* 1 when that plan runs on rows dealt to its tiles by the matrix graph (private value copy in HBM, y stored through a row map), else 0 */
int uspmv_dmat_plan_rows_dealt(const uspmv_dmat_t *m, int *dealt);
/* the phased block plan laid out as a flat schedule for persistent workgroups (tuning "spmmv_stream" > 0 at plan time; the streaming form of
 * block_spmv_omp_scs_general, code/kernels.hpp:306-398): meta[2] = workgroups of the launch (0: no schedule on the handle), phase descriptors */
int uspmv_dmat_stream_info(const uspmv_dmat_t *m, int64_t meta[2]);
/* FNV-1a digests of the phased block plan's device arrays: phase pointers, first groups, list pointers, X-row lists, index offsets,
 * local indices, the group-major values, the row map (tests: a plan built on the device equals the host planner's) */
int uspmv_dmat_block_plan_digest(const uspmv_dmat_t *m, uint64_t digest[8]);

/* Block vectors with 64-byte X rows (dp b = 8, sp b = 16): the column-window sweep plan for block vectors.  A tile of 1 024 - 4 096 rows
 * walks the windows of 2^wlog X rows its rows touch, each staged in LDS once per tile (the phased plan above stages every X row once per
 * slot range: 11.9 against ~5 rows per matrix row on a 3-dof 27-point stencil), a lane owning whole rows; column-major X and Y are taken
 * as they are, without a re-layout pass.  Installed only when every tile qualifies (rows column-sorted at window granularity); uspmv_spmmv
 * then prefers it.  wlog / tile_rows 0 = defaults.  *n_sweep < *n_tiles: not installed.  Same bits as every other path. */
int uspmv_dmat_optimize_block_sweep(uspmv_dmat_t *m, const uspmv_scs_t *s, int block_vec_size, int wlog, int tile_rows, int64_t *n_tiles, int64_t *n_sweep);

/* The block plan for a handle without a host struct (uspmv_dmat_wrap): the device arrays are copied to the host once and the plan is
 * built there.  No permutation is known then: ties of the sigma sort are ordered by first column instead of by original row. */
int uspmv_dmat_optimize_block_device(uspmv_dmat_t *m, int block_vec_size, int64_t *n_tiles, int64_t *n_staged);

/* The same plan built ON THE DEVICE from the handle's own arrays: for handles without a host struct
 * (uspmv_dmat_wrap around the reference's cudaMalloc'ed arrays, uspmv_convert_to_scs_device).  Chunk heights that
 * divide 256; tiles whose line range exceeds 65 536 lines stay on the gather path (the host planner may still
 * stage those); otherwise the plan is identical to uspmv_dmat_optimize's.  Narrow chunks (C in {1,2,4,8,16}, incl. crs) get the
 * same internal C = 32 re-chunking as in uspmv_dmat_optimize, copied on the device. */
int uspmv_dmat_optimize_device(uspmv_dmat_t *m, int max_lines, int64_t *n_tiles, int64_t *n_staged);
/* The shared plan of an ap[dp_sp] pair (uspmv_dmat_optimize_ap) built on the device from the two handles' own arrays. */
int uspmv_dmat_optimize_device_ap(uspmv_dmat_t *dp, uspmv_dmat_t *sp, int max_lines, int64_t *n_tiles, int64_t *n_staged);
/* host copies of a handle's plan (tests): meta = {n_tiles, n_lines_total, n_col16, max_lines_used}; call with NULL
 * arrays first to size them */
int uspmv_dmat_plan_download(const uspmv_dmat_t *m, int64_t meta[4], int32_t *tile_line_ptr, int32_t *tile_lines,
                             uint32_t *c16_ptrs, uint16_t *col16);

/* Rows per tile of the plan (256 | 512 | 1024; 0 = no plan).  A tile covers tile_rows/C consecutive chunks. */
int uspmv_dmat_tile_rows(const uspmv_dmat_t *m, int *tile_rows);
/* Bits per tile-local column index the plan's kernel streams: 16, or 12 where the plan packed them (tiles of at most 256 lines; kept when
 * the mean row length is >= 8; tuning key "tlc_idx12" 0 switches it off, 2 keeps it wherever it can be built); 0 = no plan */
int uspmv_dmat_index_bits(const uspmv_dmat_t *m, int *bits);
/* uspmv_spmv over a subset of tiles (d_tile_ids[n_ids]) of a handle with a plan: the interior /
 * boundary split of the halo-overlap scheme at tile granularity.  Entries < 0 are skipped (a list another kernel switches on or off). */
int uspmv_spmv_tiles(const uspmv_dmat_t *A, const int32_t *d_tile_ids, int64_t n_ids, const void *d_x, void *d_y,
                     void *stream);
/* Mark a C = 1 struct as "crs" (uspmv <mtx> crs): uspmv_spmv then uses the CRS kernel (several lanes
 * per row, twin of spmv_omp_csr) instead of the generic SELL kernel (twin of spmv_omp_scs). */
int uspmv_dmat_set_crs(uspmv_dmat_t *m, int on);

/* y = A x.  Replaces SpmvKernel::execute_one_prec -> spmv_gpu_scs_adv_launcher /
 * spmv_gpu_scs_launcher / spmv_gpu_csr_launcher (code/classes_structs.hpp:997-1035,
 * code/kernels.hpp:579-775), i.e. the GPU twins of spmv_omp_scs_adv / spmv_omp_scs /
 * spmv_omp_csr.  d_x: padded_vec_size elements; d_y: n_rows_padded elements written. */
int uspmv_spmv(const uspmv_dmat_t *A, const void *d_x, void *d_y, void *stream);
/* Same over a subset of chunks (d_chunk_ids[n_ids], ascending): interior / boundary split used to
 * overlap the halo exchange with the kernel (absent in the reference, code/main.cpp:464-468). */
int uspmv_spmv_chunks(const uspmv_dmat_t *A, const int32_t *d_chunk_ids, int64_t n_ids, const void *d_x,
                      void *d_y, void *stream);
/* Y = A X, b right-hand sides.  Replaces block_spmv_gpu_scs_{adv,general}_launcher /
 * block_spmv_gpu_csr_launcher, which the reference only stubs on GPU (code/kernels.hpp:777-844);
 * semantics = block_spmv_omp_scs_general (code/kernels.hpp:306-398).  ld = vec_length of the
 * colwise layout (ignored for rowwise). */
int uspmv_spmmv(const uspmv_dmat_t *A, const void *d_X, void *d_Y, int b, int64_t ld, int layout, void *stream);
/* Column-major block vectors whose X does NOT change between calls -- the reference's bench loop multiplies the same X in every
 * iteration (code/main.cpp:458-519): re-lay X out into the handle's row-major workspace once, now.  Until uspmv_spmmv_x_release, every
 * uspmv_spmmv(m, d_X, ..., b, ld, USPMV_COLWISE) with THIS pointer, b and ld skips its re-layout pass (another X, b or ld takes the
 * per-call re-layout as before and overwrites the workspace: call again afterwards).  The caller promises that the CONTENTS of X are
 * unchanged since this call; a caller that writes X (solve mode: x <- y) must call again or release.  Widths without a re-layout pass
 * (b not in {2, 4, 8, 16}) return USPMV_OK and skip nothing. */
int uspmv_spmmv_x_prepared(const uspmv_dmat_t *m, const void *d_X, int block_vec_size, int64_t ld, void *stream);
int uspmv_spmmv_x_release(const uspmv_dmat_t *m);
/* Adaptive precision dp+sp.  Replaces execute_two_prec -> spmv_gpu_ap_scs_adv_launcher
 * (code/classes_structs.hpp:1037-1075, code/ap_kernels.hpp:821-953); numerics follow the CPU
 * kernel scs_ap_impl_cpu (code/ap_kernels.hpp:24-82): both parts accumulated in double from the
 * double x, y = dp_sum + sp_sum.  dp and sp must share C and n_chunks. */
int uspmv_spmv_ap(const uspmv_dmat_t *dp, const uspmv_dmat_t *sp, const double *d_x, double *d_y, void *stream);
/* Generic-C variant of the reference, spmv_omp_scs_ap / spmv_gpu_ap_scs (code/ap_kernels.hpp:562-634,
 * :721-816; selected for C outside {2,4,...,128}, code/classes_structs.hpp:630-636): the sp part
 * multiplies with the FLOAT copy of x (d_x_sp), the float product being rounded before it is added
 * to the double accumulator. */
int uspmv_spmv_ap_generic(const uspmv_dmat_t *dp, const uspmv_dmat_t *sp, const double *d_x, const float *d_x_sp,
                          double *d_y, void *stream);

/* Raw-array forms with the argument lists of the library kernels of code/interface.hpp
 * (uspmv_scs_gpu :1766-1793, uspmv_scs_c_gpu :1835-1867, uspmv_csr_gpu :1741-1760). */
int uspmv_scs_gpu_f64(int64_t C, int64_t n_chunks, const int32_t *d_chunk_ptrs, const int32_t *d_chunk_lengths,
                      const int32_t *d_col_idxs, const double *d_values, const double *d_x, double *d_y,
                      void *stream);
int uspmv_scs_gpu_f32(int64_t C, int64_t n_chunks, const int32_t *d_chunk_ptrs, const int32_t *d_chunk_lengths,
                      const int32_t *d_col_idxs, const float *d_values, const float *d_x, float *d_y,
                      void *stream);
int uspmv_csr_gpu_f64(int64_t n_rows, const int32_t *d_row_ptrs, const int32_t *d_col_idxs,
                      const double *d_values, const double *d_x, double *d_y, void *stream);
int uspmv_csr_gpu_f32(int64_t n_rows, const int32_t *d_row_ptrs, const int32_t *d_col_idxs,
                      const float *d_values, const float *d_x, float *d_y, void *stream);

/* Read one scalar through a pointer that may point to host OR device memory (the reference's GPU build keeps C and n_chunks
 * in device memory, code/utilities.hpp:3803-3811; include/uspmv_launchers.hpp reads them once per set of arrays). */
int uspmv_peek_i64(const void *p, int64_t *out);
int uspmv_peek_i32(const void *p, int32_t *out);

/* Device apply_permutation: d_out[i] = d_in[d_perm[i]] (code/utilities.hpp:1768-1782). */
int uspmv_apply_permutation_dev(void *d_out, const void *d_in, const int32_t *d_perm, int64_t n, int dtype,
                                void *stream);

/* Kernel-variant selection for A/B measurements (DESIGN.md "variants").  key/value pairs:
 *   "unroll" 1|2|4|8, "nontemporal" 0|1, "block" 64|128|256|512|1024,
 *   "xcd_remap" 0 (hardware order) | 1 (one contiguous eighth of the grid per XCD) | G >= 2 (groups
 *   of G consecutive workgroups per XCD), "spmv_variant" 0 (lane per row, bit-exact) | 1 (two lanes
 *   per row, C=32 only), "csr_lanes" 0 (auto) | 1..64 lanes per row of the CRS kernel,
 *   "tlc" 1|0 use the tile-local-column kernel when the handle has a plan, "tail_batch" 0|1,
 *   "tlc_tile_rows" 0|256|512|1024 rows (= threads) per tile of the NEXT uspmv_dmat_optimize[_ap|_device|_device_ap]
 *   (0 = 256 for one struct, 512 for an ap[dp_sp] pair), "tlc_auto_tile" 1|0: with "tlc_tile_rows" 0 and one struct, take
 *   1024- or 512-row tiles instead when the largest 256-row tile needs more than 250 x lines and the larger tiles stage >= 99 %
 *   of the tiles (same y bits; measured in profiles/r03/tile_rows_sweep.txt), "tlc_measure_tile" 1|0: structs of >= 2^20 padded rows get
 *   the tile size that MEASURES fastest (plans for 256 / 512 / 1024 rows built on the device, three timed launches each, once per matrix
 *   shape and process),
 *   "spmmv_variant" 0 (auto) | 1 (generic) | 2 (row-major, transposing X phase) | 3 (row-major, lane per row)
 *                   | 4 (single-wave block-plan tiles) | 5 (3 over the plan's tie-re-ordered copy) | 6 (four lanes per row, 64-byte rows)
 *                   | 8 (four lanes per row over the phased plan; what auto picks for 64-byte rows),
 *   "spmmv_phased" 1|0 and "spmmv_phase_rows" 256|512: NEXT uspmv_dmat_optimize_block with 64-byte rows also builds the phased plan,
 *   "spmmv_xcol" 0|1: phased kernel on column-major X behind a re-layout pass (0) or assembling its X rows from the caller's vector itself (1),
 *   "spmmv_swizzle" 0|1 bank-swizzled LDS rows in the block-plan kernel,
 *   "spmmv_reorder" 1|2|0 NEXT uspmv_dmat_optimize_block, row order of the plan's private copy of the entries: 1 = the sigma sort's tie
 *   scrambling undone, 2 = rows re-dealt to the 64-row tiles as breadth-first balls of the matrix graph (fewer X rows per tile, but
 *   scattered y rows: measured slower on config 3, kept as an option), 0 = the caller's order;
 *   "spmmv_idx8" 1|0 NEXT uspmv_dmat_optimize_block: one-byte phase-local indices when no phase lists more than 256 X rows,
 *   "spmmv_list_plan" 0|1 NEXT uspmv_dmat_optimize_block: also build the one-list-per-tile plan of variants 4-6 (and its column-major copy
 *   of the entries) when the phased kernel can take the matrix,
 *   "sweep" 1|0 use a handle's column-window sweep plan, "sweep_nbuf" 1|2 LDS buffers, "sweep_unroll" 2|4|8, "sweep_remap" tiles per XCD group,
 *   "sweep_threads" 0|256|512|1024 threads per sweep workgroup (0 = min(tile rows, 1024); fewer threads = more rows per lane, at most 4),
 *   "sweep_wlog" / "sweep_tile_rows" / "sweep_max_stage" defaults of the NEXT sweep plan (window = 2^wlog elements; rows per tile;
 *   largest staging cost in bytes per non-zero for a tile to qualify, 0 = 24),
 *   "raw_plan_cache" 0|1 uspmv_scs_gpu_f64/f32 keep a device-built plan per set of array addresses (the caller
 *   promises not to put another matrix behind the same pointers; uspmv_raw_plan_cache_clear() otherwise),
 *   "spmmv_tile_rows" 0 (auto) | 32 | 64 rows per tile and "spmmv_lds_kb" 0 (= 80) | LDS KiB per tile of the NEXT
 *   uspmv_dmat_optimize_block,
 *   "spmmv_prefetch" 1|0 (lane-per-row kernel: next batch of matrix entries requested behind the X rows),
 *   "spmmv_unroll" 0 (auto) | 1|2|4|8 slots per batch,
 *   "ablate" 0 | 1 | 2 (measurement only: gathers collapsed / removed, results are wrong). */
void uspmv_raw_plan_cache_clear(void);
int uspmv_set_tuning(const char *key, int value);
int uspmv_get_tuning(const char *key, int *value);

/* ------------------------------------------------------------------ L4: halo exchange     */
/* seg_work_sharing_arr (code/mpi_funcs.hpp:424-622), seg-rows and seg-nnz: wsa[P+1]. */
int uspmv_seg_work_sharing_arr(const uspmv_coo_t *total, int seg_method, int P, int32_t *wsa);
/* -seg_metis (code/mpi_funcs.hpp:494-598) without METIS: a part id per row from the built-in partitioner (breadth-first level sets
 * from a pseudo-peripheral vertex, P equal pieces of that ordering, boundary refinement; vertex counts balanced like
 * METIS_PartGraphKway without weights) or from a file with one part id per line (the output format of gpmetis) ... */
int uspmv_graph_partition(const uspmv_coo_t *total, int P, int32_t *part /* n_rows */);
int uspmv_read_partition(const char *path, int64_t n_rows, int P, int32_t *part);
/* ... and the reference's post-processing of it (:529-598): rows stable-sorted by part (new row r = old row perm[r]), the matrix
 * permuted symmetrically (entry order inside a row kept), wsa[P+1] from the part sizes.  perm may be NULL. */
int uspmv_coo_apply_partition(const uspmv_coo_t *total, int P, const int32_t *part, uspmv_coo_t **permuted, int32_t *wsa, int32_t *perm);
/* the same rule from per-row entry counts alone (bit-identical wsa to the call above on the row-sorted COO) */
int uspmv_seg_from_row_counts(const int32_t *row_nnz, int64_t n_rows, int seg_method, int P, int32_t *wsa);
/* seg_mtx_struct + localize_row_idx (code/mpi_funcs.hpp:636-674, :862-877): rows
 * [wsa[rank], wsa[rank+1]) with process-local row ids and GLOBAL column ids. */
int uspmv_seg_local_coo(const uspmv_coo_t *total, const int32_t *wsa, int rank, uspmv_coo_t **out);
/* collect_local_needed_heri (code/mpi_funcs.hpp:242-415): discovers remote columns of the local
 * SCS struct (scanned in storage order, padding entries included), rewrites col_idxs to
 * local + halo numbering, and records per owner which of ITS local rows this rank needs. */
int uspmv_halo_discover(uspmv_scs_t *local_scs, const int32_t *wsa, int rank, int P, uspmv_halo_t **out);
/* n_halo = recv_counts_cumsum.back(); recv_counts_cumsum[P+1]; recv_idxs grouped by owner rank
 * ascending with recv_counts[P] entries each (borrowed pointers). */
int uspmv_halo_meta(const uspmv_halo_t *h, int64_t *n_halo, const int32_t **recv_counts_cumsum,
                    const int32_t **recv_idxs, const int32_t **recv_counts);
void uspmv_halo_free(uspmv_halo_t *h);
/* Chunk ids whose columns are all local (< n_local) and the rest (touch the halo region). Arrays
 * are malloc'ed; release with uspmv_free. */
int uspmv_scs_split_chunks(const uspmv_scs_t *s, int64_t n_local, int32_t **interior, int64_t *n_interior,
                           int32_t **boundary, int64_t *n_boundary);
/* The same split in three classes, classes[n_chunks]: 0 = no halo column; 1 = halo columns only through the reference's padding -- entries
 * that are +0.0 on the ONE column *pad_col (value 0, column 0: code/utilities.hpp:1991-2002; a halo column on every rank but the first,
 * code/mpi_funcs.hpp:279-306); 2 = other halo references.  *pad_col = -1 when no chunk is of class 1.  (What "pad_split" of
 * uspmv_dist_set_option is built on.) */
int uspmv_scs_chunk_classes(const uspmv_scs_t *s, int64_t n_local, uint8_t *classes, int32_t *pad_col);
void uspmv_free(void *p);
/* Send-buffer gather for ALL neighbours in one launch over the concatenated index list:
 * d_send[i] = d_x[d_perm[d_send_idxs[i]] + block_offset].  Replaces pack_send_buf /
 * pack_d_send_buf (code/classes_structs.hpp:786-855, code/kernels.hpp:554-577: one launch +
 * device sync per neighbour). */
int uspmv_pack_send_buf(const void *d_x, const int32_t *d_perm, const int32_t *d_send_idxs, int64_t n,
                        int64_t block_offset, void *d_send, int dtype, void *stream);

/* ------------------------------------------------------------------ L4a: set-up transport, host communicator, exchange plan */
/* collect_comm_info (code/mpi_funcs.hpp:1061-1124) needs two exchanges between the ranks: every rank tells every owner HOW MANY
 * of the owner's rows it needs (the reference all-gathers the cumsums, :190-196) and WHICH (the MPI_INT index all-to-all,
 * :143-171).  Both run over this small transport, so the SAME C++ set-up is driven by RCCL (device staging, the default of
 * uspmv_dist_create), by the host communicator below (real processes, no GPU: tests/test_dist_setup_mp.py) and in loopback.
 * All buffers are HOST memory; offsets are in bytes with size+1 entries:
 *   alltoallv: recv[recv_off[q] .. recv_off[q+1]) <- rank q's send[send_off_q[me] .. send_off_q[me+1])
 *   allgather: recv[q*bytes .. (q+1)*bytes) <- rank q's send[0 .. bytes) */
typedef struct uspmv_transport {
    void *ctx;
    int rank, size;
    int (*alltoallv)(void *ctx, const void *send, const int64_t *send_off, void *recv, const int64_t *recv_off);
    int (*allgather)(void *ctx, const void *send, void *recv, int64_t bytes_per_rank);
    int (*barrier)(void *ctx);
} uspmv_transport_t;

/* Host communicator of ONE node: the ranks of a job meet in a memory-mapped segment (<USPMV_HC_DIR or /dev/shm>/uspmv_hc_<job>,
 * published by rank 0 with an atomic rename and unlinked as soon as every rank has attached; leftovers of a crashed job are
 * recognised by their dead creator and ignored).  Stands where the reference has MPI_Bcast / MPI_Allgather / the index exchange
 * during set-up (code/mpi_funcs.hpp:143-171, :190-196, :732-736); every wait has a deadline (timeout_s, <= 0: 300 s) and fails
 * on all ranks with USPMV_ERR_COMM when a peer dies. */
typedef struct uspmv_hostcomm uspmv_hostcomm_t;
int uspmv_hostcomm_create(const char *job, int rank, int size, double timeout_s, uspmv_hostcomm_t **out);
int uspmv_hostcomm_info(const uspmv_hostcomm_t *h, int *rank, int *size, uint64_t *nonce /* random per segment, same on all ranks */);
int uspmv_hostcomm_barrier(uspmv_hostcomm_t *h);
int uspmv_hostcomm_abort(uspmv_hostcomm_t *h);   /* tell the peers this rank cannot go on: their next wait fails at once */
int uspmv_hostcomm_bcast(uspmv_hostcomm_t *h, void *buf, int64_t bytes, int root);
int uspmv_hostcomm_allgather(uspmv_hostcomm_t *h, const void *send, void *recv, int64_t bytes_per_rank);
int uspmv_hostcomm_alltoallv(uspmv_hostcomm_t *h, const void *send, const int64_t *send_off, void *recv, const int64_t *recv_off);
int uspmv_hostcomm_allreduce_max_f64(uspmv_hostcomm_t *h, double *value);
int uspmv_hostcomm_transport(uspmv_hostcomm_t *h, uspmv_transport_t *t);   /* the transport view (h must outlive its users) */
void uspmv_hostcomm_free(uspmv_hostcomm_t *h);

/* organize_cumsums + collect_comm_idxs (code/mpi_funcs.hpp:117-232): from what this rank NEEDS from whom (its halo description)
 * to what it must SEND to whom -- collective over the transport, no GPU involved.  send_off[P+1] / recv_off[P+1] count elements;
 * send_idxs[send_off[p] .. send_off[p+1]) are THIS rank's local rows (original order) rank p asked for, i.e. the reference's
 * comm_send_idxs[p].  Every id is checked against [0, n_local): a peer asking for a row this rank does not own is an error on
 * this rank (USPMV_ERR_INVALID), never an out-of-bounds gather. */
typedef struct uspmv_comm_plan uspmv_comm_plan_t;
int uspmv_comm_plan_create(const uspmv_transport_t *t, const uspmv_halo_t *halo, uspmv_comm_plan_t **out);
int uspmv_comm_plan_meta(const uspmv_comm_plan_t *p, int64_t *n_send, const int64_t **send_off, const int32_t **send_idxs,
                         const int64_t **recv_off);
void uspmv_comm_plan_free(uspmv_comm_plan_t *p);

/* ------------------------------------------------------------------ L4b: the distributed step on RCCL */
/* One process per GPU.  Replaces the per-iteration MPI flow of the reference -- init_local_structs (code/main.cpp:1075-1334),
 * collect_comm_info (code/mpi_funcs.hpp:1061-1124), init/finalize_halo_exchange (code/classes_structs.hpp:857-995) and the
 * exchange-then-kernel iteration of bench_spmv (code/main.cpp:458-474) -- by a C++ object that owns an RCCL communicator:
 * per SpMV one pack kernel, one grouped ncclSend/ncclRecv landing in the tail of x on a side stream, the interior tiles /
 * chunks meanwhile, the boundary ones after it; uspmv_dist_run replays the whole step from ONE captured hipGraph. */
typedef struct uspmv_dist uspmv_dist_t;
#define USPMV_COMM_ID_BYTES 128
/* ncclGetUniqueId on the root rank; the bytes reach the other ranks by any side channel (a file, torch.distributed, MPI_Bcast) */
int uspmv_comm_unique_id(void *id128);
/* Partition block `rank` of P on communicator rank comm_rank of comm_size.  comm_size == P (one rank per block), or comm_size == 1
 * with P > 1: LOOPBACK -- this process plays block `rank` and every neighbour is itself (RCCL self send/recv of the ids it asked
 * for); with an x that repeats with the block height that reproduces the multi-rank result of the block's rows on one GPU.
 * A / halo / the id lists are the caller's (A must outlive the object): halo from uspmv_halo_discover, old_to_new_idx the row
 * permutation of the local struct, interior / boundary ids from uspmv_scs_split_chunks (chunk ids) or tile ids of A's plan. */
int uspmv_dist_create(const void *comm_id, int comm_rank, int comm_size, int rank, int P, uspmv_dmat_t *A, const uspmv_halo_t *halo,
                      const int32_t *old_to_new_idx, const int32_t *interior_ids, int64_t n_interior, const int32_t *boundary_ids,
                      int64_t n_boundary, int ids_are_tiles, uspmv_dist_t **out);
/* The whole of init_local_structs for one block: convert_to_scs -> halo discovery -> column permutation -> upload -> plan ->
 * interior / boundary split -> uspmv_dist_create.  `local` = rows [wsa[rank], wsa[rank+1]) with local row ids and GLOBAL
 * column ids (uspmv_seg_local_coo or a generator's row range).  The object owns everything it built. */
int uspmv_dist_create_from_coo(const void *comm_id, int comm_rank, int comm_size, int rank, int P, const uspmv_coo_t *local,
                               const int32_t *wsa, int64_t C, int64_t sigma, int dtype, int tlc, uspmv_dist_t **out);
/* The same two with options.  transport: who carries the set-up exchanges (NULL = the RCCL communicator, device-staged; loopback
 * uses an identity transport).  exchange: USPMV_EXCHANGE_RCCL (default) or USPMV_EXCHANGE_HOST -- the per-step halo exchange
 * staged through host memory over `transport` (pack kernel -> D2H -> all-to-all-v -> H2D into the tail of x), which lets P real
 * processes share ONE GPU: it is how a single-GPU box runs the C++ step with unequal seg-nnz blocks and asymmetric send / recv
 * lists (tests/test_dist_native_gpu.py).  With USPMV_EXCHANGE_HOST no RCCL communicator is created (comm_id may be NULL);
 * transport->size must equal P and transport->rank the block.  The transport must outlive the object. */
typedef enum { USPMV_EXCHANGE_RCCL = 0, USPMV_EXCHANGE_HOST = 1 } uspmv_exchange;
typedef struct uspmv_dist_options {
    const uspmv_transport_t *transport;
    int exchange;
} uspmv_dist_options_t;
int uspmv_dist_create_ex(const void *comm_id, int comm_rank, int comm_size, int rank, int P, uspmv_dmat_t *A, const uspmv_halo_t *halo,
                         const int32_t *old_to_new_idx, const int32_t *interior_ids, int64_t n_interior, const int32_t *boundary_ids,
                         int64_t n_boundary, int ids_are_tiles, const uspmv_dist_options_t *opt, uspmv_dist_t **out);
int uspmv_dist_create_from_coo_ex(const void *comm_id, int comm_rank, int comm_size, int rank, int P, const uspmv_coo_t *local,
                                  const int32_t *wsa, int64_t C, int64_t sigma, int dtype, int tlc, const uspmv_dist_options_t *opt,
                                  uspmv_dist_t **out);
/* the exchange plan of the object (borrowed): n_send, send_off[P+1], send_idxs[n_send], recv_off[P+1] as in uspmv_comm_plan_meta */
int uspmv_dist_comm_plan(const uspmv_dist_t *d, int64_t *n_send, const int64_t **send_off, const int32_t **send_idxs,
                         const int64_t **recv_off);
/* Options by name: "overlap" 1|0, "no_pack" 0|1, "ba_synch" 0|1 (a stream-ordered one-element all-reduce after every step: the
 * MPI_Barrier the reference issues per iteration by default, code/main.cpp:467, :417; part of the captured graph),
 * "capture_mode" 0 global | 1 thread-local | 2 relaxed (hipStreamCaptureMode of uspmv_dist_run's capture),
 * "diag_skip_exchange" 0|1 (diagnosis only: the step skips the RCCL group, results are wrong),
 * "fused_step" 0|1 (default 0; tile lists: the step's tiles in ONE launch -- interior and padding tiles first, the boundary tiles at the end of
 *   the grid, each of which looks once whether the exchange has completed and otherwise defers itself to a small second launch behind
 *   the exchange; nothing spins.  0: interior launch, exchange, boundary launch),
 * "pad_split" 0|1 (default 0; 1: padding tiles run with the interior ones, see uspmv_dist_pad_info; 0: with the boundary tiles),
 * "autotune_all" 0|1 (default 0: uspmv_dist_autotune times overlap | plain; 1: the pad / fused arrangements as well),
 * "diag_spmmv_part" 0|1|2 (diagnosis only: the two-part block-vector step runs both parts, its interior part, its boundary part),
 * "block_plan" b (block vectors: build the phased block plan for b columns on the rank's matrix, 0 drops it; see uspmv_dist_spmmv). */
int uspmv_dist_set_option(uspmv_dist_t *d, const char *key, int value);
/* How the single-vector step is arranged around the exchange: interior tiles during the exchange and boundary tiles after it | the
 * exchange, then the whole matrix (the reference's order) | overlap with the padding-only tiles in front of the exchange ("pad_split") |
 * that in one launch ("fused_step", eager steps only).  uspmv_dist_autotune times the first two (all four with "autotune_all" 1: the last
 * two have never run next to an exchange between different GPUs) for 2 x (10 + 40) steps on the machine at hand --
 * COLLECTIVE: every rank calls it; the ranks agree on the slowest rank's clock -- sets the options of the fastest and returns it with
 * the candidates' ms per step (0: not tried; negative: fastest but refused by the self-check).  With `local` and `wsa` (as for
 * uspmv_dist_check, which it then runs) a pad / fused winner must pass the bitwise self-check on every rank or the faster of the first
 * two takes over.  d_x keeps its local part (its halo tail and d_y are overwritten).  All arrangements give the same bits. */
typedef enum { USPMV_STEP_OVERLAP = 0, USPMV_STEP_PLAIN = 1, USPMV_STEP_PAD = 2, USPMV_STEP_FUSED = 3 } uspmv_step_form;
int uspmv_dist_autotune(uspmv_dist_t *d, void *d_x, void *d_y, int use_graph, const uspmv_coo_t *local, const int32_t *wsa, void *stream,
                        int *form, double ms[4]);
/* Self-check of the whole distributed path (partition, halo discovery, exchange plan, exchange, kernels) on the object's own
 * matrix: one step with x_global[j] = 1 + 1e-3 * (j mod 1000) -- every halo element differs from its neighbours, unlike the
 * benchmark's constant 5.0 -- and y of the local rows compared BITWISE with the rows' entry-ordered FMA chains evaluated on the
 * host straight from `local` (global column ids; the role of the reference's MKL validation, code/write_results.hpp:442-556).
 * mismatches = rows that differ on this rank, checksum = sum of the local y in original order (both optional). */
int uspmv_dist_check(uspmv_dist_t *d, const uspmv_coo_t *local, const int32_t *wsa, void *d_x, void *d_y, int use_graph,
                     void *stream, int64_t *mismatches, double *checksum);
/* The host half of that check alone (no GPU): y_ref[i] (n_rows of `local`, dtype) = the entry-ordered FMA chain of local row i over
 * x_global[j] = 1 + 1e-3 * (j mod 1000), for steps that do not run on a uspmv_dist object (bench.py's torch.distributed twin). */
int uspmv_dist_check_reference(const uspmv_coo_t *local, const int32_t *wsa, int rank, int P, int dtype, void *y_ref);
/* HIP / RCCL versions this library was COMPILED against and the ones it RUNS on in this process:
 * v[0] HIP_VERSION (build), v[1] hipRuntimeGetVersion, v[2] NCCL_VERSION_CODE (build), v[3] ncclGetVersion. */
int uspmv_runtime_versions(int v[4]);
/* Ranks of the RCCL communicator the step's exchange runs on (ncclCommCount; the reference prints MPI_Comm_size, code/main.cpp:1838);
 * 0 when the exchange is staged through the host (no communicator exists). */
int uspmv_dist_comm_count(const uspmv_dist_t *d, int *n_ranks);
/* meta[12] = n_local, n_halo, padded_vec_size, n_send, n_interior, n_boundary, ids_are_tiles, n_rows_padded, loopback,
 *            graph captured, graph launches so far, eager steps so far */
int uspmv_dist_info(const uspmv_dist_t *d, int64_t meta[12]);
/* borrowed handles of an object made by uspmv_dist_create_from_coo (NULL scs / halo otherwise) */
int uspmv_dist_parts(const uspmv_dist_t *d, const uspmv_scs_t **scs, const uspmv_dmat_t **A, const uspmv_halo_t **halo);
int uspmv_dist_set_overlap(uspmv_dist_t *d, int overlap);
/* -no_pack 1 of the reference (code/classes_structs.hpp:941): skip the pack kernel, the exchange ships a stale buffer (timing only) */
int uspmv_dist_set_no_pack(uspmv_dist_t *d, int no_pack);
/* one step, issued eagerly: d_x (padded_vec_size elements; its halo tail is written), d_y (n_rows_padded written) */
int uspmv_dist_spmv(uspmv_dist_t *d, void *d_x, void *d_y, int comm_halos, void *stream);
/* n_steps steps back to back.  use_graph != 0: the step is captured once per (d_x, d_y, stream) into a hipGraph and
 * replayed (needs an explicit stream); falls back to eager steps when the runtime refuses the capture. */
int uspmv_dist_run(uspmv_dist_t *d, void *d_x, void *d_y, int n_steps, int use_graph, void *stream);
/* Block vectors: Y = A X with the halo exchange of the b vectors in one of the reference's message patterns (compile-time modes
 * there: SINGLEVEC / MULTIVEC / BULKVEC_MPI_MODE, code/classes_structs.hpp:875-924, code/mpi_funcs.hpp:35-60), then uspmv_spmmv on
 * the block (ld = padded_vec_size).  Column-wise X: all three; row-wise X: USPMV_BULKVEC (its per-neighbour block IS the halo
 * region of X, no staging).
 * The reference exchanges first and computes then (code/mpi_funcs.hpp:25-60).  Here the step has the same two parts as the
 * single-vector one ("overlap" 1, the default): the chunks that touch no halo row run on a side stream while the exchange is under
 * way, the others after it -- the same kernel twice over chunk-length arrays in which the other part's chunks are marked, so every
 * row's FMA chain is the one-part step's (bit-identical Y).  Column-wise X: the interior part re-lays out the local rows into the
 * handle's row-major workspace, the boundary part the halo rows once they are there.  uspmv_dist_set_option(d, "block_plan", b)
 * builds the phased block plan of uspmv_dmat_optimize_block on the rank's matrix (64-byte X rows) together with the
 * interior / boundary classes of its tiles; uspmv_dist_set_option(d, "overlap", 0) restores exchange-then-compute.
 * With USPMV_EXCHANGE_HOST the same wire formats travel through pinned host memory and the transport's all-to-all-v (tests: real ranks
 * sharing one GPU on unequal blocks). */
typedef enum { USPMV_BULKVEC = 0, USPMV_MULTIVEC = 1, USPMV_SINGLEVEC = 2 } uspmv_vecmode;
int uspmv_dist_spmmv(uspmv_dist_t *d, void *d_X, void *d_Y, int b, int layout, int mode, int comm_halos, void *stream);
/* Padding tiles of the single-vector step (tile lists, "pad_split" 1; an option, off by default -- see DESIGN 6.4 for the A/B).  The reference pads chunks with (value +0, column 0);
 * on every rank but the first, column 0 is a halo column (code/mpi_funcs.hpp:279-306), so most tiles of a rank touch the halo only
 * through fma(+0, x[pad_col], acc).  They run with the interior tiles, before the exchange has delivered x[pad_col]: for finite
 * operands of one sign the product is the same signed zero and y comes out bit for bit as with the delivered value; a guard kernel
 * compares the slot's value before and after the exchange and re-runs those tiles after the boundary tiles otherwise.
 * meta[4] = padding tiles, boundary tiles with real halo references, the padding column's local index (-1: none), re-runs so far */
int uspmv_dist_pad_info(const uspmv_dist_t *d, int64_t meta[4]);
/* meta[6] = block-vector steps taken in two parts, in one part, block width of the plan built through "block_plan" (0 = none),
 * tiles of that plan, its boundary tiles, 1 when the handle can run two-part steps at all */
int uspmv_dist_spmmv_info(const uspmv_dist_t *d, int64_t meta[6]);
/* MPI_Barrier / MPI_Allreduce(MAX) / MPI_Allgather twins on the object's communicator (bench loop, code/main.cpp:461-474) */
int uspmv_dist_barrier(uspmv_dist_t *d, void *stream);
int uspmv_dist_allreduce_max(uspmv_dist_t *d, double *value, void *stream);
int uspmv_dist_allgather_i64(uspmv_dist_t *d, int64_t value, int64_t *all /* comm_size (loopback: P) */, void *stream);
void uspmv_dist_free(uspmv_dist_t *d);

/* ------------------------------------------------------------------ measurement helpers   */
/* Device STREAM kernels (copy: a=b, triad: a=b+s*c, read: sum-reduce b) used as the roofline
 * denominator measured in the same run.  n = number of doubles. */
int uspmv_stream_copy(double *d_a, const double *d_b, int64_t n, void *stream);
int uspmv_stream_triad(double *d_a, const double *d_b, const double *d_c, double s, int64_t n, void *stream);
int uspmv_stream_read(const double *d_b, int64_t n, double *d_partial, void *stream);
/* Second yardstick next to the streams: the x traffic of the tile-local-column SpMV without its matrix stream.  ceil(n / rows) workgroups,
 * mapped to the XCDs like the kernel's tiles ("xcd_remap"), each read the nine runs of x lines a `rows`-row tile of a 27-point stencil
 * with the given plane / line strides touches (16-byte loads).  d_partial: 4 * ceil(n / rows) doubles.  *bytes = bytes gathered. */
int uspmv_stream_gather_lines(const double *d_x, int64_t n, int64_t plane, int64_t line, int rows, double *d_partial, void *stream, int64_t *bytes);
/* Time `reps` back-to-back launches of one entry point with HIP events on `stream`;
 * what: 0 spmv(A,x,y) 1 stream_copy 2 stream_triad 3 stream_read 4 spmv_ap(A,B,x,y)
 *       5 spmmv(A,X,Y,b,ld,layout) 6 stream_gather_lines(x, n, plane = ld, line = b, rows = layout).  Two untimed launches precede the timed ones; returns the average milliseconds per launch.
 * For the STREAM kinds d_x is the source (n doubles; 2n for the triad: b = d_x, c = d_x + n) and d_y
 * the destination (n doubles; 8192 for the read kernel's partial sums). */
int uspmv_time_launches(int what, int reps, const uspmv_dmat_t *A, const uspmv_dmat_t *B, const void *d_x,
                        void *d_y, int64_t n, int b, int64_t ld, int layout, void *stream, double *avg_ms);

#ifdef __cplusplus
}
#endif
#endif /* USPMV_H */
