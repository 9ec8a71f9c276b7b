// Shared declarations of the device half of libuspmv: the matrix handle, the tuning knobs, the
// device helpers every kernel file uses and the launch entry points the C ABI (uspmv_api.hip) calls.
// Kernels live in spmv_kernels.hip, spmmv_kernels.hip and ap_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../host/uspmv_internal.hpp"

struct uspmv_dmat {
    int64_t C = 0, n_chunks = 0, n_elements = 0;
    int dtype = USPMV_F64;
    const int32_t *chunk_ptrs = nullptr, *chunk_lengths = nullptr, *col_idxs = nullptr;
    const void *values = nullptr;
    bool owns = false;
    bool crs = false;
    long n_store = 0;              // rows of y the kernels may write (= n_chunks*C unless re-chunked)
    uspmv_dmat *alt = nullptr;     // internal C = 32 re-chunking of a C in {1,2,4,8,16} struct (same row order)
    // scratch for the internal row-major copies of column-major block vectors (uspmv_spmmv); grown
    // on demand, released with the handle.  Not thread-safe per handle, like the reference's kernel object.
    mutable void *ws = nullptr;
    mutable size_t ws_bytes = 0;
    // uspmv_spmmv_x_prepared: the workspace holds the re-laid-out copy of THIS column-major X (b, ld; form 1 plain / 2 sigma permutation undone)
    mutable const void *xprep_ptr = nullptr;
    mutable int xprep_b = 0, xprep_form = 0;
    mutable long xprep_ld = 0;
    // SpMMV in two parts (the halo overlap of uspmv_dist_spmmv, csrc/uspmv_dist_api.hip): chunk-length arrays in which the chunks of the
    // OTHER part carry USPMV_SKIP_LEN -- a kernel that meets it leaves those rows of Y alone.  [order][part - 1]: order 0 = the caller's
    // row order (gather kernels), order 1 = the phased plan's tie-re-ordered rows (scs_spmmv_quadph), classified per 64-row plan tile.
    // `part` selects around ONE launch: 0 the whole matrix, 1 interior, 2 boundary; column-major callers: part 1 re-lays out X rows
    // [0, part_split) into the workspace, part 2 the rest (the halo rows, after the exchange).
    int32_t *part_len[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    int part = 0;
    long part_split = 0;
    // tile-local-column plan (host/tlc_plan.cpp), device copies owned by the handle
    bool tlc = false;
    bool tlc_elem = false;   // the plan lists single x ELEMENTS instead of 16-element lines (tlc_max_lines counts elements): rows whose columns are scattered
                             // (a numbering that is only locally coherent) -- uspmv_dmat_optimize falls to it when the line plan stages too few tiles
    int tlc_max_lines = 0, tlc_tile_rows = 256;
    int64_t tlc_x_len = 0, tlc_n_tiles = 0, tlc_staged = 0;
    uint64_t tlc_plan_id = 0;   // structs planned together (ap pair) carry the same non-zero id
    int32_t *tlc_line_ptr = nullptr, *tlc_lines = nullptr;
    uint32_t *tlc_c16_ptrs = nullptr;
    uint16_t *tlc_col16 = nullptr;
    // ... and the same local indices packed to 12 bits (plans of at most 256 lines per tile: 9.5 instead of 10 bytes per non-zero of the stream):
    // per chunk, at tlc_c12_ptrs[c] dwords: for every PAIR of slot groups [row][3 dwords] (96 bits per row = 8 indices, one 12-byte load per
    // lane; three planes of C dwords measured 3 % slower), then for an odd last group one plane of C dwords + one of C ushorts (48 bits per
    // row).  What scs_spmv_tlc reads when present.
    uint32_t *tlc_c12_ptrs = nullptr, *tlc_col12 = nullptr;
    // element plan over rows DEALT TO THE TILES BY THE MATRIX GRAPH (uspmv_scs_reorder_rows mode 4, as for the block plan): a private copy of the values in that
    // order and row_map[plan row] = row of y.  Null when the caller's row order is kept.
    void *tlc_values = nullptr;
    int32_t *tlc_row_map = nullptr, *tlc_cols = nullptr;     // (tlc_cols: the column indices in that order, for the few tiles that do not stage)
    // block (SpMMV) plan: 64-row tiles, per tile the list of X rows it touches (uspmv_dmat_optimize_block)
    bool bt = false;
    int bt_max_rows = 0, bt_tile_rows = 64;
    int64_t bt_n_tiles = 0, bt_staged = 0;
    int32_t *bt_line_ptr = nullptr, *bt_xrows = nullptr;
    uint32_t *bt_c16_ptrs = nullptr;
    uint16_t *bt_col16 = nullptr;
    // the plan's private copy of the entries with ties of the sigma sort back in original-row order
    // (uspmv_scs_reorder_ties): values (+ 32-bit columns when some tile keeps the gather path) and
    // row_map[plan row] = row of y.  All null when the caller's order is kept.
    void *bt_values = nullptr;
    int32_t *bt_cols = nullptr, *bt_row_map = nullptr;
    // phased block plan (uspmv_build_phased_plan; 64-byte X rows): per tile a run of phases, each with its own X-row list
    bool pb = false;
    int pb_cap_rows = 0, pb_ngp = 0, pb_max_rows = 0;
    int64_t pb_n_tiles = 0, pb_n_phases = 0, pb_rows_staged = 0;
    int32_t *pb_ph_ptr = nullptr, *pb_g0 = nullptr, *pb_list_ptr = nullptr, *pb_xrows = nullptr;
    void *pb_values = nullptr;          // the entries again, GROUP-major like pb_col16 ([chunk][group of four slots][row][slot % 4])
    uint32_t *pb_c16_ptrs = nullptr;
    uint16_t *pb_col16 = nullptr;       // phase-local indices; ONE BYTE each when pb_idx8 (no phase lists more than 256 rows)
    bool pb_idx8 = false;
    bool pb_device_built = false;       // the plan's index part was built by csrc/block_plan_kernels.hip
    // the phased plan once more as a flat schedule of 32-byte phase descriptors for persistent workgroups (spmmv_stream.hip; "spmmv_stream")
    int ps_grid = 0;
    bool ps_per_tile = false;           // the schedule has one tile per workgroup ("spmmv_stream" 99)
    int64_t ps_n_desc = 0;
    int32_t *ps_wg_ptr = nullptr;
    void *ps_desc = nullptr;
    // the same plan once more with LINE lists (128 bytes of one column: 16 doubles / 32 floats) for column-major block vectors:
    // shares pb_values / pb_c16_ptrs / the row map; one-byte local indices (line << shift | row in line)
    bool pl = false;
    int pl_shift = 0, pl_max_rows = 0;
    int64_t pl_n_phases = 0, pl_rows_staged = 0;
    int32_t *pl_ph_ptr = nullptr, *pl_g0 = nullptr, *pl_list_ptr = nullptr, *pl_lines = nullptr;
    uint8_t *pl_col8 = nullptr;
    // ... and once more for column-major block vectors behind the re-layout pass, with the X rows numbered in the ORIGINAL row order:
    // the pass that turns the caller's column-major X into the row-major workspace also undoes the sigma permutation
    // (Xr[r] = X[old_to_new[r]]), so a tile's X rows form long runs again instead of ~30 fragments per phase and a 128-byte line of the
    // workspace holds two rows the tile needs instead of 1.1 (shares pb_values / pb_c16_ptrs / the row map)
    bool pu = false;
    int pu_max_rows = 0;
    int64_t pu_n_phases = 0, pu_n_perm = 0;
    int32_t *pu_ph_ptr = nullptr, *pu_g0 = nullptr, *pu_list_ptr = nullptr, *pu_xrows = nullptr, *pu_perm = nullptr;
    uint8_t *pu_col8 = nullptr;
    // block-vector column-window sweep plan (uspmv_build_block_sweep_plan, csrc/spmmv_sweep.hip): 64-byte X rows, windows of 2^bw_wlog X rows,
    // per tile the list of windows its rows touch
    bool bw = false;
    int bw_tile_rows = 2048, bw_wlog = 9, bw_b = 0;
    int64_t bw_n_tiles = 0, bw_all_tiles = 0, bw_x_rows = 0, bw_windows = 0;
    int32_t *bw_tile_ids = nullptr, *bw_win_ptr = nullptr, *bw_wins = nullptr, *bw_pad = nullptr;
    uint64_t *bw_cnt_off = nullptr;
    uint32_t *bw_wave_off = nullptr;
    uint8_t *bw_cnt = nullptr;
    void *bw_vals = nullptr;
    uint16_t *bw_idx = nullptr;
    // column-window sweep plan (host/sweep_plan.cpp, uspmv_dmat_optimize_sweep[_ap]); the _b arrays are the sp part of
    // an ap[dp_sp] pair and live on the dp handle, the sp handle only carries the plan id
    bool sw = false;
    int sw_tile_rows = 1024, sw_wlog = 13;
    int64_t sw_n_tiles = 0, sw_all_tiles = 0, sw_x_len = 0, sw_n_rest = 0;
    int64_t sw_n_vals = 0, sw_n_vals_b = 0, sw_cnt_bytes = 0;   // elements of the compacted streams (without the spare tail), bytes of a count array
    uint64_t sw_plan_id = 0;
    int32_t *sw_tile_ids = nullptr, *sw_smin = nullptr, *sw_S = nullptr, *sw_pad = nullptr, *sw_pad_b = nullptr, *sw_rest = nullptr;
    uint64_t *sw_cnt_off = nullptr;
    uint32_t *sw_wave_off = nullptr, *sw_wave_off_b = nullptr;
    uint8_t *sw_cnt = nullptr, *sw_cnt_b = nullptr;
    void *sw_vals = nullptr;
    float *sw_vals_b = nullptr;
    uint16_t *sw_idx = nullptr, *sw_idx_b = nullptr;
};

namespace uspmv_dev {

// "do not MEASURE the rows per tile" (tlc_measure_tile) for the planner calls of the current thread, as a scope
extern thread_local int tl_measure_off;
struct MeasureOff { MeasureOff() { ++tl_measure_off; } ~MeasureOff() { --tl_measure_off; } };

// One-launch distributed step (csrc/uspmv_dist_api.hip): the tile list of a step is [early | late | conditional | early].  Early entries
// (interior + padding tiles) run at once.  A late entry (a tile with real halo references) looks ONCE at the exchange counter: if the
// exchange of this step has completed it proceeds (acquire), otherwise it appends its position to the step's deferred list and
// leaves -- nothing ever spins, so no schedule can deadlock.  A second, small launch after the exchange runs the deferred entries.
// Conditional entries are the padding tiles again; they only run when x[pad_col] changed sign or is not finite (uspmv_dist::pad_col).
struct StepSync {          // device memory, 32 bytes
    int flag;              // exchanges completed so far (incremented by a one-thread kernel behind the RCCL group)
    int expect;            // value `flag` must have reached for the current step's late entries
    int count[2];          // deferred entries of the current / next step (indexed by expect & 1)
    int reruns;            // steps whose conditional entries ran
    int spare[3];
};
struct StepArgs {
    long n_early = 0, n_real = 0, n_cond = 0, defer_cap = 0;
    long late0 = 0;                // the late + conditional entries sit at [late0, late0 + n_real + n_cond) of the list, early entries around them
    StepSync *ss = nullptr;
    int *defer = nullptr;          // [2][defer_cap] positions in the step list
    const void *stale = nullptr;   // value of x[pad_col] before the exchange
    int pad_col = -1;
};

constexpr int USPMV_SKIP_LEN = -4;     // (a multiple of four: no kernel sees a partial group or a tail in it)

// the chunk lengths a launcher hands to its kernel: the handle's own, or the selected part's (order 0 caller's rows, 1 phased plan rows)
inline const int32_t *part_lengths(const uspmv_dmat *A, int order) {
    return A->part && A->part_len[order][A->part - 1] ? A->part_len[order][A->part - 1] : A->chunk_lengths;
}
// false: a part is selected and this order has no arrays for it -- the launcher must not run
inline bool part_ok(const uspmv_dmat *A, int order) { return !A->part || A->part_len[order][A->part - 1]; }

struct Tuning {
    // defaults = fastest of the interleaved sweep on the nlpkkt200-class matrix (profiles/r01_sweep253.txt)
    int unroll = 8;
    int nontemporal = 1;
    int xcd_remap = 256;  // groups of 256 consecutive workgroups per XCD (profiles/r01/sweepH.txt); bits 20+: the stagger (remap_block)
    int block = 256;
    int spmv_variant = 0;
    int csr_lanes = 0;  // 0 = choose from average row length
    int ablate = 0;     // measurement only
    int tlc = 1;            // use the tile-local-column kernel when the handle carries a plan
    int rechunk = 1;        // uspmv_dmat_optimize may re-chunk C < 32 structs to C = 32 internally
    int tlc_tile_rows = 0;    // rows (= threads) per tile used by the NEXT uspmv_dmat_optimize[_ap / _device]: 256 | 512 | 1024, 0 = 256 for one
                              // struct and 512 for an ap[dp_sp] pair (two streams: the x window of 256 rows left 20 waves per CU, profiles/r02/ap_tile_rows.txt)
    int tlc_auto_tile = 1;    // with tlc_tile_rows 0 and one struct: 1024- or 512-row tiles when the largest 256-row tile needs > 250 x lines
                              // (<= 16 waves per CU) and the larger tiles still stage >= 99 % of the tiles (profiles/r03/tile_rows_sweep.txt)
    int tlc_measure_tile = 1; // large single structs (>= 2^20 padded rows): build the plan for 256 / 512 / 1024 rows on the device, time the kernel, keep
                              // a larger tile when it is > 3 % ahead (uspmv_api.hip measured_tile_rows; the 304^3 stencil: 512 rows, 7 % ahead)
    int tail_batch = 0;     // ragged tail of a chunk as one predicated batch
    int spmmv_unroll = 0;   // 0 = auto (256 bytes of X rows per lane and batch)
    int spmmv_lds_kb = 0;    // block plan: LDS budget per tile in KiB for the NEXT uspmv_dmat_optimize_block (0 = 80)
    int spmmv_tile_rows = 0; // block plan: 0 = auto (32-row tiles for >= 64-byte rows on C = 32), 64 = always 64
    int raw_plan_cache = 0; // uspmv_scs_gpu_*: remember a device-built plan per set of array addresses (opt-in)
    int spmmv_swizzle = 0;  // block-plan kernel: 1 = piece-swizzled X rows in LDS (all 64 banks); 0 = plain layout, which
                            // needs ~10 fewer address instructions per non-zero and measures 5-7 % faster (spmmv_probe16.txt)
    int spmmv_prefetch = 1; // row-major lane-per-row kernel: request batch k+1's matrix entries behind batch k's X rows
    int sweep = 1;          // use the column-window sweep kernel when the handle carries a sweep plan
    int sweep_threads = 0;     // threads per sweep workgroup (0 = min(tile rows, 1024); 256 | 512: a lane owns tile rows / threads rows, at most 4)
    int sweep_nbuf = 1;     // LDS buffers per workgroup: 1 = two 1024-thread workgroups per CU cover each other's staging (0.63 vs 0.72 ms on
                            // config 4b); 2 = one workgroup, window s+1 lands while window s is consumed
    int sweep_pair = 2;     // 1: two chains per lane side by side (two rows of the lane, or the dp and the sp part of a row): twice the entries in flight per
                            // wave (config 4b: ap 0.536 -> 0.514 ms, dp 0.585 -> 0.576); 2: ... and the FMAs under the rounds' lane masks (EXEC) instead of
                            // copy + FMA + two selects (ap 0.527 -> 0.511, dp 0.578 -> 0.570; profiles/r03/config4b_sweep.txt); 0: one chain at a time
    int sweep_unroll = 8;   // rounds per batch
    int sweep_remap = 8;    // consecutive sweep tiles per XCD (neighbouring tiles share their x windows)
    int sweep_wlog = 0;     // NEXT uspmv_dmat_optimize_sweep: log2 of the window width in elements (0 = 64 KiB of VT)
    int sweep_tile_rows = 0;  // NEXT uspmv_dmat_optimize_sweep: 256 | 512 | 1024 | 2048 | 4096 rows per tile (0 = default; above 1024: several rows per lane)
    int sweep_max_stage = 0;  // NEXT plan: largest staging cost in bytes per non-zero for a tile to sweep (0 = 24)
    int spmmv_unscramble = 0;  // 1: NEXT uspmv_dmat_optimize_block (64-byte rows, sigma > 1, host struct with its permutation) also builds the plan over
                               // ORIGINAL X-row numbering; uspmv_spmmv on column-major vectors then lets the re-layout pass undo the sigma permutation (a tile's X
                               // rows become runs of the workspace, two needed rows per 128-byte line instead of 1.1).  Measured 1 % SLOWER on config 3
                               // (0.925 vs 0.912-0.919 ms, profiles/r03/config3_colwise.txt): the staging is not bound by L2 -> L1 lines.  Off by default.
    int block_plan_device = 1;  // uspmv_dmat_optimize_block_device: 1 = row order, phases, lists and indices built on the device (block_plan_kernels.hip),
                               // 0 = index arrays copied to the host and planned there (values gathered on the device either way)
    int spmmv_xline = 0;       // 1: NEXT uspmv_dmat_optimize_block with 64-byte rows also builds the LINE plan and uspmv_spmmv stages column-major X by 128-byte
                               // lines, without the re-layout pass.  Off by default: under sigma > 1 the column numbering is scrambled inside the windows (lines
                               // are 1/3 used, the planner turns the plan down), and at sigma = 1, where it qualifies, it measures 0.930 ms against 0.905 ms
                               // behind the re-layout pass on config 3 (1.35 x the phases; profiles/r03/config3_colwise.txt)
    int spmmv_ycol_nt = 0;     // phased SpMMV kernel, column-major Y: 1 = non-temporal element stores, 0 = plain (write-back) stores that the L2 can merge into whole lines
    int spmmv_xcol = 0;        // phased SpMMV kernel on column-major X: 0 = separate re-layout pass first (1.078 ms on config 3), 1 = rows assembled in LDS by
                               // the kernel itself from the column-major vector (no workspace, no extra launch, but 1.123 ms: 74 registers, six workgroups per CU)
    int spmmv_phased = 1;      // NEXT uspmv_dmat_optimize_block with 64-byte rows: also build the phased plan (eight workgroups per CU)
    int spmmv_phase_rows = 256;  // ... X rows per phase (256 | 512)
    int spmmv_list_plan = 0;   // NEXT optimize_block: also build the one-list-per-tile plan (variants 4 / 5 / 6) when the phased kernel can take the matrix
    int spmmv_idx8 = 1;        // NEXT optimize_block: one-byte phase-local indices when every phase lists <= 256 rows
    int tlc_elem = 1;        // NEXT uspmv_dmat_optimize (host planner, one struct): when the 16-element-line plan stages fewer than half of the tiles, try the
                             // plan over single x elements (each distinct column of a tile gathered once into LDS) before the column-window sweep
    int tlc_elem_rows = 1;   // ... when the element plan over the caller's row order fails too: 1 = deal the rows to the tiles by the matrix graph first (private value copy + row map)
    int tlc_elem_seg_rows = 65536;  // ... rows of the segments the row dealing clusters independently (larger: numberings that scatter related rows further apart)
    int tlc_elem_cap = 4096; // ... most elements a tile may list (4096: 32 KiB of doubles, local indices still fit 12 bits)
    int tlc_idx12 = 1;       // NEXT optimize: tile-local-column plans of <= 256 lines per tile also get their local indices packed to 12 bits: 0 = never,
                             // 1 = kept when the mean row length is >= 8, 2 = kept wherever it can be built
    int spmmv_reorder = 4;  // block plan's private copy of the entries (host planner): 1 = rows of equal-length chunks of a sigma window back in original order;
                            // 4 = on top of that, rows re-dealt to the tiles as FLAT patches of the matrix graph (grown along the slots of one phase
                            // around the diagonal: 7.9 instead of 11.9 staged X rows per row on config 3; kept only where a sample of tiles
                            // confirms it); 2 = the same with balls over all slots (fewer distinct X rows per tile, but more per phase); 0 = as is.
                            // The device-side builder (handles without a host struct) always does 1.
    long spmmv_brick_stride = 0;  // spmmv_reorder 3 (measurement aid): rows of a mesh line in ORIGINAL numbering; tiles = flat bricks of spmmv_brick_lines lines
    int spmmv_brick_lines = 4;
    int spmmv_phase_dp = 24;  // NEXT optimize_block (host planner): > 0 = phase cuts by dynamic programming (least staged rows + this many rows' worth per
                              // phase), 0 = every phase filled to the brim (what the device-side builder does)
    int spmmv_stream = 0;   // NEXT optimize_block (64-byte rows, C = 32, one-byte indices): > 0 = also lay the phased plan out as a flat schedule for this many
                            // persistent workgroups per CU (at most 5: 32 KiB of LDS each) and let uspmv_spmmv run the streaming kernel (spmmv_stream.hip)
    int spmmv_stream_waves = 4;  // ... depth 1: register budget of the kernel in waves per SIMD (4 | 5)
    int spmmv_stream_depth = 1;  // ... 1 = X rows and entries one phase ahead (two LDS buffers, full wait per phase); 2 = two phases ahead (three buffers, partial wait)
    int spmmv_stream_xcd = 1;  // ... 1 = the workgroups of an XCD take consecutive tiles, 0 = tile t goes to workgroup t % grid
    int spmmv_variant = 0;  // 0 = auto (= 3 where a B-specialised kernel exists); 1 = generic kernel; 2 = row-major with transposing X phase; 3 = row-major, lane per row
};
extern Tuning g_tune;   // uspmv_api.hip

int require_device();                                    // uspmv_api.hip
int check_dmat(const uspmv_dmat *A, const char *who);    // uspmv_api.hip
inline unsigned grid_for(long work_items, int block) { return (unsigned)((work_items + block - 1) / block); }

constexpr size_t BT_LDS_CAP = 80 * 1024;  // LDS per single-wave SpMMV tile (block plan): two tiles per CU at worst

// launch entry points (explicitly instantiated for double and float in their kernel files)
template <typename VT>
int launch_spmv_scs(const uspmv_dmat *A, const int *chunk_ids, long n_ids, const VT *x, VT *y, hipStream_t st);   // spmv_kernels.hip
template <typename VT>
int launch_spmv_tlc(const uspmv_dmat *A, const int *tile_ids, long n_tiles, const VT *x, VT *y, hipStream_t st);  // spmv_kernels.hip
// the one-launch distributed step over `step_ids` = [early | late | conditional] (sync 1) and its deferred entries (sync 2)
template <typename VT>
int launch_spmv_tlc_step(const uspmv_dmat *A, const int *step_ids, const StepArgs &sa, int sync, const VT *x, VT *y, hipStream_t st);
template <typename VT>
int launch_csr(long n_rows, long nnz_hint, const int *rp, const int *ci, const VT *va, const VT *x, VT *y, hipStream_t st);  // spmv_kernels.hip
template <typename VT>
int launch_spmmv(const uspmv_dmat *A, const VT *X, VT *Y, int b, long ld, int layout, hipStream_t st);            // spmmv_kernels.hip
template <typename VT>
int launch_spmmv_sweep(const uspmv_dmat *A, const VT *X, VT *Y, int b, long ld, bool xcol, bool ycol, hipStream_t st);   // spmmv_sweep.hip; false-y: USPMV_OK when launched, > 0 when the plan does not apply
template <typename VT>
int prepare_x(const uspmv_dmat *A, const VT *X, int b, long ld, hipStream_t st);                                    // spmmv_kernels.hip
// phased block plan, 64-byte X rows (spmmv_phased.hip); false: no plan / schedule on the handle or it does not fit the compiled shapes
// xmode: 0 = row-major X, 1 = column-major X assembled through registers, 2 = column-major X staged by 128-byte lines (line plan)
bool spmmv_phased(const uspmv_dmat *A, const double *X, double *Y, long ld, bool ycol, int xmode, hipStream_t st);
bool spmmv_phased(const uspmv_dmat *A, const float *X, float *Y, long ld, bool ycol, int xmode, hipStream_t st);
// the same plan walked by persistent workgroups (spmmv_stream.hip); false: no schedule on the handle / shape not covered
bool spmmv_stream(const uspmv_dmat *A, const double *X, double *Y, long ld, bool ycol, hipStream_t st);
bool spmmv_stream(const uspmv_dmat *A, const float *X, float *Y, long ld, bool ycol, hipStream_t st);
int dmat_stream_schedule(uspmv_dmat *A, int wgs_per_cu);
void dmat_stream_release(uspmv_dmat *A);
int launch_spmv_ap(const uspmv_dmat *dp, const uspmv_dmat *sp, const double *d_x, const float *d_x_sp, double *d_y,
                   hipStream_t stream);                                                                           // ap_kernels.hip
template <typename VT>
int launch_spmv_sweep(const uspmv_dmat *A, const VT *x, VT *y, hipStream_t st);                                    // sweep_kernels.hip (sweep tiles only)
int launch_spmv_sweep_ap(const uspmv_dmat *dp, const double *x, double *y, hipStream_t st);                       // sweep_kernels.hip
int launch_spmv_ap_chunks(const uspmv_dmat *dp, const uspmv_dmat *sp, const int *chunk_ids, long n_ids, const double *d_x,
                          double *d_y, hipStream_t stream);                                                        // ap_kernels.hip

// (A2 / the *_2 arrays: optional second struct sharing the plan -- the sp part of an ap[dp_sp] pair)
int launch_plan_count(const uspmv_dmat *A, long n_tiles, int max_lines, int *d_n_lines, int *d_max_col, hipStream_t st,
                      const uspmv_dmat *A2 = nullptr, int tile_rows = 256);                                               // plan_kernels.hip
int launch_plan_pack12(const uspmv_dmat *A, const unsigned *d_c16_ptrs, const unsigned short *d_col16, const unsigned *d_c12_ptrs, unsigned *d_col12, hipStream_t st);   // plan_kernels.hip
int launch_plan_write(const uspmv_dmat *A, long n_tiles, const int *d_tile_line_ptr, const unsigned *d_c16_ptrs, int *d_tile_lines,
                      unsigned short *d_col16, hipStream_t st, const uspmv_dmat *A2 = nullptr, const unsigned *d_c16_ptrs2 = nullptr,
                      unsigned short *d_col16_2 = nullptr, int tile_rows = 256);                                           // plan_kernels.hip
int launch_rechunk32(const uspmv_dmat *A, const int *d_cp_new, int *d_ci_new, void *d_va_new, hipStream_t st);             // plan_kernels.hip
int launch_block_values_gather(const uspmv_dmat *A, const int *d_row_map, const unsigned *d_c16_ptrs, void *d_out, bool group_major, hipStream_t st);   // plan_kernels.hip
// device-side builder of the phased block plan (block_plan_kernels.hip)
int launch_block_reorder(const uspmv_dmat *A, int *d_row_map, int *d_changed, hipStream_t st);
int launch_block_tile_class(const uspmv_dmat *A, long n_local, unsigned char *d_flags, hipStream_t st);
int launch_part_len_fill(const uspmv_dmat *A, int rows_per_flag, const unsigned char *d_flags, int *d_len_int, int *d_len_bnd, hipStream_t st);
// two-part SpMMV (uspmv_dmat::part_len): order 0 from per-chunk flags (1 = the chunk touches a halo column), order 1 from the lists of the
// handle's phased plan (no-op without one); the arrays belong to the handle (order 1 goes with the plan)
void dmat_block_plan_release(uspmv_dmat *A);
int dmat_part_set_chunks(uspmv_dmat *A, const unsigned char *h_chunk_flags);
int dmat_part_set_plan(uspmv_dmat *A, long n_local, int64_t *n_boundary_tiles);
int launch_block_phase_plan(const uspmv_dmat *A, bool write, int cap, int ngp, const int *d_row_map, const unsigned *d_c16_ptrs, int *d_t_phases,
                            int *d_t_list, int *d_ph_g0, int *d_ph_list_ptr, int *d_xrows, unsigned char *d_col8, int *d_max_rows, hipStream_t st);
// device-side builder of the column-window sweep plan (sweep_plan_kernels.hip)
int launch_sweep_scan(const uspmv_dmat *A, int wlog, int *d_row_le, int *d_row_pad, int *d_grp, int *d_max_col, hipStream_t st);
int launch_sweep_fill(const uspmv_dmat *A, int wlog, int R, long n_sweep_tiles, const int *d_tile_ids, const int *d_smin, const int *d_S,
                      const unsigned long long *d_cnt_off, const unsigned *d_wave_off, const int *d_row_le, const int *d_row_pad,
                      unsigned char *d_cnt, void *d_vals, unsigned short *d_idx, int *d_pad_col, hipStream_t st);

}  // namespace uspmv_dev

#define HIP_TRY(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return uspmv::fail(USPMV_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                               __FILE__, __LINE__);                                                \
    } while (0)


// ------------------------------------------------------------------------------------------
template <bool NT, typename T>
__device__ __forceinline__ T ld_stream(const T *p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
// the same through an explicitly GLOBAL pointer: where the address reaches the load through a reference or a local pointer array
// the compiler no longer knows the address space and emits flat_load (which also ties up the LDS counter) instead of global_load
template <bool NT, typename T>
__device__ __forceinline__ T ld_stream_g(const T *p) {
    typedef const T __attribute__((address_space(1))) *gp_t;
    if constexpr (NT) return __builtin_nontemporal_load((gp_t)p);
    else return *(gp_t)p;
}
// y store.  A y vector is ~2 % of the bytes of an SpMV, but its HBM write stream costs 15-18 % of the
// kernel when it goes through the write-back L2 (profiles/r01_microbench.txt: 0.74 ms without the
// store, 0.88 ms with plain stores).  Non-temporal stores (`global_store ... nt`) are the cheapest form
// measured: A/B of two builds on one box (tools/ab.sh, profiles/r01/ystore_ab.txt) 0.765 ms against 0.822 ms
// for write-through agent-scope stores (`sc1`) and 0.840 ms for plain stores on the nlpkkt200-class matrix.
template <bool WT, typename T>
__device__ __forceinline__ void st_y(T *p, T v) {
    if constexpr (WT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// DPP quad broadcast: every lane of a quad gets lane U's value (the four-lanes-per-row SpMMV kernels)
template <int U>
__device__ __forceinline__ int quad_bcast(int v) { return __builtin_amdgcn_update_dpp(0, v, U * 0x55, 0xf, 0xf, true); }   // (bound_ctrl + full masks: every lane is written, `old` is dead)
template <int U>
__device__ __forceinline__ double quad_bcast(double v) {
    return __hiloint2double(quad_bcast<U>(__double2hiint(v)), quad_bcast<U>(__double2loint(v)));
}
template <int U>
__device__ __forceinline__ float quad_bcast(float v) { return __int_as_float(quad_bcast<U>(__float_as_int(v))); }

// logical block id.  Hardware deals blocks round-robin over the 8 XCDs (b, b+8, b+16, ... share
// one).  mode 0: identity.  mode 1: every XCD walks one contiguous eighth of the grid.
// mode G >= 2: groups of G consecutive logical blocks per XCD, the 8 groups of a super-block
// of 8*G blocks being processed concurrently (keeps all XCDs inside one moving DRAM window while
// neighbouring blocks -- which share x lines -- share an L2).
// Bits 20+ of `mode` (tuning "xcd_stagger" S): XCD k starts its group S*k blocks in, wrapping around -- the eight XCDs, which move
// through their groups in step, then no longer touch addresses that differ by exact multiples of the group's byte size (256 tiles x 2 KB
// of y = 512 KiB: eight write streams 2^19 bytes apart land on whatever memory channels the physical address bits above 2^19 select --
// the same few, or all different, by the luck of the process's physical pages: profiles/r04/placement_*.txt).
__device__ __forceinline__ unsigned remap_block(unsigned b, unsigned nb, int mode) {
    const unsigned S = (unsigned)mode >> 20;
    mode &= 0xFFFFF;
    if (mode == 0 || nb < 16) return b;
    if (mode == 1) {
        const unsigned xcd = b & 7u, q = nb >> 3, r = nb & 7u;
        const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        return base + (b >> 3);
    }
    const unsigned G = (unsigned)mode, SG = 8u * G;
    const unsigned full = (nb / SG) * SG;
    if (b >= full) return b;
    const unsigned sup = b / SG, rem = b - sup * SG;
    const unsigned xcd = rem & 7u;
    unsigned pos = (rem >> 3) + S * xcd;
    if (S) pos %= G;
    return sup * SG + xcd * G + pos;
}

