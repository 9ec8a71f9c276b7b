// Internal definitions shared by the host data layer and the HIP C-ABI layer of libuspmv.so.
#pragma once
#include <sys/mman.h>
#include <cstdint>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <memory>
#include <utility>
#include <vector>

#include "uspmv.h"

// COO matrix, 0-based, entries stable-sorted by row when produced by uspmv_read_mtx
// (role of MtxData<double,int>, reference code/classes_structs.hpp:1169-1238).
// allocator whose resize() leaves new elements uninitialised: the three arrays of a large COO are first touched by the threads that fill
// them (a value-initialising resize is a serial 3 GB memset for 2e8 entries, host/mtx_io.cpp)
template <class T> struct uspmv_noinit_alloc : std::allocator<T> {
    template <class U> struct rebind { using other = uspmv_noinit_alloc<U>; };
    uspmv_noinit_alloc() = default;
    template <class U> uspmv_noinit_alloc(const uspmv_noinit_alloc<U> &) {}
    template <class U, class... A> void construct(U *p, A &&...a) {
        if constexpr (sizeof...(A) == 0) ::new ((void *)p) U;
        else ::new ((void *)p) U(std::forward<A>(a)...);
    }
};
using uspmv_ivec = std::vector<int32_t, uspmv_noinit_alloc<int32_t>>;
using uspmv_dvec = std::vector<double, uspmv_noinit_alloc<double>>;

struct uspmv_coo {
    int64_t n_rows = 0, n_cols = 0, nnz = 0;
    uspmv_ivec I, J;
    uspmv_dvec values;
};

// SELL-C-sigma matrix on the host (role of ScsData<VT,int>, code/classes_structs.hpp:1313-1339).
// Element (row-in-chunk i, slot j) of chunk c lives at chunk_ptrs[c] + j*C + i (column-major
// inside a chunk); padding entries have value 0 and column 0.
struct uspmv_scs {
    int64_t C = 0, sigma = 0, n_rows = 0, n_cols = 0, n_rows_padded = 0, n_chunks = 0, n_elements = 0, nnz = 0;
    int dtype = USPMV_F64;
    std::vector<int32_t> chunk_ptrs, chunk_lengths, col_idxs, old_to_new_idx, new_to_old_idx;
    std::vector<double> values_f64;
    std::vector<float> values_f32;
    const void *values_ptr() const {
        return dtype == USPMV_F64 ? (const void *)values_f64.data() : (const void *)values_f32.data();
    }
};

// Per-rank halo description (role of ContextData, code/classes_structs.hpp:156-184).
struct uspmv_halo {
    int P = 0, rank = 0;
    int64_t n_local = 0, n_halo = 0;
    std::vector<int32_t> recv_counts_cumsum;  // P+1, assembled as code/mpi_funcs.hpp:403-414
    std::vector<int32_t> recv_counts;         // P
    std::vector<int32_t> recv_idxs;           // grouped by owner ascending, owner-local row ids
};

// Exchange plan of one rank (host/comm_plan.cpp): what to send to whom, derived from the peers' halo descriptions
// (role of comm_send_idxs / send_counts_cumsum, code/mpi_funcs.hpp:117-232)
struct uspmv_comm_plan {
    int P = 0, rank = 0;
    int64_t n_local = 0, n_send = 0;
    std::vector<int64_t> send_off, recv_off;   // P+1, elements
    std::vector<int32_t> send_idxs;            // this rank's local rows (original order), grouped by receiver
};

// resize() of a large vector of trivially constructible elements without the serial 4-KiB page faults of its zero fill: the storage is
// reserved first and the kernel asked for huge pages on it (a 4 GB copy of config 3's entries: 16 s -> under a second in the build container)
template <typename V>
inline void uspmv_resize_huge(V &v, size_t n) {
    v.reserve(n);
    const uintptr_t a = ((uintptr_t)v.data() + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1), e = ((uintptr_t)(v.data() + v.capacity())) & ~(uintptr_t)((2u << 20) - 1);
    if (n * sizeof(typename V::value_type) >= (64u << 20) && e > a) {
        (void)madvise((void *)a, e - a, MADV_HUGEPAGE);
        volatile char *b = (volatile char *)a;               // first touch by all threads (the storage is raw until resize() fills it)
        const int64_t pages = (int64_t)((e - a) >> 12);
#pragma omp parallel for schedule(static)
        for (int64_t k = 0; k < pages; ++k) b[(size_t)k << 12] = 0;
    }
    v.resize(n);
}

// Tile-local-column plan (host copy), see host/tlc_plan.cpp
struct uspmv_tlc_plan {
    bool valid = false;
    int chunks_per_tile = 0, max_lines_used = 0, tile_rows = 256, line_shift = 4;
    int64_t n_tiles = 0, n_staged_tiles = 0, x_len_min = 0;
    std::vector<int32_t> tile_line_ptr;   // n_tiles+1
    std::vector<int32_t> tile_lines;      // line ids (col >> 4), sorted per tile; empty tile list = gather path
    std::vector<uint32_t> c16_ptrs;       // n_chunks+1, offsets into col16
    std::vector<uint16_t> col16;          // [chunk][slot/4][row][slot%4]
    std::vector<uint32_t> c16_ptrs_b;     // same for the optional second struct (sp part of an ap pair)
    std::vector<uint16_t> col16_b;
};
int uspmv_build_tlc_plan(const uspmv_scs *s, const uspmv_scs *s2, int max_lines, int tile_rows, uspmv_tlc_plan *plan, int line_shift = 4);

// Column-window sweep plan (host copy), see host/sweep_plan.cpp
struct uspmv_sweep_plan {
    bool valid = false;
    int tile_rows = 1024, wlog = 13;
    int64_t n_tiles = 0, n_sweep_tiles = 0, x_len_min = 0;
    std::vector<int32_t> tile_ids, t_smin, t_S;   // per sweep tile: tile number, first window, number of windows
    std::vector<uint64_t> t_cnt_off;              // per sweep tile: offset of its S*tile_rows count bytes
    std::vector<uint32_t> wave_off, wave_off_b;   // per (sweep tile, wave): first element of the wave's compacted stream
    std::vector<uint8_t> cnt, cnt_b;              // [tile][window][row]: entries of the row in the window
    std::vector<uint16_t> idx, idx_b;             // column - window start
    std::vector<double> vals_f64, vals_b_f64;
    std::vector<float> vals_f32, vals_b_f32;
    std::vector<int32_t> pad_col, pad_col_b;      // per (sweep tile, row): column of the stripped trailing padding, -1 = none
    std::vector<int32_t> rest_chunks;             // chunks of the tiles that do not sweep (gather kernel)
};
int uspmv_build_sweep_plan(const uspmv_scs *s, const uspmv_scs *s2, int wlog, int tile_rows, double max_stage_bytes_per_nnz,
                           uspmv_sweep_plan *plan);   // host/sweep_plan.cpp

// Block-vector column-window sweep plan (host copy; host/sweep_plan.cpp, csrc/spmmv_sweep.hip): like uspmv_sweep_plan, but the windows
// are windows of X ROWS (2^wlog rows of block_vec_size values each) and a tile lists only the windows its rows TOUCH -- the rows of a
// 3-D grid matrix reach into three plane-sized index ranges far apart, and staging the gaps between them would cost more than it saves
struct uspmv_block_sweep_plan {
    bool valid = false;
    int tile_rows = 2048, wlog = 9;
    int64_t n_tiles = 0, n_sweep_tiles = 0, x_rows_min = 0, windows_staged = 0;
    std::vector<int32_t> tile_ids;       // per sweep tile: tile number
    std::vector<int32_t> t_win_ptr;      // n_sweep_tiles + 1: the tile's windows in `wins`
    std::vector<int32_t> wins;           // window ids, ascending per tile
    std::vector<uint64_t> t_cnt_off;     // per sweep tile: offset of its (windows x tile_rows) count bytes
    std::vector<uint32_t> wave_off;      // per (sweep tile, wave): first element of the wave's compacted stream
    std::vector<uint8_t> cnt;            // [tile][window of the tile][row]: entries of the row in the window
    std::vector<uint16_t> idx;           // X row - first row of the window
    std::vector<double> vals_f64;
    std::vector<float> vals_f32;
    std::vector<int32_t> pad_col;        // per (sweep tile, row): column of the stripped trailing padding, -1 = none
    std::vector<int32_t> rest_chunks;    // chunks of the tiles that do not sweep
};
int uspmv_build_block_sweep_plan(const uspmv_scs *s, int wlog, int tile_rows, int row_bytes, double max_stage_bytes_per_nnz,
                                 uspmv_block_sweep_plan *plan);   // host/sweep_plan.cpp

// Phased block plan (host copy), see host/tlc_plan.cpp
struct uspmv_phased_plan {
    bool valid = false;
    int cap_rows = 256, ngp = 8, max_rows_used = 0, line_shift = 0;   // line_shift > 0: the lists hold lines of 2^line_shift X rows
    int64_t n_tiles = 0, n_phases = 0;
    std::vector<int32_t> ph_ptr;        // n_tiles+1: phases of a tile
    std::vector<int32_t> ph_g0;         // n_phases: first group (of four slots) of the phase; it ends where the tile's next phase starts
    std::vector<int32_t> ph_list_ptr;   // n_phases+1: the phase's X-row list in xrows
    std::vector<int32_t> xrows;         // sorted per phase
    std::vector<uint32_t> c16_ptrs;     // n_chunks+1
    std::vector<uint16_t> col16;        // [chunk][slot/4][row][slot%4], index into the list of the slot's phase
};
int uspmv_build_phased_plan(const uspmv_scs *s, int cap_rows, int ngp, uspmv_phased_plan *plan, int line_shift = 0, int phase_cost = 0);   // host/tlc_plan.cpp

int uspmv_scs_rechunk32(const uspmv_scs *s, uspmv_scs *out);   // host/tlc_plan.cpp
// per chunk 0 = no halo column, 1 = halo only through +0.0 padding entries on the one column *pad_col, 2 = other halo references (host/halo_plan.cpp)
int uspmv_scs_classify_chunks(const uspmv_scs *s, int64_t n_local, std::vector<uint8_t> *cls, int32_t *pad_col);
// private copy of the entries with the rows of equal-length chunks of a sigma window back in original order;
// returns 1 when anything moved, 0 when the copy is identical (row_map = identity)
int uspmv_scs_reorder_ties(const uspmv_scs *s, uspmv_scs *r, std::vector<int32_t> *row_map);   // host/tlc_plan.cpp
// mode 1 = the above; mode 2 = rows re-dealt to the 64-row tiles as breadth-first balls of the matrix graph (per chunk-length
// class, so the chunk structure is still untouched); mode -1 = copy under the caller's row_map
int uspmv_scs_reorder_rows(const uspmv_scs *s, int mode, uspmv_scs *r, std::vector<int32_t> *row_map, int tile_rows = 64, int64_t seg_rows = 65536);   // host/tlc_plan.cpp
int uspmv_scs_reorder_bricks(const uspmv_scs *s, int64_t stride, int64_t lines, uspmv_scs *r, std::vector<int32_t> *row_map);   // host/tlc_plan.cpp (measurement aid)

int uspmv_scs_layout(const uspmv_coo *m, int64_t C, int64_t sigma, int dtype, const int32_t *fixed_permutation,
                     uspmv_scs *s, std::vector<int64_t> *row_start, const char *who);   // host/scs_convert.cpp

namespace uspmv {
int fail(int status, const char *fmt, ...);  // records the thread-local error text, returns status
// false for the layout-only structs uspmv_convert_to_scs_device hands back (entries live on the device only)
inline bool scs_has_entries(const uspmv_scs *s) { return (int64_t)s->col_idxs.size() == s->n_elements; }
// host side of uspmv_dist_check (host/dist_check.cpp)
double check_x(int64_t global_col);
int64_t check_col(int64_t j, const int32_t *wsa, int P, int rank, bool loopback);
int dist_reference_rows(const uspmv_coo *local, const int32_t *wsa, int P, int rank, bool loopback, int dtype, void *y_ref);
}
