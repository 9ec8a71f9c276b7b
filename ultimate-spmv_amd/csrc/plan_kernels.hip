// Device-side builder of the tile-local-column plan (the structure host/tlc_plan.cpp derives on the CPU): for
// handles whose SCS arrays exist only in HBM (uspmv_dmat_wrap, uspmv_convert_to_scs_device).  Two sweeps over
// col_idxs, one 256-thread workgroup per tile (thread <-> row): the first counts a tile's distinct 16-element
// x lines with a bitmap in LDS, the second -- after an exclusive scan of those counts on the host (n_tiles
// integers) -- writes the sorted line list and the 16-bit local indices.  The bitmap covers 65 536 lines: the
// tile's whole line range when it fits, else up to 16 WINDOWS of 4 096 lines (65 536 columns) each, wherever they lie -- the tile's
// distinct windows are collected first (a 16-entry table in LDS, sorted), a line's bit is (its window's rank, line within the window),
// so bit order is still line order.  That covers tiles whose columns form a few clusters far apart: the padding column 0 next to a
// band anywhere in the matrix (permute_scs_cols(0)), and saddle-point (KKT) matrices whose rows reach into index ranges millions of
// columns apart (round 3: before, only one cluster at either end of the range was representable and the KKT matrix of
// uspmv_gen_kkt came out almost unstaged).  Tiles that need more than 16 windows are left unstaged (the host planner sorts those),
// which changes the plan, never y; everywhere else the plan is identical to the host planner's.
#include "uspmv_device.hpp"

using namespace uspmv_dev;

namespace {

constexpr int PLAN_BITS = 65536;            // lines a tile's bitmap covers
constexpr int PLAN_WORDS = PLAN_BITS / 32;
constexpr int WIN_SHIFT = 12, WIN_LINES = 1 << WIN_SHIFT, NWIN = PLAN_BITS / WIN_LINES;   // 16 windows of 4 096 lines

// line -> bit (or -1: not representable) and back.  win == nullptr: the tile's whole range [lo, lo + PLAN_BITS) is the bitmap;
// otherwise win[0 .. nwin) are the tile's window ids (line >> WIN_SHIFT), ascending
__device__ __forceinline__ int line_to_bit(int l, int lo, const int *win, int nwin) {
    if (!win) return l - lo;
    const int w = l >> WIN_SHIFT;
    for (int k = 0; k < nwin; ++k)
        if (win[k] == w) return (k << WIN_SHIFT) | (l & (WIN_LINES - 1));
    return -1;
}
__device__ __forceinline__ int bit_to_line(int b, int lo, const int *win) {
    return win ? (win[b >> WIN_SHIFT] << WIN_SHIFT) | (b & (WIN_LINES - 1)) : lo + b;
}

// Marks the lines of the tile in bits[]; returns (lo line, hi line, window table) through LDS scalars.  *s_hi < 0: empty tile.
// *s_nwin == 0: contiguous mode (the whole range fits the bitmap); > 0: window mode with s_win[0 .. *s_nwin) sorted.
// (chunk_ptrs2 != nullptr: a second struct with the same row layout -- the sp part of an ap[dp_sp] pair -- marks the same bitmap)
__device__ void tile_bitmap(const long n_chunks, const int C, const int *__restrict__ chunk_ptrs,
                            const int *__restrict__ chunk_lengths, const int *__restrict__ col_idxs, const long tile,
                            unsigned *bits, int *s_lo, int *s_hi, int *s_maxcol, int *s_bad, int *s_win, int *s_nwin,
                            const int rpt, const int *__restrict__ chunk_ptrs2 = nullptr, const int *__restrict__ chunk_lengths2 = nullptr,
                            const int *__restrict__ col_idxs2 = nullptr) {
    // a tile is rpt * 256 rows: thread <-> rows tile*rpt*256 + h*256 + threadIdx.x, h < rpt
    if (threadIdx.x == 0) { *s_lo = INT32_MAX; *s_hi = -1; *s_bad = 0; *s_nwin = 0; }
    if (threadIdx.x < NWIN) s_win[threadIdx.x] = -1;
    for (int w = threadIdx.x; w < PLAN_WORDS; w += 256) bits[w] = 0u;
    __syncthreads();
    // every entry of the tile, twice over: f(column)
    auto for_each_col = [&](auto &&f) {
        for (int h = 0; h < rpt; ++h) {
            const long row = (tile * rpt + h) * 256 + threadIdx.x;
            const long c = row / C;
            const int i = (int)(row - c * C);
            if (c >= n_chunks) continue;
            const int cs = chunk_ptrs[c], L = chunk_lengths[c];
            for (int j = 0; j < L; ++j) f(col_idxs[(long)cs + (long)j * C + i]);
            if (chunk_ptrs2) {
                const int cs2 = chunk_ptrs2[c], L2 = chunk_lengths2[c];
                for (int j = 0; j < L2; ++j) f(col_idxs2[(long)cs2 + (long)j * C + i]);
            }
        }
    };
    int lo = INT32_MAX, hi = -1, last_w = -1;
    bool over = false;
    for_each_col([&](int col) {
        lo = min(lo, col); hi = max(hi, col);
        const int w = col >> (4 + WIN_SHIFT);
        if (w != last_w) {                         // (consecutive entries of a row mostly share their window)
            last_w = w;
            bool placed = false;
            for (int k = 0; k < NWIN && !placed; ++k) {
                const int old = atomicCAS(&s_win[k], -1, w);
                placed = old == -1 || old == w;
            }
            over |= !placed;
        }
    });
    if (hi >= 0) { atomicMin(s_lo, lo >> 4); atomicMax(s_hi, hi >> 4); atomicMax(s_maxcol, hi); }
    if (over) *s_bad = 1;
    __syncthreads();
    const int tlo = *s_lo, thi = *s_hi;
    if (thi < 0) return;
    const bool contiguous = thi - tlo < PLAN_BITS;
    if (threadIdx.x == 0 && !contiguous) {         // sort the table (<= 16 entries), count it
        int n = 0;
        while (n < NWIN && s_win[n] >= 0) ++n;
        for (int a = 1; a < n; ++a) { const int v = s_win[a]; int b = a - 1; while (b >= 0 && s_win[b] > v) { s_win[b + 1] = s_win[b]; --b; } s_win[b + 1] = v; }
        *s_nwin = n;
    }
    if (threadIdx.x == 0 && contiguous) *s_bad = 0;  // (the window table may have overflowed on a range that fits as a whole)
    __syncthreads();
    if (*s_bad) return;
    const int nwin = *s_nwin;
    const int *win = contiguous ? nullptr : s_win;
    for_each_col([&](int col) {
        const int b = line_to_bit(col >> 4, tlo, win, nwin);
        atomicOr(&bits[b >> 5], 1u << (b & 31));
    });
    __syncthreads();
}

__global__ void __launch_bounds__(256) plan_count_lines(const long n_chunks, const int C, const int *__restrict__ chunk_ptrs,
        const int *__restrict__ chunk_lengths, const int *__restrict__ col_idxs, const int max_lines,
        int *__restrict__ n_lines, int *__restrict__ max_col, const int *__restrict__ chunk_ptrs2, const int *__restrict__ chunk_lengths2,
        const int *__restrict__ col_idxs2, const int rpt) {
    __shared__ unsigned bits[PLAN_WORDS];
    __shared__ int s_lo, s_hi, s_cnt, s_maxcol, s_bad, s_nwin, s_win[NWIN];
    if (threadIdx.x == 0) { s_cnt = 0; s_maxcol = 0; }
    const long tile = blockIdx.x;
    tile_bitmap(n_chunks, C, chunk_ptrs, chunk_lengths, col_idxs, tile, bits, &s_lo, &s_hi, &s_maxcol, &s_bad, s_win, &s_nwin, rpt, chunk_ptrs2, chunk_lengths2, col_idxs2);
    int n = 0;
    if (s_hi >= 0 && !s_bad) {
        int cnt = 0;
        for (int w = threadIdx.x; w < PLAN_WORDS; w += 256) cnt += __popc(bits[w]);
        atomicAdd(&s_cnt, cnt);
        __syncthreads();
        n = s_cnt <= max_lines ? s_cnt : 0;
    }
    if (threadIdx.x == 0) {
        n_lines[tile] = n;
        atomicMax(max_col, s_maxcol);
    }
}

__global__ void __launch_bounds__(256) plan_write(const long n_chunks, const int C, const int *__restrict__ chunk_ptrs,
        const int *__restrict__ chunk_lengths, const int *__restrict__ col_idxs, const int *__restrict__ tile_line_ptr,
        const unsigned *__restrict__ c16_ptrs, int *__restrict__ tile_lines, unsigned short *__restrict__ col16,
        const int *__restrict__ chunk_ptrs2, const int *__restrict__ chunk_lengths2, const int *__restrict__ col_idxs2,
        const unsigned *__restrict__ c16_ptrs2, unsigned short *__restrict__ col16_2, const int rpt) {
    __shared__ unsigned bits[PLAN_WORDS];
    __shared__ unsigned short rank0[PLAN_WORDS];   // set bits in the words before this one (< 4096)
    __shared__ int s_lo, s_hi, s_dummy, s_bad, s_nwin, s_win[NWIN];
    __shared__ int wsum[256];
    const long tile = blockIdx.x;
    const int lp0 = tile_line_ptr[tile];
    if (tile_line_ptr[tile + 1] == lp0) return;   // unstaged or empty tile: col16 stays zero, the kernel gathers
    if (threadIdx.x == 0) s_dummy = 0;
    tile_bitmap(n_chunks, C, chunk_ptrs, chunk_lengths, col_idxs, tile, bits, &s_lo, &s_hi, &s_dummy, &s_bad, s_win, &s_nwin, rpt, chunk_ptrs2, chunk_lengths2, col_idxs2);
    const int *win = s_nwin > 0 ? s_win : nullptr;
    const int nwin = s_nwin;
    // exclusive prefix of the popcounts: 8 consecutive words per thread, then a block scan of the 256 partial sums
    constexpr int WPT = PLAN_WORDS / 256;
    int part = 0;
    for (int k = 0; k < WPT; ++k) part += __popc(bits[threadIdx.x * WPT + k]);
    wsum[threadIdx.x] = part;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int v = threadIdx.x >= o ? wsum[threadIdx.x - o] : 0;
        __syncthreads();
        wsum[threadIdx.x] += v;
        __syncthreads();
    }
    int run = wsum[threadIdx.x] - part;
    for (int k = 0; k < WPT; ++k) {
        const int w = threadIdx.x * WPT + k;
        rank0[w] = (unsigned short)run;
        unsigned b = bits[w];
        int r = run;
        while (b) {                                 // the tile's sorted line list
            const int bit = __ffs(b) - 1;
            tile_lines[lp0 + r++] = bit_to_line(w * 32 + bit, s_lo, win);
            b &= b - 1;
        }
        run += __popc(bits[w]);
    }
    __syncthreads();
    const int tlo = s_lo;
    for (int h = 0; h < rpt; ++h) {
        const long row = (tile * rpt + h) * 256 + threadIdx.x;
        const long c = row / C;
        const int i = (int)(row - c * C);
        if (c >= n_chunks) continue;
        const int cs = chunk_ptrs[c], L = chunk_lengths[c];
        unsigned short *q = col16 + c16_ptrs[c];
        for (int j = 0; j < L; ++j) {
            const int col = col_idxs[(long)cs + (long)j * C + i];
            const int l = line_to_bit(col >> 4, tlo, win, nwin);
            const unsigned below = bits[l >> 5] & ((1u << (l & 31)) - 1u);
            const int pos = rank0[l >> 5] + __popc(below);
            q[(long)(j >> 2) * 4 * C + i * 4 + (j & 3)] = (unsigned short)((pos << 4) | (col & 15));
        }
        if (chunk_ptrs2) {
            const int cs2 = chunk_ptrs2[c], L2 = chunk_lengths2[c];
            unsigned short *q2 = col16_2 + c16_ptrs2[c];
            for (int j = 0; j < L2; ++j) {
                const int col = col_idxs2[(long)cs2 + (long)j * C + i];
                const int l = line_to_bit(col >> 4, tlo, win, nwin);
                const unsigned below = bits[l >> 5] & ((1u << (l & 31)) - 1u);
                const int pos = rank0[l >> 5] + __popc(below);
                q2[(long)(j >> 2) * 4 * C + i * 4 + (j & 3)] = (unsigned short)((pos << 4) | (col & 15));
            }
        }
    }
}

// uspmv_scs_rechunk32 (host/tlc_plan.cpp) on the device: new chunk k = old chunks [k*32/C, (k+1)*32/C), row order untouched;
// thread <-> row of the new struct, whose arrays were zero-filled before (padding = value 0, column 0)
template <typename VT>
__global__ void rechunk32_kernel(const long n_rows_old, const int C, const int *__restrict__ cp_old, const int *__restrict__ cl_old,
                                 const int *__restrict__ ci_old, const VT *__restrict__ va_old, const int *__restrict__ cp_new,
                                 int *__restrict__ ci_new, VT *__restrict__ va_new) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows_old) return;
    const long co = row / C;
    const int io = (int)(row - co * C);
    const long cs = cp_old[co], base = cp_new[row >> 5];
    const int L = cl_old[co], i = (int)(row & 31);
    for (int j = 0; j < L; ++j) {
        ci_new[base + (long)j * 32 + i] = ci_old[cs + (long)j * C + io];
        va_new[base + (long)j * 32 + i] = va_old[cs + (long)j * C + io];
    }
}

// 16-bit local indices -> 12-bit (uspmv_device.hpp: tlc_col12).  One thread per (chunk, row): the row's groups of four indices, two
// groups to three consecutive dwords, an odd last group to a dword + a ushort (in two planes of the chunk).
__global__ void __launch_bounds__(256) plan_pack12(const long n_chunks, const int C, const int *__restrict__ chunk_lengths,
        const unsigned *__restrict__ c16_ptrs, const unsigned short *__restrict__ col16, const unsigned *__restrict__ c12_ptrs, unsigned *__restrict__ col12) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long c = row / C;
    if (c >= n_chunks) return;
    const int i = (int)(row - c * C);
    const int L = chunk_lengths[c], ngt = (L + 3) >> 2, np = ngt >> 1;
    const unsigned short *q = col16 + c16_ptrs[c];
    unsigned *w = col12 + c12_ptrs[c];
    auto idx = [&](int g, int k) -> unsigned { return (unsigned)q[((long)g * C + i) * 4 + k] & 0xFFFu; };
    for (int p = 0; p < np; ++p) {
        unsigned long long lo = 0; unsigned hi = 0;              // 96 bits: index u at bits [12u, 12u + 12)
        for (int u = 0; u < 8; ++u) {
            const unsigned long long v = idx(2 * p + (u >> 2), u & 3);
            const int b = 12 * u;
            if (b < 64) { lo |= v << b; if (b + 12 > 64) hi |= (unsigned)(v >> (64 - b)); }
            else hi |= (unsigned)(v << (b - 64));
        }
        unsigned *t3 = w + ((long)p * C + i) * 3;
        t3[0] = (unsigned)lo; t3[1] = (unsigned)(lo >> 32); t3[2] = hi;
    }
    if (ngt & 1) {
        unsigned long long v48 = 0;
        for (int k = 0; k < 4; ++k) v48 |= (unsigned long long)idx(ngt - 1, k) << (12 * k);
        unsigned *t = w + (long)3 * np * C;
        t[i] = (unsigned)v48;
        ((unsigned short *)(t + C))[i] = (unsigned short)(v48 >> 32);
    }
}

}  // namespace

namespace uspmv_dev {

int launch_plan_count(const uspmv_dmat *A, long n_tiles, int max_lines, int *d_n_lines, int *d_max_col, hipStream_t st, const uspmv_dmat *A2, int tile_rows) {
    hipLaunchKernelGGL(plan_count_lines, dim3((unsigned)n_tiles), dim3(256), 0, st, (long)A->n_chunks, (int)A->C, A->chunk_ptrs,
                       A->chunk_lengths, A->col_idxs, max_lines, d_n_lines, d_max_col, A2 ? A2->chunk_ptrs : nullptr,
                       A2 ? A2->chunk_lengths : nullptr, A2 ? A2->col_idxs : nullptr, tile_rows / 256);
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

int launch_plan_pack12(const uspmv_dmat *A, const unsigned *d_c16_ptrs, const unsigned short *d_col16, const unsigned *d_c12_ptrs, unsigned *d_col12, hipStream_t st) {
    const long n_rows = (long)A->n_chunks * A->C;
    if (n_rows == 0) return USPMV_OK;
    hipLaunchKernelGGL(plan_pack12, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, (long)A->n_chunks, (int)A->C, A->chunk_lengths, d_c16_ptrs, d_col16,
                       d_c12_ptrs, d_col12);
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

int launch_plan_write(const uspmv_dmat *A, long n_tiles, const int *d_tile_line_ptr, const unsigned *d_c16_ptrs, int *d_tile_lines,
                      unsigned short *d_col16, hipStream_t st, const uspmv_dmat *A2, const unsigned *d_c16_ptrs2, unsigned short *d_col16_2, int tile_rows) {
    hipLaunchKernelGGL(plan_write, dim3((unsigned)n_tiles), dim3(256), 0, st, (long)A->n_chunks, (int)A->C, A->chunk_ptrs,
                       A->chunk_lengths, A->col_idxs, d_tile_line_ptr, d_c16_ptrs, d_tile_lines, d_col16, A2 ? A2->chunk_ptrs : nullptr,
                       A2 ? A2->chunk_lengths : nullptr, A2 ? A2->col_idxs : nullptr, d_c16_ptrs2, d_col16_2, tile_rows / 256);
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

// The block plan's private copy of the entries, gathered ON THE DEVICE from the handle's values under the plan's row map:
// plan row (chunk c, lane i) takes the entries of the caller's row row_map[c*C + i] (same chunk length: rows only move between chunks
// of equal length).  GROUP_MAJOR: the layout scs_spmmv_quadph streams ([chunk][group of four slots][row][slot % 4], at c16_ptrs[c]);
// otherwise column-major inside the chunk like the caller's arrays (the one-list-per-tile kernels).
template <typename VT, bool GROUP_MAJOR>
__global__ void __launch_bounds__(256) block_values_gather(const long n_chunks, const int C, const int *__restrict__ chunk_ptrs, const int *__restrict__ chunk_lengths,
        const VT *__restrict__ values, const int *__restrict__ row_map, const unsigned *__restrict__ c16_ptrs, VT *__restrict__ out) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    const long c = q / C;
    if (c >= n_chunks) return;
    const int i = (int)(q - c * C);
    const long src_row = row_map ? (long)row_map[q] : q;
    const long sc = src_row / C;
    const int si = (int)(src_row - sc * C);
    const long cs = chunk_ptrs[c], scs = chunk_ptrs[sc];
    const int L = chunk_lengths[c];
    const long base = GROUP_MAJOR ? (long)c16_ptrs[c] : cs;
    for (int j = 0; j < L; ++j) {
        const VT v = values[scs + (long)j * C + si];
        if (GROUP_MAJOR) out[base + (long)(j >> 2) * 4 * C + i * 4 + (j & 3)] = v;
        else out[base + (long)j * C + i] = v;
    }
}

int launch_block_values_gather(const uspmv_dmat *A, const int *d_row_map, const unsigned *d_c16_ptrs, void *d_out, bool group_major, hipStream_t st) {
    const long n_pad = A->n_chunks * A->C;
    if (n_pad == 0) return USPMV_OK;
    const dim3 grid((unsigned)((n_pad + 255) / 256)), block(256);
#define BVG(VT, GM) hipLaunchKernelGGL((block_values_gather<VT, GM>), grid, block, 0, st, (long)A->n_chunks, (int)A->C, A->chunk_ptrs, A->chunk_lengths, \
                                       (const VT *)A->values, d_row_map, d_c16_ptrs, (VT *)d_out)
    if (A->dtype == USPMV_F64) { if (group_major) BVG(double, true); else BVG(double, false); }
    else { if (group_major) BVG(float, true); else BVG(float, false); }
#undef BVG
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

int launch_rechunk32(const uspmv_dmat *A, const int *d_cp_new, int *d_ci_new, void *d_va_new, hipStream_t st) {
    const long n_rows = (long)(A->n_chunks * A->C);
    if (n_rows == 0) return USPMV_OK;
    const unsigned grid = grid_for(n_rows, 256);
    if (A->dtype == USPMV_F64)
        hipLaunchKernelGGL(rechunk32_kernel<double>, dim3(grid), dim3(256), 0, st, n_rows, (int)A->C, A->chunk_ptrs, A->chunk_lengths, A->col_idxs,
                           (const double *)A->values, d_cp_new, d_ci_new, (double *)d_va_new);
    else
        hipLaunchKernelGGL(rechunk32_kernel<float>, dim3(grid), dim3(256), 0, st, n_rows, (int)A->C, A->chunk_ptrs, A->chunk_lengths, A->col_idxs,
                           (const float *)A->values, d_cp_new, d_ci_new, (float *)d_va_new);
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

}  // namespace uspmv_dev
