// Device-side builder of the column-window sweep plan (what host/sweep_plan.cpp derives on the CPU), for handles whose SCS arrays
// exist only in HBM (uspmv_dmat_wrap around a harness' own device arrays, uspmv_convert_to_scs_device): the function-pointer
// launchers of include/uspmv_launchers.hpp reach scs_spmv_sweep through it without a host struct and without copying the matrix
// to the host.  Two kernels per struct, O(n_tiles) work on the host in between (which tiles sweep, offsets):
//   sweep_scan : lane <-> row.  Effective row length (trailing +0 padding on one repeated column stripped, sweep_plan.cpp:38-48),
//                the padding column, whether the row's window index ever decreases or a window holds more than 255 of its entries;
//                per 64-row group: entries, lowest / highest window, any bad row -- one wave reduction.
//   sweep_fill : one wave per 64-row group of a sweep tile.  Count bytes per (window, row), then the compacted entry stream by the
//                very traversal the compute kernel performs (csrc/sweep_kernels.hip: windows ascending, rounds, one ballot per round,
//                a lane's element at base + active lanes below it) -- writing where that one reads, so the layout agrees by construction.
// The arrays are bit-identical to the host planner's (tests/test_gpu_sweep.py compares them).
#include "uspmv_device.hpp"

using namespace uspmv_dev;

namespace {

__device__ __forceinline__ bool is_pos_zero(double v) { return __double_as_longlong(v) == 0ll; }
__device__ __forceinline__ bool is_pos_zero(float v) { return __float_as_int(v) == 0; }

__device__ __forceinline__ unsigned lanes_below_m(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

template <typename VT>
__global__ void __launch_bounds__(256) sweep_scan(const long n_chunks, const int C, const int wlog, const int *__restrict__ chunk_ptrs,
        const int *__restrict__ chunk_lengths, const int *__restrict__ col_idxs, const VT *__restrict__ values,
        int *__restrict__ row_le, int *__restrict__ row_pad, int *__restrict__ grp /* [n_groups][4]: entries, lo window, hi window, bad */,
        int *__restrict__ max_col) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x, n_pad = n_chunks * C;
    int le = 0, pad = -1, lo = INT32_MAX, hi = -1, bad = 0, mc = 0;
    if (q < n_pad) {
        const long c = q / C;
        const int i = (int)(q - c * C);
        const long cs = chunk_ptrs[c];
        const int L = chunk_lengths[c];
        if (L > 0) {
            const int pc = col_idxs[cs + (long)(L - 1) * C + i];
            le = L;
            while (le > 0 && col_idxs[cs + (long)(le - 1) * C + i] == pc && is_pos_zero(values[cs + (long)(le - 1) * C + i])) --le;
            if (le < L) { pad = pc; mc = pc; }
            int prev = -1, run = 0;
            for (int j = 0; j < le; ++j) {
                const int col = col_idxs[cs + (long)j * C + i];
                const int sw = col >> wlog;
                mc = max(mc, col);
                if (sw < prev) { bad = 1; break; }
                run = sw == prev ? run + 1 : 1;
                if (run > 255) { bad = 1; break; }
                prev = sw;
                lo = min(lo, sw); hi = max(hi, sw);
            }
        }
        row_le[q] = le; row_pad[q] = pad;
    }
    int n = le;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        n += __shfl_xor(n, d);
        lo = min(lo, __shfl_xor(lo, d)); hi = max(hi, __shfl_xor(hi, d));
        bad |= __shfl_xor(bad, d); mc = max(mc, __shfl_xor(mc, d));
    }
    if ((threadIdx.x & 63) == 0) {
        const long g = q >> 6;
        if (g * 64 < n_pad) { grp[g * 4 + 0] = n; grp[g * 4 + 1] = lo; grp[g * 4 + 2] = hi; grp[g * 4 + 3] = bad; }
        if (mc > 0) atomicMax(max_col, mc);
    }
}

template <typename VT>
__global__ void __launch_bounds__(256) sweep_fill(const long n_chunks, const int C, const int wlog, const int R, const long n_groups_sweep,
        const int *__restrict__ chunk_ptrs, const int *__restrict__ col_idxs, const VT *__restrict__ values,
        const int *__restrict__ tile_ids, const int *__restrict__ t_smin, const int *__restrict__ t_S, const unsigned long long *__restrict__ t_cnt_off,
        const unsigned *__restrict__ wave_off, const int *__restrict__ row_le, const int *__restrict__ row_pad,
        unsigned char *__restrict__ cnt, VT *__restrict__ vals, unsigned short *__restrict__ idx, int *__restrict__ pad_col) {
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);             // (sweep tile, 64-row group)
    if (w >= n_groups_sweep) return;
    const int wpt = R >> 6;
    const long k = w / wpt;
    const int v = (int)(w - k * wpt);
    const long n_pad = n_chunks * C;
    const long q = (long)tile_ids[k] * R + v * 64 + lane;
    const int lo = t_smin[k], nS = t_S[k];
    const bool valid = q < n_pad;
    const int le = valid ? row_le[q] : 0;
    long cs = 0;
    int i = 0;
    if (valid) { const long c = q / C; i = (int)(q - c * C); cs = chunk_ptrs[c]; }
    pad_col[k * R + v * 64 + lane] = valid ? row_pad[q] : -1;
    unsigned char *my = cnt + t_cnt_off[k] + (v * 64 + lane);             // + sw * R
    // ---- counts: the row's entries per window (windows ascend inside a row: every cell is written once, by its own lane)
    {
        int prev = -1, run = 0;
        for (int j = 0; j < le; ++j) {
            const int sw = (col_idxs[cs + (long)j * C + i] >> wlog) - lo;
            if (sw != prev) { if (prev >= 0) my[(long)prev * R] = (unsigned char)run; prev = sw; run = 0; }
            ++run;
        }
        if (prev >= 0) my[(long)prev * R] = (unsigned char)run;
    }
    __threadfence_block();
    // ---- the compacted stream: windows ascending, rounds, lanes ascending inside a round
    unsigned out = wave_off[w];
    int j = 0;
    for (int sw = 0; sw < nS; ++sw) {
        const int c = valid ? (int)my[(long)sw * R] : 0;
        for (int kk = 0;; ++kk) {
            const unsigned long long m = __ballot(kk < c);
            if (m == 0ull) break;
            if (kk < c) {
                const unsigned off = out + lanes_below_m(m);
                const long src = cs + (long)j * C + i;
                vals[off] = values[src];
                idx[off] = (unsigned short)(col_idxs[src] - ((lo + sw) << wlog));
                ++j;
            }
            out += (unsigned)__popcll(m);
        }
    }
}

}  // namespace

namespace uspmv_dev {

int launch_sweep_scan(const uspmv_dmat *A, int wlog, int *d_row_le, int *d_row_pad, int *d_grp, int *d_max_col, hipStream_t st) {
    const long n_pad = A->n_chunks * A->C;
    if (n_pad == 0) return USPMV_OK;
    const dim3 grid((unsigned)((n_pad + 255) / 256)), block(256);
    if (A->dtype == USPMV_F64)
        hipLaunchKernelGGL(sweep_scan<double>, grid, block, 0, st, (long)A->n_chunks, (int)A->C, wlog, A->chunk_ptrs, A->chunk_lengths, A->col_idxs,
                           (const double *)A->values, d_row_le, d_row_pad, d_grp, d_max_col);
    else
        hipLaunchKernelGGL(sweep_scan<float>, grid, block, 0, st, (long)A->n_chunks, (int)A->C, wlog, A->chunk_ptrs, A->chunk_lengths, A->col_idxs,
                           (const float *)A->values, d_row_le, d_row_pad, d_grp, d_max_col);
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

int launch_sweep_fill(const uspmv_dmat *A, int wlog, int R, long n_sweep_tiles, const int *d_tile_ids, const int *d_smin, const int *d_S,
                      const unsigned long long *d_cnt_off, const unsigned *d_wave_off, const int *d_row_le, const int *d_row_pad,
                      unsigned char *d_cnt, void *d_vals, unsigned short *d_idx, int *d_pad_col, hipStream_t st) {
    const long n_groups = n_sweep_tiles * (R / 64);
    if (n_groups == 0) return USPMV_OK;
    const dim3 grid((unsigned)((n_groups + 3) / 4)), block(256);
    if (A->dtype == USPMV_F64)
        hipLaunchKernelGGL(sweep_fill<double>, grid, block, 0, st, (long)A->n_chunks, (int)A->C, wlog, R, n_groups, A->chunk_ptrs, A->col_idxs,
                           (const double *)A->values, d_tile_ids, d_smin, d_S, d_cnt_off, d_wave_off, d_row_le, d_row_pad, d_cnt, (double *)d_vals, d_idx, d_pad_col);
    else
        hipLaunchKernelGGL(sweep_fill<float>, grid, block, 0, st, (long)A->n_chunks, (int)A->C, wlog, R, n_groups, A->chunk_ptrs, A->col_idxs,
                           (const float *)A->values, d_tile_ids, d_smin, d_S, d_cnt_off, d_wave_off, d_row_le, d_row_pad, d_cnt, (float *)d_vals, d_idx, d_pad_col);
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

}  // namespace uspmv_dev
