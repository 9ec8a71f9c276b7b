// Device-side builder of the PHASED block plan (what host/tlc_plan.cpp derives on the CPU: uspmv_scs_reorder_rows mode 1 without a
// permutation + uspmv_build_phased_plan), for handles whose SCS arrays exist only in HBM (uspmv_dmat_wrap, the function-pointer
// launchers): uspmv_dmat_optimize_block_device then needs no copy of the matrix -- indices or values -- on the host.
//   block_reorder_window : one workgroup per window of 16 chunks.  Inside runs of equal-length chunks the rows are sorted by (first column,
//                          row) -- for locally numbered matrices the original row order the sigma sort scrambled -- by one bitonic sort
//                          of (run, first column, row) keys in LDS: row_map[plan row] = the caller's row.
//   block_phase_plan     : one 256-thread workgroup per 64-row tile, thread <-> (row, slot % 4).  The greedy segmentation of the host
//                          planner, group by group: the phase's distinct X rows live in an LDS hash set; a group's entries that are not
//                          in it are de-duplicated through a second set and counted ("fresh"); the phase closes when it would exceed
//                          `cap` rows or `ngp` groups.  COUNT pass: phases and list entries per tile.  WRITE pass (after an exclusive
//                          scan of those on the host, O(n_tiles)): the sorted X-row list of every phase (bitonic sort in LDS), its first
//                          group, and the phase-local one-byte index of every entry (binary search in the sorted list).
// Same sets, same rule, sorted lists: the arrays equal the host planner's bit for bit (tests/test_gpu_parity.py compares digests).
#include "uspmv_device.hpp"
#include <algorithm>

using namespace uspmv_dev;

namespace {

constexpr int HCAP = 1024;          // hash-set capacity (a phase holds <= 256 rows, a group adds <= 256)
constexpr int EMPTY = -1;

__device__ __forceinline__ unsigned hslot(int key) { return ((unsigned)key * 2654435761u) >> 22; }   // top 10 bits

__device__ __forceinline__ bool hset_find(const int *H, int key) {
    unsigned s = hslot(key);
    for (;;) {
        const int k = H[s];
        if (k == key) return true;
        if (k == EMPTY) return false;
        s = (s + 1) & (HCAP - 1);
    }
}
// true: this thread inserted the key (it was absent)
__device__ __forceinline__ bool hset_insert(int *H, int key) {
    unsigned s = hslot(key);
    for (;;) {
        const int old = atomicCAS(&H[s], EMPTY, key);
        if (old == EMPTY) return true;
        if (old == key) return false;
        s = (s + 1) & (HCAP - 1);
    }
}

// ascending bitonic sort of N (power of two) keys in LDS by blockDim.x >= N / 2 ... here: one element per thread, N == blockDim.x
template <typename K>
__device__ void bitonic_sort(K *a, int N) {
    const int t = threadIdx.x;
    for (int k = 2; k <= N; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            const int p = t ^ j;
            if (t < N && p > t) {
                const K x = a[t], y = a[p];
                const bool up = (t & k) == 0;
                if ((x > y) == up) { a[t] = y; a[p] = x; }
            }
        }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) block_reorder_window(const long n_chunks, const int C, const int *__restrict__ chunk_ptrs,
        const int *__restrict__ chunk_lengths, const int *__restrict__ col_idxs, int *__restrict__ row_map, int *__restrict__ changed) {
    __shared__ unsigned long long keys[1024];
    __shared__ int run_of[16];
    const int W = 16 * C;                                   // rows per window = blockDim.x
    const long c0 = (long)blockIdx.x * 16;
    const int nch = (int)min((long)16, n_chunks - c0);
    if (threadIdx.x == 0) {
        int run = 0;
        for (int k = 0; k < 16; ++k) {
            if (k > 0 && k < nch && chunk_lengths[c0 + k] != chunk_lengths[c0 + k - 1]) ++run;
            run_of[k] = run;
        }
    }
    __syncthreads();
    const int t = threadIdx.x, ck = t / C, i = t - ck * C;
    unsigned long long key = ~0ull;
    if (ck < nch) {
        const long c = c0 + ck;
        const unsigned first = chunk_lengths[c] > 0 ? (unsigned)col_idxs[(long)chunk_ptrs[c] + i] : 0xFFFFFFFFu;   // empty rows last, by row
        key = ((unsigned long long)run_of[ck] << 42) | ((unsigned long long)first << 10) | (unsigned long long)t;
    }
    keys[t] = key;
    if (W < 1024) for (int k = W + t; k < 1024; k += W) keys[k] = ~0ull;
    bitonic_sort(keys, W);
    if (ck < nch) {
        const int src = (int)(keys[t] & 1023ull);
        row_map[c0 * C + t] = (int)(c0 * C) + src;
        if (src != t) atomicOr(changed, 1);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
template <bool WRITE>
__global__ void __launch_bounds__(256) block_phase_plan(const long n_chunks, const int C, const int cap, const int ngp, const int *__restrict__ chunk_ptrs,
        const int *__restrict__ chunk_lengths, const int *__restrict__ col_idxs, const int *__restrict__ row_map, const unsigned *__restrict__ c16_ptrs,
        int *__restrict__ t_phases, int *__restrict__ t_list,                 // COUNT: per tile out.  WRITE: per tile in (exclusive scans)
        int *__restrict__ ph_g0, int *__restrict__ ph_list_ptr, int *__restrict__ xrows, unsigned char *__restrict__ col8, int *__restrict__ max_rows) {
    __shared__ int H[HCAP], G[HCAP];
    __shared__ int cur[256], fresh_list[256], sorted[256];
    __shared__ int s_nfresh, s_ng;
    const long tile = blockIdx.x;
    const int tid = threadIdx.x, i = tid >> 2, u = tid & 3;
    const long p = tile * 64 + i;                            // plan row
    const long c = p / C;
    const int li = (int)(p - c * C);
    const bool row_ok = c < n_chunks;
    int L = 0;
    long src_base = 0;
    unsigned q0 = 0;
    if (row_ok) {
        L = chunk_lengths[c];
        const long sr = row_map ? (long)row_map[p] : p;      // the caller's row behind this plan row (same chunk length)
        const long sc = sr / C;
        src_base = (long)chunk_ptrs[sc] + (sr - sc * C);
        q0 = c16_ptrs[c];
    }
    if (tid == 0) s_ng = 0;
    for (int k = tid; k < HCAP; k += 256) H[k] = EMPTY;
    __syncthreads();
    if (u == 0 && row_ok) atomicMax(&s_ng, (L + 3) >> 2);
    __syncthreads();
    const int ng = s_ng;
    if (ng == 0) { if (!WRITE && tid == 0) { t_phases[tile] = 0; t_list[tile] = 0; } return; }
    int ncur = 0, first = 0, n_ph = 0, n_list = 0;
    const int ph_base = WRITE ? t_phases[tile] : 0, list_base = WRITE ? t_list[tile] : 0;

    auto emit = [&](int g_first, int g_end) {                // the phase [g_first, g_end) with the ncur rows of cur[]
        if (WRITE) {
            sorted[tid] = tid < ncur ? cur[tid] : INT32_MAX;
            bitonic_sort(sorted, 256);
            if (tid < ncur) xrows[list_base + n_list + tid] = sorted[tid];
            if (tid == 0) { ph_g0[ph_base + n_ph] = g_first; ph_list_ptr[ph_base + n_ph] = list_base + n_list; atomicMax(max_rows, ncur); }
            for (int gg = g_first; gg < g_end; ++gg) {
                const int j = gg * 4 + u;
                if (row_ok && j < L) {
                    const int col = col_idxs[src_base + (long)j * C];
                    int lo = 0, hi = ncur - 1;
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (sorted[mid] < col) lo = mid + 1; else hi = mid; }
                    col8[(long)q0 + (long)gg * 4 * C + li * 4 + u] = (unsigned char)lo;
                }
            }
            __syncthreads();
        }
        ++n_ph; n_list += ncur;
    };

    for (int g = 0; g < ng; ++g) {
        const int j = g * 4 + u;
        const bool valid = row_ok && j < L;
        const int col = valid ? col_idxs[src_base + (long)j * C] : 0;
        for (int attempt = 0; attempt < 2; ++attempt) {
            // ---- the group's entries that the phase does not hold yet, de-duplicated
            for (int k = tid; k < HCAP; k += 256) G[k] = EMPTY;
            if (tid == 0) s_nfresh = 0;
            __syncthreads();
            if (valid && !hset_find(H, col) && hset_insert(G, col)) fresh_list[atomicAdd(&s_nfresh, 1)] = col;
            __syncthreads();
            const int fresh = s_nfresh;
            if (g > first && (ncur + fresh > cap || g - first >= ngp)) {     // close the phase before this group and look at the group again
                emit(first, g);
                for (int k = tid; k < HCAP; k += 256) H[k] = EMPTY;
                ncur = 0; first = g;
                __syncthreads();
                continue;
            }
            if (tid < fresh) { hset_insert(H, fresh_list[tid]); cur[ncur + tid] = fresh_list[tid]; }
            ncur += fresh;
            __syncthreads();
            break;
        }
    }
    emit(first, ng);
    if (!WRITE && tid == 0) { t_phases[tile] = n_ph; t_list[tile] = n_list; }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Two-part SpMMV (uspmv_dist_spmmv): which 64-row tiles of the phased plan touch an X row >= n_local (a halo row), and the two
// chunk-length arrays in which the other part's chunks carry USPMV_SKIP_LEN.
__global__ void block_tile_class(const long n_tiles, const int *__restrict__ ph_ptr, const int *__restrict__ list_ptr, const int *__restrict__ xrows,
                                 const int n_local, unsigned char *__restrict__ flags) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const int p0 = ph_ptr[t], p1 = ph_ptr[t + 1];
    unsigned char f = 0;
    if (p1 > p0)
        for (int k = list_ptr[p0], e = list_ptr[p1]; k < e; ++k) f |= (unsigned char)(xrows[k] >= n_local);
    flags[t] = f;
}

__global__ void part_len_fill(const long n_chunks, const int C, const int rows_per_flag, const unsigned char *__restrict__ flags,
                              const int *__restrict__ chunk_lengths, int *__restrict__ len_int, int *__restrict__ len_bnd) {
    const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    // a chunk spans C / rows_per_flag flags (C >= rows_per_flag) or shares one with its neighbours
    unsigned char f = 0;
    const long r0 = c * C, r1 = r0 + C;
    for (long k = r0 / rows_per_flag; k * rows_per_flag < r1; ++k) f |= flags[k];
    const int L = chunk_lengths[c];
    len_int[c] = f ? USPMV_SKIP_LEN : L;
    len_bnd[c] = f ? L : USPMV_SKIP_LEN;
}

}  // namespace

namespace uspmv_dev {

int launch_block_tile_class(const uspmv_dmat *A, long n_local, unsigned char *d_flags, hipStream_t st) {
    if (A->pb_n_tiles == 0) return USPMV_OK;
    hipLaunchKernelGGL(block_tile_class, dim3((unsigned)((A->pb_n_tiles + 255) / 256)), dim3(256), 0, st, (long)A->pb_n_tiles, A->pb_ph_ptr, A->pb_list_ptr,
                       A->pb_xrows, (int)std::min<long>(n_local, INT32_MAX), d_flags);
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

int launch_part_len_fill(const uspmv_dmat *A, int rows_per_flag, const unsigned char *d_flags, int *d_len_int, int *d_len_bnd, hipStream_t st) {
    if (A->n_chunks == 0) return USPMV_OK;
    hipLaunchKernelGGL(part_len_fill, dim3((unsigned)((A->n_chunks + 255) / 256)), dim3(256), 0, st, (long)A->n_chunks, (int)A->C, rows_per_flag, d_flags,
                       A->chunk_lengths, d_len_int, d_len_bnd);
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

int launch_block_reorder(const uspmv_dmat *A, int *d_row_map, int *d_changed, hipStream_t st) {
    const long nw = (A->n_chunks + 15) / 16;
    if (nw == 0) return USPMV_OK;
    hipLaunchKernelGGL(block_reorder_window, dim3((unsigned)nw), dim3((unsigned)(16 * A->C)), 0, st, (long)A->n_chunks, (int)A->C, A->chunk_ptrs,
                       A->chunk_lengths, A->col_idxs, d_row_map, d_changed);
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

int launch_block_phase_plan(const uspmv_dmat *A, bool write, int cap, int ngp, const int *d_row_map, const unsigned *d_c16_ptrs, int *d_t_phases,
                            int *d_t_list, int *d_ph_g0, int *d_ph_list_ptr, int *d_xrows, unsigned char *d_col8, int *d_max_rows, hipStream_t st) {
    const long n_tiles = (A->n_chunks * A->C + 63) / 64;
    if (n_tiles == 0) return USPMV_OK;
    if (write)
        hipLaunchKernelGGL(block_phase_plan<true>, dim3((unsigned)n_tiles), dim3(256), 0, st, (long)A->n_chunks, (int)A->C, cap, ngp, A->chunk_ptrs, A->chunk_lengths,
                           A->col_idxs, d_row_map, d_c16_ptrs, d_t_phases, d_t_list, d_ph_g0, d_ph_list_ptr, d_xrows, d_col8, d_max_rows);
    else
        hipLaunchKernelGGL(block_phase_plan<false>, dim3((unsigned)n_tiles), dim3(256), 0, st, (long)A->n_chunks, (int)A->C, cap, ngp, A->chunk_ptrs, A->chunk_lengths,
                           A->col_idxs, d_row_map, d_c16_ptrs, d_t_phases, d_t_list, d_ph_g0, d_ph_list_ptr, d_xrows, d_col8, d_max_rows);
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

}  // namespace uspmv_dev
