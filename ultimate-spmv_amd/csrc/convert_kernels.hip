// convert_to_scs ON THE DEVICE from device-resident COO arrays: a matrix that lives in HBM becomes a SELL-C-sigma
// handle without a trip through host memory.  Reference: convert_to_scs, code/utilities.hpp:1842-2104 --
//   row populations          :1901-1903   -> run_bounds_kernel (one pass over the row indices: the start of every row's run) + a difference
//   sigma-window ordering    :1930-1941   -> (a) USPMV_SORT_HOST: the reference's std::sort on the identical pair type and comparator, on the
//                                            HOST, over the O(n_rows) count array only (4 bytes per row down, 4 up): bit-exact incl. the tie
//                                            order of the unstable sort; (b) USPMV_SORT_DEVICE_STABLE: a stable rank per window on the device
//                                            (rows of equal length keep their original order) -- chunk_lengths / chunk_ptrs and y in original
//                                            row order come out bit-identical, the order of equal-length rows inside a window differs
//   chunk lengths            :1949-1966   -> chunk_max_kernel (max over the C rows of a chunk in the new order)
//   chunk pointers           :1949-1966   -> scan_* kernels (exclusive scan of C * length in 64-bit, overflow reported as in :1959-1962)
//   permutations             :1976-1982, :2060-2069 -> perm_kernel
//   scatter                  :2013-2036   -> scs_fill_kernel (uspmv_api.hip), permute_scs_cols (:1802-1831) folded in
// The fixed-permutation form (:1911-1928: the sp struct of an ap[dp_sp] pair takes the dp struct's permutation) skips the ordering.
#include "uspmv_device.hpp"

#include <utility>
#include <vector>

using namespace uspmv_dev;

namespace uspmv_dev {
int launch_scs_fill(int dtype, long nnz, int C, int n_rows, const int *I, const int *J, const double *V, const int *row_start, const int *row_map,
                    const int *perm, const int *chunk_ptrs, int *col_idxs, void *values, hipStream_t st);   // uspmv_api.hip
}

namespace {

// start[r] = first entry of row r (= lower bound of r in the row-sorted index array) for every r in [0, n_rows], written by the thread
// that sits on the boundary between two runs; unsorted input raises the flag
__global__ void run_bounds_kernel(const int *__restrict__ I, const long nnz, const int n_rows, int *__restrict__ start, int *__restrict__ flag) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > nnz) return;
    const int prev = k == 0 ? -1 : I[k - 1];
    const int cur = k == nnz ? n_rows : I[k];
    if (k < nnz && (cur < 0 || cur >= n_rows)) { atomicOr(flag, 2); return; }
    if (cur < prev) { atomicOr(flag, 1); return; }
    for (int r = prev + 1; r <= cur; ++r) start[r] = (int)k;
}

// Stable rank of a window's rows by descending length: new position of row i = #(rows of the window with a longer row) + #(rows before
// i with the same length).  One workgroup per window, lengths staged in LDS, O(sigma^2 / 256) compares per thread.  Rows >= n_rows (the
// padding of the last chunk) take part with length 0, as in the reference's rl array (code/utilities.hpp:1895-1899).
__global__ void __launch_bounds__(256) window_rank_kernel(const int *__restrict__ start, const int n_rows, const long n_pad, const int sigma,
                                                          int *__restrict__ new_pos, int *__restrict__ len_new) {
    extern __shared__ int wl[];
    const long b = (long)blockIdx.x * sigma;
    const int w = (int)min((long)sigma, n_pad - b);
    for (int i = threadIdx.x; i < w; i += 256) { const long r = b + i; wl[i] = r < n_rows ? start[r + 1] - start[r] : 0; }
    __syncthreads();
    for (int i = threadIdx.x; i < w; i += 256) {
        const int me = wl[i];
        int rank = 0;
        for (int j = 0; j < w; ++j) { const int o = wl[j]; rank += (o > me) || (o == me && j < i); }
        new_pos[b + i] = (int)(b + rank);
        len_new[b + rank] = me;
    }
}

// the caller's order of rows: len_new[pos[i]] = length of row i (rows without a position -- i >= n_rows -- keep length 0 where they sit)
__global__ void place_lengths_kernel(const int *__restrict__ start, const int *__restrict__ pos, const int n_rows, const long n_pad, int *__restrict__ len_new, int *__restrict__ flag) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const int p = pos[i];
    if (p < 0 || p >= n_pad) { atomicOr(flag, 4); return; }
    len_new[p] = start[i + 1] - start[i];
}

// fixed permutation: the slots of the padding rows are re-zeroed AFTER the move (loop order of code/utilities.hpp:1913-1923) ...
__global__ void zero_tail_kernel(int *__restrict__ len_new, const int n_rows, const long n_pad) {
    const long i = n_rows + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_pad) len_new[i] = 0;
}
// ... and a non-empty row that now sits in a shorter chunk would overrun it in the scatter (the reference does, :1919-1922): refused
__global__ void check_fixed_kernel(const int *__restrict__ start, const int *__restrict__ pos, const int *__restrict__ chunk_lengths, const int n_rows, const int C, int *__restrict__ flag) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    if (start[i + 1] - start[i] > chunk_lengths[pos[i] / C]) atomicOr(flag, 8);
}

__global__ void chunk_max_kernel(const int *__restrict__ len_new, const long n_chunks, const int C, int *__restrict__ chunk_lengths, long *__restrict__ elems) {
    const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    int mx = 0;
    for (int i = 0; i < C; ++i) mx = max(mx, len_new[c * C + i]);
    chunk_lengths[c] = mx;
    elems[c] = (long)mx * C;
}

// exclusive scan of `in` (n longs) in three passes: per block of 1024 (scan_block_kernel), over the block sums (one workgroup walking them
// 1024 at a time), and the add (scan_add_kernel), which also narrows to the 32-bit chunk_ptrs of the reference (IT = int)
__global__ void __launch_bounds__(256) scan_block_kernel(const long *__restrict__ in, const long n, long *__restrict__ out, long *__restrict__ sums) {
    __shared__ long s[256];
    const long base = (long)blockIdx.x * 1024 + threadIdx.x * 4;
    long v[4], t = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) { v[u] = base + u < n ? in[base + u] : 0; t += v[u]; }
    s[threadIdx.x] = t;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const long a = threadIdx.x >= o ? s[threadIdx.x - o] : 0;
        __syncthreads();
        s[threadIdx.x] += a;
        __syncthreads();
    }
    long run = s[threadIdx.x] - t;
#pragma unroll
    for (int u = 0; u < 4; ++u) { if (base + u < n) out[base + u] = run; run += v[u]; }
    if (threadIdx.x == 255) sums[blockIdx.x] = s[255];
}
__global__ void __launch_bounds__(1024) scan_sums_kernel(long *__restrict__ sums, const long nb, long *__restrict__ total) {
    __shared__ long s[1024];
    __shared__ long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (long b = 0; b < nb; b += 1024) {
        const long i = b + threadIdx.x;
        const long v = i < nb ? sums[i] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const long a = threadIdx.x >= o ? s[threadIdx.x - o] : 0;
            __syncthreads();
            s[threadIdx.x] += a;
            __syncthreads();
        }
        if (i < nb) sums[i] = carry + s[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += s[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ void scan_add_kernel(const long *__restrict__ part, const long *__restrict__ sums, const long *__restrict__ total, const long n, int *__restrict__ ptrs) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ptrs[i] = (int)(part[i] + sums[i >> 10]);
    if (i == n) ptrs[n] = (int)*total;
}

// old_to_new / new_to_old of the struct (code/utilities.hpp:1976-1982, :2060-2069); identity for a fixed-permutation struct (:1915-1920)
__global__ void perm_kernel(const int *__restrict__ new_pos, const int n_rows, const bool identity, int *__restrict__ o2n, int *__restrict__ n2o) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const int p = identity ? (int)i : new_pos[i];
    o2n[i] = p;
    if (p < n_rows) n2o[p] = (int)i;                       // (the reference writes out of bounds otherwise; such slots stay 0)
}

struct Bufs {   // scoped device scratch
    std::vector<void *> p;
    ~Bufs() { for (void *q : p) (void)hipFree(q); }
    template <typename T> hipError_t get(T **out, size_t n) {
        void *q = nullptr;
        const hipError_t e = hipMalloc(&q, std::max<size_t>(n * sizeof(T), 16));
        if (e == hipSuccess) { p.push_back(q); *out = (T *)q; }
        return e;
    }
};

}  // namespace

extern "C" int uspmv_convert_to_scs_device_from_arrays(const int32_t *d_I, const int32_t *d_J, const double *d_V, int64_t n_rows, int64_t n_cols,
                                                       int64_t nnz, int64_t C, int64_t sigma, int dtype, const int32_t *d_fixed_permutation,
                                                       int permute_cols, int sort_mode, void *stream, uspmv_scs_t **layout, int32_t *d_old_to_new,
                                                       int32_t *d_new_to_old, uspmv_dmat_t **out) {
    const char *who = "uspmv_convert_to_scs_device_from_arrays";
    if (!out || n_rows < 1 || nnz < 0 || (nnz > 0 && (!d_I || !d_J || !d_V))) return uspmv::fail(USPMV_ERR_INVALID, "%s: bad argument", who);
    if (C < 1 || sigma < 1) return uspmv::fail(USPMV_ERR_INVALID, "%s: C and sigma must be >= 1", who);
    if (dtype != USPMV_F64 && dtype != USPMV_F32) return uspmv::fail(USPMV_ERR_INVALID, "%s: unknown dtype %d", who, dtype);
    if (sort_mode != USPMV_SORT_HOST && sort_mode != USPMV_SORT_DEVICE_STABLE) return uspmv::fail(USPMV_ERR_INVALID, "%s: unknown sort mode %d", who, sort_mode);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { (void)hipGetLastError(); return uspmv::fail(USPMV_ERR_NO_DEVICE, "%s: no HIP device", who); }
    const int64_t n_chunks = (n_rows + C - 1) / C, n_pad = n_chunks * C;
    if (n_pad > INT32_MAX || nnz > INT32_MAX) return uspmv::fail(USPMV_ERR_OVERFLOW, "%s: padded rows / entries exceed int32", who);
    if (sort_mode == USPMV_SORT_DEVICE_STABLE && !d_fixed_permutation && std::min<int64_t>(sigma, n_pad) > 8192)
        return uspmv::fail(USPMV_ERR_UNSUPPORTED, "%s: the device-side stable ordering takes sorting scopes of up to 8192 rows (sigma = %lld): use USPMV_SORT_HOST", who, (long long)sigma);
    hipStream_t st = (hipStream_t)stream;
    Bufs B;
    int *start = nullptr, *flag = nullptr, *new_pos = nullptr, *len_new = nullptr;
    long *elems = nullptr, *part = nullptr, *sums = nullptr, *total = nullptr;
    const long nb = (long)((n_chunks + 1023) / 1024);
    HIP_TRY(B.get(&start, (size_t)n_rows + 1));
    HIP_TRY(B.get(&flag, 1));
    HIP_TRY(B.get(&new_pos, (size_t)n_pad));
    HIP_TRY(B.get(&len_new, (size_t)n_pad));
    HIP_TRY(B.get(&elems, (size_t)n_chunks));
    HIP_TRY(B.get(&part, (size_t)n_chunks));
    HIP_TRY(B.get(&sums, (size_t)nb));
    HIP_TRY(B.get(&total, 1));
    HIP_TRY(hipMemsetAsync(flag, 0, 4, st));
    HIP_TRY(hipMemsetAsync(len_new, 0, 4 * (size_t)n_pad, st));

    // ---- row populations (code/utilities.hpp:1901-1903)
    hipLaunchKernelGGL(run_bounds_kernel, dim3((unsigned)((nnz + 1 + 255) / 256)), dim3(256), 0, st, d_I, (long)nnz, (int)n_rows, start, flag);
    HIP_TRY(hipGetLastError());

    // ---- where every row goes
    const int32_t *row_map = d_fixed_permutation;           // (what the scatter uses: the caller's permutation, else the ordering's)
    std::vector<int32_t> h_o2n;
    if (d_fixed_permutation) {
        hipLaunchKernelGGL(place_lengths_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, start, d_fixed_permutation, (int)n_rows, (long)n_pad, len_new, flag);
        if (n_pad > n_rows) hipLaunchKernelGGL(zero_tail_kernel, dim3((unsigned)((n_pad - n_rows + 255) / 256)), dim3(256), 0, st, len_new, (int)n_rows, (long)n_pad);
        HIP_TRY(hipGetLastError());
    } else if (sort_mode == USPMV_SORT_DEVICE_STABLE) {
        const int sg = (int)std::min<int64_t>(sigma, n_pad);
        const unsigned nwin = (unsigned)((n_pad + sg - 1) / sg);
        hipLaunchKernelGGL(window_rank_kernel, dim3(nwin), dim3(256), (size_t)sg * 4, st, start, (int)n_rows, (long)n_pad, sg, new_pos, len_new);
        HIP_TRY(hipGetLastError());
        row_map = new_pos;
    } else {
        // the reference's own ordering: std::sort on std::pair<long,long>{row, length} with the comparator a.second > b.second, window by
        // window (code/utilities.hpp:1930-1941), on the host over the COUNTS only -- 4 bytes per row cross the bus each way
        std::vector<int32_t> hs((size_t)n_rows + 1);
        HIP_TRY(hipMemcpyAsync(hs.data(), start, 4 * ((size_t)n_rows + 1), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        using row_len = std::pair<long, long>;
        std::vector<row_len> rl((size_t)(n_pad + sigma));
        for (int64_t i = 0; i < n_pad; ++i) { rl[(size_t)i].first = i; rl[(size_t)i].second = i < n_rows ? hs[(size_t)i + 1] - hs[(size_t)i] : 0; }
        const int64_t n_win = (n_pad + sigma - 1) / sigma;
#pragma omp parallel for schedule(dynamic, 64)
        for (int64_t w = 0; w < n_win; ++w) {
            const int64_t b = w * sigma, e = std::min(b + sigma, n_pad);
            std::sort(rl.begin() + b, rl.begin() + e, [](const row_len &a, const row_len &b2) { return a.second > b2.second; });
        }
        std::vector<int32_t> pos((size_t)n_pad), ln((size_t)n_pad);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n_pad; ++i) { pos[(size_t)rl[(size_t)i].first] = (int32_t)i; ln[(size_t)i] = (int32_t)rl[(size_t)i].second; }
        HIP_TRY(hipMemcpyAsync(new_pos, pos.data(), 4 * (size_t)n_pad, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(len_new, ln.data(), 4 * (size_t)n_pad, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));                   // (pos / ln go out of scope)
        row_map = new_pos;
    }

    // ---- chunk lengths and pointers (code/utilities.hpp:1949-1966)
    struct Guard { uspmv_dmat *A; ~Guard() { if (A) uspmv_dmat_free(A); } } guard{new uspmv_dmat};
    uspmv_dmat *A = guard.A;
    A->C = C; A->n_chunks = n_chunks; A->dtype = dtype; A->owns = true; A->n_store = (long)n_pad;
    int *cl = nullptr, *cp = nullptr;
    hipError_t e = hipMalloc((void **)&cl, 4 * (size_t)n_chunks);
    if (e == hipSuccess) e = hipMalloc((void **)&cp, 4 * ((size_t)n_chunks + 1));
    A->chunk_lengths = cl; A->chunk_ptrs = cp;
    if (e != hipSuccess) { return uspmv::fail(USPMV_ERR_ALLOC, "%s: %s", who, hipGetErrorString(e)); }
    hipLaunchKernelGGL(chunk_max_kernel, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, st, len_new, (long)n_chunks, (int)C, cl, elems);
    hipLaunchKernelGGL(scan_block_kernel, dim3((unsigned)nb), dim3(256), 0, st, elems, (long)n_chunks, part, sums);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, st, sums, nb, total);
    hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)((n_chunks + 1 + 255) / 256)), dim3(256), 0, st, part, sums, total, (long)n_chunks, cp);
    if (d_fixed_permutation)
        hipLaunchKernelGGL(check_fixed_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, start, d_fixed_permutation, cl, (int)n_rows, (int)C, flag);
    long h_total = 0;
    int h_flag = 0;
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&h_total, total, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(&h_flag, flag, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { return uspmv::fail(USPMV_ERR_HIP, "%s: %s", who, hipGetErrorString(e)); }
    if (h_flag & 2) { return uspmv::fail(USPMV_ERR_INVALID, "%s: a row index lies outside [0, n_rows)", who); }
    if (h_flag & 1) { return uspmv::fail(USPMV_ERR_UNSUPPORTED, "%s: COO entries must be sorted by row (uspmv_read_mtx and the generators produce that order)", who); }
    if (h_flag & 4) return uspmv::fail(USPMV_ERR_INVALID, "%s: fixed_permutation has an entry outside [0, n_rows_padded)", who);
    if (h_flag & 8) return uspmv::fail(USPMV_ERR_INVALID, "%s: fixed_permutation maps a non-empty row onto a padded slot (the reference overruns its chunk here, code/utilities.hpp:1919-1922)", who);
    if (h_total > INT32_MAX) { return uspmv::fail(USPMV_ERR_OVERFLOW, "%s: chunk_ptrs exceed the 32-bit index type", who); }   // (:1959-1962)
    A->n_elements = h_total;

    // ---- permutations, then the scatter (padding: value 0, column 0 -- mapped by permute_scs_cols like any local column, :1820-1826)
    int *o2n = d_old_to_new, *n2o = d_new_to_old;
    if (!o2n) HIP_TRY(B.get(&o2n, (size_t)n_rows));
    if (!n2o) HIP_TRY(B.get(&n2o, (size_t)n_rows));
    HIP_TRY(hipMemsetAsync(n2o, 0, 4 * (size_t)n_rows, st));
    hipLaunchKernelGGL(perm_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, new_pos, (int)n_rows, d_fixed_permutation != nullptr, o2n, n2o);
    const size_t vsz = dtype == USPMV_F64 ? 8 : 4, ne = (size_t)std::max<int64_t>(h_total, 1);
    void *ci = nullptr, *va = nullptr;
    e = hipMalloc(&ci, 4 * ne);
    if (e == hipSuccess) e = hipMalloc(&va, vsz * ne);
    A->col_idxs = (const int32_t *)ci; A->values = va;
    int pad_col = 0;
    if (e == hipSuccess && permute_cols) e = hipMemcpyAsync(&pad_col, o2n, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && permute_cols) e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipMemsetD32Async((hipDeviceptr_t)ci, pad_col, ne, st);
    if (e == hipSuccess) e = hipMemsetAsync(va, 0, vsz * ne, st);
    if (e != hipSuccess) { return uspmv::fail(USPMV_ERR_ALLOC, "%s: %s", who, hipGetErrorString(e)); }
    if (nnz > 0)
        if (int rc = launch_scs_fill(dtype, (long)nnz, (int)C, (int)n_rows, d_I, d_J, d_V, start, row_map, permute_cols ? o2n : nullptr, cp, (int *)ci, va, st)) { return rc; }

    // ---- optional host struct without entries (meta data, chunk arrays, permutations), as uspmv_convert_to_scs_device returns it
    if (layout) {
        auto *s = new uspmv_scs;
        s->C = C; s->sigma = sigma; s->n_rows = n_rows; s->n_cols = n_cols; s->nnz = nnz; s->n_chunks = n_chunks; s->n_rows_padded = n_pad; s->dtype = dtype;
        s->n_elements = h_total;
        s->chunk_lengths.resize((size_t)n_chunks); s->chunk_ptrs.resize((size_t)n_chunks + 1);
        s->old_to_new_idx.resize((size_t)n_rows); s->new_to_old_idx.resize((size_t)n_rows);
        e = hipMemcpyAsync(s->chunk_lengths.data(), cl, 4 * (size_t)n_chunks, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(s->chunk_ptrs.data(), cp, 4 * ((size_t)n_chunks + 1), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(s->old_to_new_idx.data(), o2n, 4 * (size_t)n_rows, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(s->new_to_old_idx.data(), n2o, 4 * (size_t)n_rows, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { delete s; return uspmv::fail(USPMV_ERR_HIP, "%s: %s", who, hipGetErrorString(e)); }
        *layout = s;
    }
    e = hipStreamSynchronize(st);                            // (the scratch of this call is released on return)
    if (e != hipSuccess) { if (layout) { delete *layout; *layout = nullptr; } return uspmv::fail(USPMV_ERR_HIP, "%s: %s", who, hipGetErrorString(e)); }
    guard.A = nullptr;
    *out = A;
    return USPMV_OK;
}
