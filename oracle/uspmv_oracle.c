/*
 * TEST INFRASTRUCTURE ONLY -- the parity oracle.  Never linked into, imported by or shipped
 * with the product (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load liboracle.so, and only as the checker).
 *
 * Plain-C restatement of the CPU hot path of RRZE-HPC/Ultimate-SpMV.  Each function cites the
 * reference file:line it follows (paths relative to /root/reference/).  PINNED: every function
 * here is checked against the genuine reference (oracle/_ref, built from the reference's own
 * sources by oracle/Makefile) and against the golden vectors in tests/golden/ that were
 * generated from it by oracle/make_golden.py -- see tests/test_oracle_vs_golden.py.
 *
 * Floating point: the reference is built with g++ -O3 -march=native, i.e. GNU's default
 * -ffp-contract=fast turns every `tmp += a * b` of the kernels into one fused multiply-add per
 * element, evaluated in slot order j = 0,1,2,...  This file is compiled with
 * -ffp-contract=off and spells those FMAs out with fma()/fmaf(), so the result does not depend
 * on compiler flags; equality with the real reference build is bit-for-bit (tested).
 *
 * Deliberate difference: orc_convert_to_scs sorts each sigma window with a STABLE descending
 * sort, whereas the reference uses std::sort (unstable, code/utilities.hpp:1936-1940).  All
 * tie-independent outputs (chunk_lengths, chunk_ptrs, n_elements, y in original row order)
 * are identical; the exact tie order of the reference is pinned by the golden vectors instead.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * SELL-C-sigma SpMV, any C.   code/kernels.hpp:159-211 (spmv_omp_scs) and :216-258
 * (scs_impl_cpu<C>): per chunk, tmp[i] = 0; for j < chunk_lengths[c]: for i < C:
 * tmp[i] += values[cs + j*C + i] * x[col_idxs[cs + j*C + i]];  y[c*C + i] = tmp[i].
 * ---------------------------------------------------------------------------------------- */
#define DEF_SPMV_SCS(SUF, VT, FMA)                                                            \
    void orc_spmv_scs_##SUF(long C, long n_chunks, const int *chunk_ptrs,                     \
                            const int *chunk_lengths, const int *col_idxs, const VT *values,  \
                            const VT *x, VT *y) {                                             \
        _Pragma("omp parallel for schedule(static)")                                          \
        for (long c = 0; c < n_chunks; ++c) {                                                 \
            long cs = chunk_ptrs[c];                                                          \
            for (long i = 0; i < C; ++i) {                                                    \
                VT tmp = 0;                                                                   \
                for (long j = 0; j < chunk_lengths[c]; ++j) {                                 \
                    long k = cs + j * C + i;                                                  \
                    tmp = FMA(values[k], x[col_idxs[k]], tmp);                                \
                }                                                                             \
                y[c * C + i] = tmp;                                                           \
            }                                                                                 \
        }                                                                                     \
    }
DEF_SPMV_SCS(f64, double, fma)
DEF_SPMV_SCS(f32, float, fmaf)

/* CRS SpMV.  code/kernels.hpp:22-63 (spmv_omp_csr).  The reference's inner loop carries
 * `#pragma omp simd simdlen(SIMD_LENGTH)` (:49), so its summation order is compiler-chosen
 * (partial sums per SIMD lane); this restatement sums strictly left to right and is compared
 * with a tolerance, not bit-for-bit. */
#define DEF_SPMV_CSR(SUF, VT, FMA)                                                            \
    void orc_spmv_csr_##SUF(long n_rows, const int *row_ptrs, const int *col_idxs,            \
                            const VT *values, const VT *x, VT *y) {                           \
        _Pragma("omp parallel for schedule(static)")                                          \
        for (long r = 0; r < n_rows; ++r) {                                                   \
            VT sum = 0;                                                                       \
            for (long j = row_ptrs[r]; j < row_ptrs[r + 1]; ++j)                              \
                sum = FMA(values[j], x[col_idxs[j]], sum);                                    \
            y[r] = sum;                                                                       \
        }                                                                                     \
    }
DEF_SPMV_CSR(f64, double, fma)
DEF_SPMV_CSR(f32, float, fmaf)

/* SELL-C-sigma SpMMV (block vectors), any C.  code/kernels.hpp:306-398
 * (block_spmv_omp_scs_general).  colwise (:352,:372): X[col + v*ld], Y[(c*C+i) + v*ld];
 * rowwise (:358,:378): X[col*b + v], Y[(c*C+i)*b + v].  This -- not block_scs_impl_cpu, whose
 * accumulator is never zeroed (code/kernels.hpp:434) -- is the oracle for SpMMV. */
#define DEF_SPMMV_SCS(SUF, VT, FMA)                                                           \
    void orc_spmmv_scs_##SUF(long C, long n_chunks, const int *chunk_ptrs,                    \
                             const int *chunk_lengths, const int *col_idxs, const VT *values, \
                             const VT *X, VT *Y, int b, long ld, int rowwise) {               \
        _Pragma("omp parallel for schedule(static)")                                          \
        for (long c = 0; c < n_chunks; ++c) {                                                 \
            long cs = chunk_ptrs[c];                                                          \
            for (long i = 0; i < C; ++i) {                                                    \
                for (int v = 0; v < b; ++v) {                                                 \
                    VT tmp = 0;                                                               \
                    for (long j = 0; j < chunk_lengths[c]; ++j) {                             \
                        long k = cs + j * C + i;                                              \
                        long col = col_idxs[k];                                               \
                        VT xv = rowwise ? X[col * b + v] : X[col + (long)v * ld];             \
                        tmp = FMA(values[k], xv, tmp);                                        \
                    }                                                                         \
                    if (rowwise) Y[(c * C + i) * b + v] = tmp;                                \
                    else Y[(c * C + i) + (long)v * ld] = tmp;                                 \
                }                                                                             \
            }                                                                                 \
        }                                                                                     \
    }
DEF_SPMMV_SCS(f64, double, fma)
DEF_SPMMV_SCS(f32, float, fmaf)

/* CRS SpMMV.  code/kernels.hpp:68-154 (block_spmv_omp_csr). */
#define DEF_SPMMV_CSR(SUF, VT, FMA)                                                           \
    void orc_spmmv_csr_##SUF(long n_rows, const int *row_ptrs, const int *col_idxs,           \
                             const VT *values, const VT *X, VT *Y, int b, long ld,            \
                             int rowwise) {                                                   \
        _Pragma("omp parallel for schedule(static)")                                          \
        for (long r = 0; r < n_rows; ++r) {                                                   \
            for (int v = 0; v < b; ++v) {                                                     \
                VT tmp = 0;                                                                   \
                for (long j = row_ptrs[r]; j < row_ptrs[r + 1]; ++j) {                        \
                    long col = col_idxs[j];                                                   \
                    VT xv = rowwise ? X[col * b + v] : X[col + (long)v * ld];                 \
                    tmp = FMA(values[j], xv, tmp);                                            \
                }                                                                             \
                if (rowwise) Y[r * b + v] = tmp;                                              \
                else Y[r + (long)v * ld] = tmp;                                               \
            }                                                                                 \
        }                                                                                     \
    }
DEF_SPMMV_CSR(f64, double, fma)
DEF_SPMMV_CSR(f32, float, fmaf)

/* Adaptive precision dp+sp, compile-time-C variant.  code/ap_kernels.hpp:24-82
 * (scs_ap_impl_cpu<C>): dp part accumulated in double from dp_x (:61); sp part: float value
 * times DOUBLE dp_x accumulated in double (:68); y = dp_tmp + sp_tmp (:74). */
void orc_spmv_scs_ap_adv(long C, long n_chunks, const int *dp_cp, const int *dp_cl,
                         const int *dp_ci, const double *dp_va, const int *sp_cp,
                         const int *sp_cl, const int *sp_ci, const float *sp_va,
                         const double *dp_x, double *dp_y) {
#pragma omp parallel for schedule(static)
    for (long c = 0; c < n_chunks; ++c) {
        long dcs = dp_cp[c], scs = sp_cp[c];
        for (long i = 0; i < C; ++i) {
            double dt = 0.0, st = 0.0;
            for (long j = 0; j < dp_cl[c]; ++j) {
                long k = dcs + j * C + i;
                dt = fma(dp_va[k], dp_x[dp_ci[k]], dt);
            }
            for (long j = 0; j < sp_cl[c]; ++j) {
                long k = scs + j * C + i;
                st = fma((double)sp_va[k], dp_x[sp_ci[k]], st);
            }
            dp_y[c * C + i] = dt + st;
        }
    }
}

/* Adaptive precision dp+sp, generic-C variant.  code/ap_kernels.hpp:562-634
 * (spmv_omp_scs_ap): differs from the _adv variant in reading the FLOAT vector sp_x for the sp
 * part (:621) -- the float*float product is rounded to float before it is widened and added. */
void orc_spmv_scs_ap(long C, long n_chunks, const int *dp_cp, const int *dp_cl, const int *dp_ci,
                     const double *dp_va, const int *sp_cp, const int *sp_cl, const int *sp_ci,
                     const float *sp_va, const double *dp_x, const float *sp_x, double *dp_y) {
#pragma omp parallel for schedule(static)
    for (long c = 0; c < n_chunks; ++c) {
        long dcs = dp_cp[c], scs = sp_cp[c];
        for (long i = 0; i < C; ++i) {
            double dt = 0.0, st = 0.0;
            for (long j = 0; j < dp_cl[c]; ++j) {
                long k = dcs + j * C + i;
                dt = fma(dp_va[k], dp_x[dp_ci[k]], dt);
            }
            for (long j = 0; j < sp_cl[c]; ++j) {
                long k = scs + j * C + i;
                float p = sp_va[k] * sp_x[sp_ci[k]];
                st = st + (double)p;
            }
            dp_y[c * C + i] = dt + st;
        }
    }
}

/* Adaptive precision CRS dp+sp.  code/ap_kernels.hpp:144-223 (spmv_omp_csr_apdpsp): both
 * partial sums in double, sp values times dp_x (:204), y = dp_sum + sp_sum (:215). */
void orc_spmv_csr_apdpsp(long n_rows, const int *dp_rp, const int *dp_ci, const double *dp_va,
                         const int *sp_rp, const int *sp_ci, const float *sp_va,
                         const double *dp_x, double *dp_y) {
#pragma omp parallel for schedule(static)
    for (long r = 0; r < n_rows; ++r) {
        double dt = 0.0, st = 0.0;
        for (long j = dp_rp[r]; j < dp_rp[r + 1]; ++j) dt = fma(dp_va[j], dp_x[dp_ci[j]], dt);
        for (long j = sp_rp[r]; j < sp_rp[r + 1]; ++j)
            st = fma((double)sp_va[j], dp_x[sp_ci[j]], st);
        dp_y[r] = dt + st;
    }
}

/* out[i] = in[perm[i]].  code/utilities.hpp:1768-1782 (apply_permutation). */
void orc_apply_permutation_f64(double *out, const double *in, const int *perm, long n) {
    for (long i = 0; i < n; ++i) out[i] = in[perm[i]];
}
void orc_apply_permutation_f32(float *out, const float *in, const int *perm, long n) {
    for (long i = 0; i < n; ++i) out[i] = in[perm[i]];
}

/* col = perm[col] for local columns (col < n_rows), halo columns untouched.
 * code/utilities.hpp:1802-1831 (permute_scs_cols). */
void orc_permute_scs_cols(long n_elements, long n_rows, int *col_idxs, const int *perm) {
    for (long i = 0; i < n_elements; ++i)
        if (col_idxs[i] < n_rows) col_idxs[i] = perm[col_idxs[i]];
}

/* Halo send-buffer gather, one neighbour, colwise single-vector mode:
 * send[i] = x[perm[send_idxs[i]] + block_offset].  code/classes_structs.hpp:813-818
 * (SpmvKernel::pack_send_buf); CUDA twin code/kernels.hpp:554-577. */
void orc_pack_send_buf_f64(double *send, const double *x, const int *perm, const int *send_idxs,
                           long n, long block_offset) {
    for (long i = 0; i < n; ++i) send[i] = x[perm[send_idxs[i]] + block_offset];
}
void orc_pack_send_buf_f32(float *send, const float *x, const int *perm, const int *send_idxs,
                           long n, long block_offset) {
    for (long i = 0; i < n; ++i) send[i] = x[perm[send_idxs[i]] + block_offset];
}

/* ------------------------------------------------------------------------------------------
 * COO -> SELL-C-sigma.  code/utilities.hpp:1842-2104 (convert_to_scs).
 *   rows counted (:1901-1903); each window of sigma padded rows sorted by descending row
 *   length (:1930-1941, STABLE here -- see header) or reordered by `fixed_perm` (:1911-1928);
 *   chunk length = longest row of the chunk (:1949-1966); old_to_new (:1976-1982); padding =
 *   value 0 / column 0 (:1991-2002); entries scattered column-major per chunk, COO order kept
 *   inside a row (:2013-2036); inverse permutation (:2060-2069).
 * Two-call protocol: call with chunk_ptrs/chunk_lengths/old_to_new/new_to_old allocated
 * (n_chunks+1, n_chunks, n_rows, n_rows) and col_idxs == NULL to obtain n_elements (return
 * value); call again with col_idxs/values (as double; caller narrows for sp) to fill.
 * ---------------------------------------------------------------------------------------- */
typedef struct { long row, len; } row_len_t;

static void stable_sort_desc(row_len_t *a, long n, row_len_t *tmp) {
    if (n < 2) return;
    long h = n / 2;
    stable_sort_desc(a, h, tmp);
    stable_sort_desc(a + h, n - h, tmp);
    long i = 0, j = h, k = 0;
    while (i < h && j < n) tmp[k++] = (a[j].len > a[i].len) ? a[j++] : a[i++];
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, sizeof(row_len_t) * (size_t)n);
}

long orc_convert_to_scs(long n_rows, long nnz, const int *I, const int *J, const double *vals,
                        long C, long sigma, const int *fixed_perm, int *chunk_ptrs,
                        int *chunk_lengths, int *old_to_new, int *new_to_old, int *col_idxs,
                        double *values) {
    long n_chunks = (n_rows + C - 1) / C;
    long n_pad = n_chunks * C;
    row_len_t *rl = (row_len_t *)calloc((size_t)(n_pad + 1), sizeof(row_len_t));
    row_len_t *tmp = (row_len_t *)calloc((size_t)(n_pad + 1), sizeof(row_len_t));
    for (long i = 0; i < n_pad; ++i) rl[i].row = i;
    for (long k = 0; k < nnz; ++k) rl[I[k]].len++;
    if (fixed_perm) {
        for (long i = 0; i < n_pad; ++i) { tmp[i].row = rl[i].row; tmp[i].len = 0; }
        for (long i = 0; i < n_pad; ++i) {
            if (i < n_rows) tmp[fixed_perm[i]].len = rl[i].len;
            else tmp[i].len = rl[i].len;
        }
        memcpy(rl, tmp, sizeof(row_len_t) * (size_t)n_pad);
    } else {
        for (long i = 0; i < n_pad; i += sigma) {
            long e = (i + sigma < n_pad) ? i + sigma : n_pad;
            stable_sort_desc(rl + i, e - i, tmp);
        }
    }
    long cur = 0;
    for (long c = 0; c < n_chunks; ++c) {
        long mx = 0;
        for (long i = 0; i < C; ++i) if (rl[c * C + i].len > mx) mx = rl[c * C + i].len;
        chunk_lengths[c] = (int)mx;
        chunk_ptrs[c] = (int)cur;
        cur += mx * C;
    }
    chunk_ptrs[n_chunks] = (int)cur;
    /* NB with fixed_perm the reference keeps `.first` = identity (:1915,:1920), so the stored
     * old_to_new / new_to_old of such a struct are the identity although the entries are
     * placed with fixed_perm (:2017-2019). */
    for (long i = 0; i < n_pad; ++i)
        if (rl[i].row < n_rows) old_to_new[rl[i].row] = (int)i;
    for (long i = 0; i < n_rows; ++i) new_to_old[old_to_new[i]] = (int)i;
    if (col_idxs) {
        for (long k = 0; k < cur; ++k) { col_idxs[k] = 0; values[k] = 0.0; }
        long *fill = (long *)calloc((size_t)n_pad, sizeof(long));
        for (long k = 0; k < nnz; ++k) {
            long row = fixed_perm ? fixed_perm[I[k]] : old_to_new[I[k]];
            long idx = chunk_ptrs[row / C] + fill[row] * C + row % C;
            col_idxs[idx] = J[k];
            values[idx] = vals[k];
            fill[row]++;
        }
        free(fill);
    }
    free(rl); free(tmp);
    return cur;
}

/* dp/sp split by magnitude: |v| >= th -> dp else sp, COO order kept.
 * code/utilities.hpp:2899-2911 (partition_precisions, non-equilibrated ap[dp_sp] branch).
 * is_dp[k] = 1/0; returns the dp count. */
long orc_partition_precisions_dpsp(long nnz, const double *vals, double threshold,
                                   unsigned char *is_dp) {
    long n = 0;
    for (long k = 0; k < nnz; ++k) { is_dp[k] = fabs(vals[k]) >= threshold; n += is_dp[k]; }
    return n;
}

/* 1-D row partition.  code/mpi_funcs.hpp:446-493 (+ the empty-last-rank fix-up :602-606).
 * method 0 = seg-rows, 1 = seg-nnz.  I = row index of every COO entry (sorted by row). */
/* equilibrate_matrix (code/utilities.hpp:2667-2685): values /= largest |value| of their row (:2610-2626,
 * :2646-2654), then /= largest |value| of their column in the row-scaled matrix (:2628-2644, :2656-2664).
 * The reference sizes both scratch vectors with n_cols (:2670, :2678), so n_rows <= n_cols is assumed. */
void orc_equilibrate_matrix(long n_rows, long n_cols, long nnz, const int *I, const int *J, double *vals) {
    long n = n_rows > n_cols ? n_rows : n_cols;
    double *mx = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
    for (long k = 0; k < nnz; ++k) { double a = fabs(vals[k]); if (a > mx[I[k]]) mx[I[k]] = a; }
    for (long k = 0; k < nnz; ++k) vals[k] = vals[k] / mx[I[k]];
    memset(mx, 0, (size_t)(n > 0 ? n : 1) * sizeof(double));
    for (long k = 0; k < nnz; ++k) { double a = fabs(vals[k]); if (a > mx[J[k]]) mx[J[k]] = a; }
    for (long k = 0; k < nnz; ++k) vals[k] = vals[k] / mx[J[k]];
    free(mx);
}

void orc_seg_work_sharing_arr(int method, long n_rows, long nnz, const int *I, int P, int *wsa) {
    wsa[0] = 0;
    if (method == 0) {
        long per = n_rows / P;
        for (int s = 1; s <= P; ++s) wsa[s] = (int)(s * per);
        wsa[P] = I[nnz - 1] + 1;
    } else {
        long per = nnz / P, local = 0;
        int seg = 1;
        for (long g = 0; g < nnz; ++g) {
            if (local == per) { wsa[seg++] = I[g] + 1; local = 0; continue; }
            ++local;
        }
        wsa[P] = I[nnz - 1] + 1;
    }
    if (wsa[P - 1] == wsa[P]) for (int r = 1; r < P; ++r) wsa[r] -= 1;
}

/* Remote-column discovery and column compression for one rank.
 * code/mpi_funcs.hpp:242-415 (collect_local_needed_heri).  col_idxs (global columns, in SCS
 * storage order incl. padding entries with column 0) is rewritten in place: local columns ->
 * col - wsa[rank] (:329); remote columns -> n_local + [all lower-rank owners ascending, then
 * all higher-rank owners ascending] + first-seen order inside one owner (:357-401).
 * recv_idxs (capacity: number of distinct remote columns, <= n_cols) receives, grouped by
 * owner rank ascending, the owner-local row ids in first-seen order (:297,:318);
 * recv_counts[p] their number; recv_cumsum[P+1] as built at :403-414.  Returns halo size. */
long orc_collect_local_needed_heri(long n_elements, int *col_idxs, const int *wsa, int rank,
                                   int P, long n_cols, int *recv_idxs, int *recv_counts,
                                   int *recv_cumsum) {
    int lo = wsa[rank], hi = wsa[rank + 1];
    int *owner_of = NULL;
    int *slot = (int *)malloc(sizeof(int) * (size_t)n_cols);     /* first-seen rank inside owner */
    for (long c = 0; c < n_cols; ++c) slot[c] = -1;
    int *cnt = (int *)calloc((size_t)P, sizeof(int));
    /* counting pass: first-seen order per owner */
    int **lists = (int **)calloc((size_t)P, sizeof(int *));
    int *cap = (int *)calloc((size_t)P, sizeof(int));
    for (long k = 0; k < n_elements; ++k) {
        int col = col_idxs[k];
        if (col >= lo && col < hi) continue;
        if (slot[col] >= 0) continue;
        int p = 0;
        while (!(col >= wsa[p] && col < wsa[p + 1])) ++p;
        if (cnt[p] == cap[p]) {
            cap[p] = cap[p] ? 2 * cap[p] : 64;
            lists[p] = (int *)realloc(lists[p], sizeof(int) * (size_t)cap[p]);
        }
        lists[p][cnt[p]] = col - wsa[p];
        slot[col] = cnt[p]++;
    }
    /* offsets: owners below `rank` first (ascending), then owners above */
    long *base = (long *)calloc((size_t)P + 1, sizeof(long));
    long run = 0;
    for (int p = 0; p < P; ++p) { base[p] = run; run += cnt[p]; }   /* cnt[rank] == 0 */
    long n_local = hi - lo;
    /* assignment pass */
    for (long k = 0; k < n_elements; ++k) {
        int col = col_idxs[k];
        if (col >= lo && col < hi) { col_idxs[k] = col - lo; continue; }
        int p = 0;
        while (!(col >= wsa[p] && col < wsa[p + 1])) ++p;
        col_idxs[k] = (int)(n_local + base[p] + slot[col]);
    }
    long w = 0;
    for (int p = 0; p < P; ++p) {
        recv_counts[p] = cnt[p];
        for (int i = 0; i < cnt[p]; ++i) recv_idxs[w++] = lists[p][i];
        free(lists[p]);
    }
    /* recv_counts_cumsum exactly as the reference assembles it (:403-414) */
    {
        long *lhs = (long *)calloc((size_t)P + 1, sizeof(long));
        long *rhs = (long *)calloc((size_t)P + 1, sizeof(long));
        for (int p = 1; p <= P; ++p) {
            lhs[p] = lhs[p - 1] + ((p - 1) < rank ? cnt[p - 1] : 0);
            rhs[p] = rhs[p - 1] + ((p - 1) > rank ? cnt[p - 1] : 0);
        }
        for (int p = 0; p <= P; ++p) recv_cumsum[p] = 0;
        for (int p = 0; p < rank; ++p) recv_cumsum[p] = (int)lhs[p];
        int limit = (P == 1) ? P - (rank + 1) : P - rank + 1;
        for (int i = 0; i < limit; ++i) recv_cumsum[rank + i] = (int)(lhs[rank] + rhs[rank + i]);
        free(lhs); free(rhs);
    }
    free(lists); free(cap); free(cnt); free(slot); free(base); (void)owner_of;
    return run;
}
