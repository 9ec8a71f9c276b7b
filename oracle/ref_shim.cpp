// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or shipped with the product.
//
// extern "C" driver around the *genuine* reference implementation.  This file contains no
// reference code: it #includes the reference headers where they lie under
// /root/reference/code (see oracle/Makefile, target `ref`) and forwards plain-pointer calls
// to the reference's own templates.  The build output goes to oracle/_ref/ only.
//
// Used for two things:
//   1. oracle/make_golden.py : produce the golden vectors committed under tests/golden/
//      and check the C restatement in oracle/uspmv_oracle.c against the real thing;
//   2. bench.py cpu_baseline (kind "reference"): time the reference's own OpenMP kernel
//      (code/kernels.hpp:216-301) on the GPU box's host cores.
//
// Functions forwarded (reference file:line):
//   read_mtx                     code/utilities.hpp:2148-2309
//   convert_to_scs               code/utilities.hpp:1842-2104
//   permute_scs_cols             code/utilities.hpp:1802-1831
//   apply_permutation            code/utilities.hpp:1768-1782
//   partition_precisions         code/utilities.hpp:2810-3123
//   spmv_omp_csr                 code/kernels.hpp:22-63
//   block_spmv_omp_csr           code/kernels.hpp:68-154
//   spmv_omp_scs                 code/kernels.hpp:159-211
//   spmv_omp_scs_adv             code/kernels.hpp:265-301
//   block_spmv_omp_scs_general   code/kernels.hpp:306-398
//   spmv_omp_scs_ap_adv          code/ap_kernels.hpp:90-142
//   spmv_omp_scs_ap              code/ap_kernels.hpp:562-634
//   spmv_omp_csr_apdpsp          code/ap_kernels.hpp:144-223
//   random_init                  code/utilities.hpp:880-912
//   (USE_MPI build only)
//   seg_work_sharing_arr         code/mpi_funcs.hpp:424-622
//   seg_mtx_struct               code/mpi_funcs.hpp:636-674
//   localize_row_idx             code/mpi_funcs.hpp:862-877
//   collect_local_needed_heri    code/mpi_funcs.hpp:242-415

#include "mmio.h"
#include "utilities.hpp"
#include "kernels.hpp"
#include "ap_kernels.hpp"
#ifdef USE_MPI
#include "mpi_funcs.hpp"
#endif

#include <cstring>
#include <string>
#include <vector>

namespace {
template <typename T> void copy_out(const std::vector<T> &v, T *dst, long n) {
    if (dst && n > 0) std::memcpy(dst, v.data(), sizeof(T) * (size_t)n);
}
}  // namespace

extern "C" {

// ---------------------------------------------------------------- COO
void *ref_read_mtx(const char *path) {
    Config cfg;
    cfg.matrix_file_name = path;
    auto *m = new MtxData<double, int>;
    read_mtx(cfg, m, 0);
    return m;
}

void *ref_mtx_from_coo(long n_rows, long n_cols, long nnz, const int *I, const int *J,
                       const double *vals) {
    auto *m = new MtxData<double, int>;
    m->n_rows = n_rows; m->n_cols = n_cols; m->nnz = nnz;
    m->is_sorted = true; m->is_symmetric = false;
    m->I.assign(I, I + nnz); m->J.assign(J, J + nnz); m->values.assign(vals, vals + nnz);
    return m;
}

void ref_mtx_dims(void *h, long *out3) {
    auto *m = (MtxData<double, int> *)h;
    out3[0] = m->n_rows; out3[1] = m->n_cols; out3[2] = m->nnz;
}

void ref_mtx_arrays(void *h, int *I, int *J, double *vals) {
    auto *m = (MtxData<double, int> *)h;
    copy_out(m->I, I, m->nnz); copy_out(m->J, J, m->nnz); copy_out(m->values, vals, m->nnz);
}

void ref_mtx_free(void *h) { delete (MtxData<double, int> *)h; }

// -equilibrate 1 for a one-precision run: equilibrate_matrix (code/utilities.hpp:2667-2685), in place
void ref_equilibrate_matrix(void *h) { equilibrate_matrix<double, int>((MtxData<double, int> *)h); }

// float copy of a double COO exactly as compute_result<float,int> sees it
// (MtxData<VT,IT>::copy, code/classes_structs.hpp:1277-1299)
void *ref_mtx_to_f32(void *h) {
    auto *m = (MtxData<double, int> *)h;
    auto *f = new MtxData<float, int>;
    f->copy(*m);
    return f;
}
void ref_mtx_f32_free(void *h) { delete (MtxData<float, int> *)h; }

// ---------------------------------------------------------------- SELL-C-sigma
void *ref_convert_to_scs_f64(void *mtx, long C, long sigma, int *fixed_perm) {
    auto *s = new ScsData<double, int>;
    convert_to_scs<double, double, int>((MtxData<double, int> *)mtx, C, sigma, s, fixed_perm);
    return s;
}
void *ref_convert_to_scs_f32(void *mtx_f32, long C, long sigma, int *fixed_perm) {
    auto *s = new ScsData<float, int>;
    convert_to_scs<float, float, int>((MtxData<float, int> *)mtx_f32, C, sigma, s, fixed_perm);
    return s;
}

#define SCS_ACCESSORS(SUF, VT)                                                              \
    void ref_scs_meta_##SUF(void *h, long *o) {                                             \
        auto *s = (ScsData<VT, int> *)h;                                                    \
        o[0] = s->C; o[1] = s->sigma; o[2] = s->n_rows; o[3] = s->n_cols;                   \
        o[4] = s->n_rows_padded; o[5] = s->n_chunks; o[6] = s->n_elements; o[7] = s->nnz;   \
    }                                                                                       \
    void ref_scs_arrays_##SUF(void *h, int *cp, int *cl, int *ci, VT *va, int *o2n,         \
                              int *n2o) {                                                   \
        auto *s = (ScsData<VT, int> *)h;                                                    \
        copy_out(s->chunk_ptrs, cp, s->n_chunks + 1);                                       \
        copy_out(s->chunk_lengths, cl, s->n_chunks);                                        \
        copy_out(s->col_idxs, ci, s->n_elements);                                           \
        copy_out(s->values, va, s->n_elements);                                             \
        copy_out(s->old_to_new_idx, o2n, s->n_rows);                                        \
        if (n2o) std::memcpy(n2o, s->new_to_old_idx, sizeof(int) * (size_t)s->n_rows);      \
    }                                                                                       \
    void ref_permute_scs_cols_##SUF(void *h, int *perm) {                                   \
        permute_scs_cols<VT, int>((ScsData<VT, int> *)h, perm);                             \
    }                                                                                       \
    void ref_scs_free_##SUF(void *h) { delete (ScsData<VT, int> *)h; }

SCS_ACCESSORS(f64, double)
SCS_ACCESSORS(f32, float)

void ref_apply_permutation_f64(double *out, double *in, int *perm, int n) {
    apply_permutation<double, int>(out, in, perm, n);
}
void ref_apply_permutation_f32(float *out, float *in, int *perm, int n) {
    apply_permutation<float, int>(out, in, perm, n);
}

// ---------------------------------------------------------------- precision split (dp_sp)
// returns via out handles: MtxData<double,int>* and MtxData<float,int>*
void ref_partition_precisions_dpsp(void *mtx, double threshold, void **dp_out, void **sp_out) {
    Config cfg;
    cfg.value_type = "ap[dp_sp]";
    cfg.ap_threshold_1 = threshold;
    cfg.equilibrate = 0;
    auto *dp = new MtxData<double, int>;
    auto *sp = new MtxData<float, int>;
    std::vector<double> rmax, cmax;
    partition_precisions<double, int>(&cfg, (MtxData<double, int> *)mtx, dp, sp, &rmax, &cmax, 0);
    *dp_out = dp; *sp_out = sp;
}
void ref_mtx_f32_dims(void *h, long *out3) {
    auto *m = (MtxData<float, int> *)h;
    out3[0] = m->n_rows; out3[1] = m->n_cols; out3[2] = m->nnz;
}
void ref_mtx_f32_arrays(void *h, int *I, int *J, float *vals) {
    auto *m = (MtxData<float, int> *)h;
    copy_out(m->I, I, m->nnz); copy_out(m->J, J, m->nnz); copy_out(m->values, vals, m->nnz);
}

// ---------------------------------------------------------------- kernels (one precision)
#define ONE_PREC_KERNELS(SUF, VT)                                                              \
    void ref_spmv_omp_csr_##SUF(long n_rows, const int *rp, const int *ci, const VT *va,       \
                                VT *x, VT *y) {                                                \
        ST C = 1; int r = 0;                                                                   \
        spmv_omp_csr<VT, int>(true, &C, &n_rows, rp, nullptr, ci, va, x, y, nullptr, nullptr, &r); \
    }                                                                                          \
    void ref_spmv_omp_scs_##SUF(long C, long n_chunks, const int *cp, const int *cl,           \
                                const int *ci, const VT *va, VT *x, VT *y) {                   \
        int r = 0;                                                                             \
        spmv_omp_scs<VT, int>(true, &C, &n_chunks, cp, cl, ci, va, x, y, nullptr, nullptr, &r); \
    }                                                                                          \
    void ref_spmv_omp_scs_adv_##SUF(long C, long n_chunks, const int *cp, const int *cl,       \
                                    const int *ci, const VT *va, VT *x, VT *y) {               \
        int r = 0;                                                                             \
        spmv_omp_scs_adv<VT, int>(true, &C, &n_chunks, cp, cl, ci, va, x, y, nullptr, nullptr, &r); \
    }                                                                                          \
    void ref_block_spmv_omp_scs_general_##SUF(long C, long n_chunks, const int *cp,            \
                                              const int *cl, const int *ci, const VT *va,      \
                                              VT *X, VT *Y, int b, int vec_length) {           \
        int r = 0;                                                                             \
        block_spmv_omp_scs_general<VT, int>(true, &C, &n_chunks, cp, cl, ci, va, X, Y, &b,     \
                                            &vec_length, &r);                                  \
    }                                                                                          \
    void ref_block_spmv_omp_csr_##SUF(long n_rows, const int *rp, const int *ci, const VT *va, \
                                      VT *X, VT *Y, int b, int vec_length) {                   \
        ST C = 1; int r = 0;                                                                   \
        block_spmv_omp_csr<VT, int>(true, &C, &n_rows, rp, nullptr, ci, va, X, Y, &b,          \
                                    &vec_length, &r);                                          \
    }

ONE_PREC_KERNELS(f64, double)
ONE_PREC_KERNELS(f32, float)

// random_init (code/utilities.hpp:880-912): the -rand_x 1 vector, default-seeded std::mt19937 + uniform(matrix_min, matrix_max)
void ref_random_init_f64(double matrix_min, double matrix_max, long n, double *out) {
    Config cfg;
    cfg.matrix_min = matrix_min; cfg.matrix_max = matrix_max;
    random_init<double>(&cfg, out, out + n);
}
void ref_random_init_f32(double matrix_min, double matrix_max, long n, float *out) {
    Config cfg;
    cfg.matrix_min = matrix_min; cfg.matrix_max = matrix_max;
    random_init<float>(&cfg, out, out + n);
}

// 1 if this build uses the row-major block-vector layout (X[col*b+v]), 0 for column-major
int ref_block_layout_rowwise(void) {
#ifdef ROWWISE_BLOCK_VECTOR_LAYOUT
    return 1;
#else
    return 0;
#endif
}

// ---------------------------------------------------------------- kernels (adaptive dp+sp)
void ref_spmv_omp_scs_ap_adv(long C, long n_chunks, const int *dcp, const int *dcl,
                             const int *dci, const double *dva, double *dx, double *dy,
                             const int *scp, const int *scl, const int *sci, const float *sva,
                             float *sx, float *sy) {
    int r = 0;
    spmv_omp_scs_ap_adv<int>(true, &C, &n_chunks, dcp, dcl, dci, dva, dx, dy, &C, &n_chunks, scp,
                             scl, sci, sva, sx, sy, &r);
}
void ref_spmv_omp_scs_ap(long C, long n_chunks, const int *dcp, const int *dcl, const int *dci,
                         const double *dva, double *dx, double *dy, const int *scp,
                         const int *scl, const int *sci, const float *sva, float *sx, float *sy) {
    int r = 0;
    spmv_omp_scs_ap<int>(true, &C, &n_chunks, dcp, dcl, dci, dva, dx, dy, &C, &n_chunks, scp, scl,
                         sci, sva, sx, sy, &r);
}
void ref_spmv_omp_csr_apdpsp(long n_rows, const int *drp, const int *dci, const double *dva,
                             double *dx, double *dy, const int *srp, const int *sci,
                             const float *sva, float *sx, float *sy) {
    ST C = 1; int r = 0;
    spmv_omp_csr_apdpsp<int>(true, &C, &n_rows, drp, nullptr, dci, dva, dx, dy, &C, &n_rows, srp,
                             nullptr, sci, sva, sx, sy, &r);
}

#ifdef USE_MPI
// ---------------------------------------------------------------- fake-rank halo set-up
// No MPI call is reached by any of these (the functions used are pure integer code); the
// USE_MPI build exists only because code/mpi_funcs.hpp:15 guards them.
void ref_seg_work_sharing_arr(void *mtx, const char *seg_method, int comm_size, int *wsa) {
    Config cfg;
    cfg.seg_method = seg_method;
    seg_work_sharing_arr<double, int>(&cfg, (MtxData<double, int> *)mtx, wsa, comm_size, 0);
}

// local COO of fake rank `rank` with process-local row ids (seg_mtx_struct + localize_row_idx)
void *ref_seg_local_mtx(void *mtx, const int *wsa, int rank) {
    auto *tot = (MtxData<double, int> *)mtx;
    auto *loc = new MtxData<double, int>;
    std::vector<int> I, J;
    std::vector<double> V;
    seg_mtx_struct<double, int>(tot, &I, &J, &V, wsa, rank);
    loc->n_rows = (long)std::set<int>(I.begin(), I.end()).size();  // code/mpi_funcs.hpp:770
    loc->n_cols = tot->n_cols;
    loc->nnz = (long)V.size();
    loc->is_sorted = 1; loc->is_symmetric = 0;
    loc->I = I; loc->J = J; loc->values = V;
    localize_row_idx<double, int>(loc);
    return loc;
}

// runs collect_local_needed_heri on the f64 SCS handle (col_idxs rewritten in place).
// recv_cumsum: comm_size+1 ints.  recv_idxs_flat: concatenation by owner rank ascending
// (caller sizes it n_cols; returned count = total).  recv_idx_counts: comm_size ints.
int ref_collect_local_needed_heri(void *scs_f64, const int *wsa, int rank, int comm_size,
                                  int *recv_cumsum, int *recv_idxs_flat, int *recv_idx_counts) {
    auto *s = (ScsData<double, int> *)scs_f64;
    std::vector<std::vector<int>> recv_idxs(comm_size);
    std::vector<int> cumsum(comm_size + 1, 0);
    collect_local_needed_heri<double, int>("dp", &recv_idxs, &cumsum, s, wsa, rank, comm_size);
    int tot = 0;
    for (int p = 0; p < comm_size; ++p) {
        recv_idx_counts[p] = (int)recv_idxs[p].size();
        for (int v : recv_idxs[p]) recv_idxs_flat[tot++] = v;
    }
    for (int p = 0; p <= comm_size; ++p) recv_cumsum[p] = cumsum[p];
    return tot;
}
#endif

}  // extern "C"
