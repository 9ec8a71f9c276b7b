// uspmv_interface.hpp -- C++ convenience layer with the names of the reference's single-header
// library (code/interface.hpp, API_doc.md:7-24), implemented on top of the C ABI in uspmv.h.
//
//   MtxData<VT,IT>, ScsData<VT,IT>            code/interface.hpp:16-80
//   convert_to_scs<MT,VT,IT>(...)             code/interface.hpp:401-656
//   permute_scs_cols<VT,IT>(...)              code/interface.hpp:659-688
//   apply_permutation<VT,IT>(...)             code/interface.hpp:379-399
//   partition_precisions(...)  (dp_sp)        code/interface.hpp:690-978
//   uspmv_scs_gpu / uspmv_csr_gpu             code/interface.hpp:1741-1793  (host-callable launchers
//                                             here; the reference declares them __global__)
//   execute_uspmv(...)                        code/interface.hpp:1871-2187  (device pointers)
//
// A host application keeps its own MPI/RCCL; these calls are per-rank local (API_doc.md:5).
// Errors throw std::runtime_error carrying uspmv_last_error() instead of calling exit().
#ifndef USPMV_INTERFACE_HPP
#define USPMV_INTERFACE_HPP

#include <cstring>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "uspmv.h"

#ifndef USPMV_ST_DEFINED
#define USPMV_ST_DEFINED
using ST = long;  // code/classes_structs.hpp:31
#endif

namespace uspmv_detail {
inline void check(int rc, const char *what) {
    if (rc != USPMV_OK) throw std::runtime_error(std::string(what) + ": " + uspmv_last_error());
}
template <typename VT> constexpr int dtype_of() {
    static_assert(std::is_same<VT, double>::value || std::is_same<VT, float>::value, "VT must be double or float");
    return std::is_same<VT, double>::value ? USPMV_F64 : USPMV_F32;
}
}  // namespace uspmv_detail

template <typename VT, typename IT>
struct MtxData {
    ST n_rows{}, n_cols{}, nnz{};
    bool is_sorted{}, is_symmetric{};
    std::vector<IT> I, J;
    std::vector<VT> values;
};

template <typename VT, typename IT>
struct ScsData {
    ST C{}, sigma{}, n_rows{}, n_cols{}, n_rows_padded{}, n_chunks{}, n_elements{}, nnz{};
    std::vector<IT> chunk_ptrs, chunk_lengths, col_idxs;
    std::vector<VT> values;
    std::vector<IT> old_to_new_idx;
    std::vector<IT> new_to_old_idx;  // (raw IT* in the reference)
};

// read_mtx of the harness (code/utilities.hpp:2148-2309), offered here for convenience
inline void read_mtx(const std::string &path, MtxData<double, int> *m) {
    uspmv_coo_t *h = nullptr;
    uspmv_detail::check(uspmv_read_mtx(path.c_str(), &h), "read_mtx");
    int64_t nr, nc, nz;
    const int32_t *I, *J;
    const double *V;
    uspmv_coo_dims(h, &nr, &nc, &nz);
    uspmv_coo_arrays(h, &I, &J, &V);
    m->n_rows = nr; m->n_cols = nc; m->nnz = nz; m->is_sorted = true; m->is_symmetric = false;
    m->I.assign(I, I + nz); m->J.assign(J, J + nz); m->values.assign(V, V + nz);
    uspmv_coo_free(h);
}

template <typename MT, typename VT, typename IT>
void convert_to_scs(MtxData<MT, IT> *mtx, ST C, ST sigma, ScsData<VT, IT> *scs, int *fixed_permutation = nullptr,
                    int * /*work_sharing_arr*/ = nullptr, int /*my_rank*/ = 0) {
    static_assert(std::is_same<IT, int>::value, "IT must be int (as in every instantiation of the reference)");
    std::vector<double> v(mtx->values.begin(), mtx->values.end());
    uspmv_coo_t *coo = nullptr;
    uspmv_detail::check(uspmv_coo_create(mtx->n_rows, mtx->n_cols, mtx->nnz, mtx->I.data(), mtx->J.data(), v.data(), &coo),
                        "convert_to_scs");
    uspmv_scs_t *s = nullptr;
    int rc = uspmv_convert_to_scs(coo, C, sigma, uspmv_detail::dtype_of<VT>(), fixed_permutation, &s);
    uspmv_coo_free(coo);
    uspmv_detail::check(rc, "convert_to_scs");
    int64_t m[8];
    uspmv_scs_meta(s, m);
    scs->C = m[0]; scs->sigma = m[1]; scs->n_rows = m[2]; scs->n_cols = m[3]; scs->n_rows_padded = m[4];
    scs->n_chunks = m[5]; scs->n_elements = m[6]; scs->nnz = m[7];
    const int32_t *cp, *cl, *ci, *o2n, *n2o;
    const void *va;
    uspmv_scs_arrays(s, &cp, &cl, &ci, &va, &o2n, &n2o);
    scs->chunk_ptrs.assign(cp, cp + m[5] + 1);
    scs->chunk_lengths.assign(cl, cl + m[5]);
    scs->col_idxs.assign(ci, ci + m[6]);
    scs->values.assign((const VT *)va, (const VT *)va + m[6]);
    scs->old_to_new_idx.assign(o2n, o2n + m[2]);
    scs->new_to_old_idx.assign(n2o, n2o + m[2]);
    uspmv_scs_free(s);
}

template <typename VT, typename IT>
void permute_scs_cols(ScsData<VT, IT> *scs, IT *perm) {  // code/utilities.hpp:1802-1831
    for (ST i = 0; i < scs->n_elements; ++i)
        if (scs->col_idxs[i] < scs->n_rows) scs->col_idxs[i] = perm[scs->col_idxs[i]];
}

template <typename VT, typename IT>
void apply_permutation(VT *permuted_vec, VT *vec_to_permute, IT *perm, int num_elems_to_permute) {
    uspmv_detail::check(uspmv_apply_permutation(permuted_vec, vec_to_permute, perm, num_elems_to_permute,
                                                uspmv_detail::dtype_of<VT>()), "apply_permutation");
}

// ap[dp_sp] split, non-equilibrated (code/utilities.hpp:2899-2911)
inline void partition_precisions(double ap_threshold_1, MtxData<double, int> *local_mtx, MtxData<double, int> *dp_local_mtx,
                                 MtxData<float, int> *sp_local_mtx) {
    uspmv_coo_t *coo = nullptr, *dp = nullptr, *sp = nullptr;
    uspmv_detail::check(uspmv_coo_create(local_mtx->n_rows, local_mtx->n_cols, local_mtx->nnz, local_mtx->I.data(),
                                         local_mtx->J.data(), local_mtx->values.data(), &coo), "partition_precisions");
    int rc = uspmv_partition_precisions(coo, ap_threshold_1, &dp, &sp);
    uspmv_coo_free(coo);
    uspmv_detail::check(rc, "partition_precisions");
    auto fill = [](uspmv_coo_t *h, auto *out) {
        int64_t nr, nc, nz; const int32_t *I, *J; const double *V;
        uspmv_coo_dims(h, &nr, &nc, &nz); uspmv_coo_arrays(h, &I, &J, &V);
        out->n_rows = nr; out->n_cols = nc; out->nnz = nz; out->is_sorted = true; out->is_symmetric = false;
        out->I.assign(I, I + nz); out->J.assign(J, J + nz); out->values.assign(V, V + nz);
        uspmv_coo_free(h);
    };
    fill(dp, dp_local_mtx);
    fill(sp, sp_local_mtx);
}

// ---- device kernels: all pointers are DEVICE pointers (as in the reference's CUDA build, where even
// C and n_chunks live on the device, code/utilities.hpp:3803-3811; here the scalars are by value)
template <typename VT, typename IT>
void uspmv_scs_gpu(const ST C, const ST n_chunks, const IT *chunk_ptrs, const IT *chunk_lengths, const IT *col_idxs,
                   const VT *values, const VT *x, VT *y, void *stream = nullptr) {
    int rc = std::is_same<VT, double>::value
                 ? uspmv_scs_gpu_f64(C, n_chunks, chunk_ptrs, chunk_lengths, col_idxs, (const double *)values, (const double *)x, (double *)y, stream)
                 : uspmv_scs_gpu_f32(C, n_chunks, chunk_ptrs, chunk_lengths, col_idxs, (const float *)values, (const float *)x, (float *)y, stream);
    uspmv_detail::check(rc, "uspmv_scs_gpu");
}
template <typename VT, typename IT>
void uspmv_scs_c_gpu(const ST C, const ST n_chunks, const IT *chunk_ptrs, const IT *chunk_lengths, const IT *col_idxs,
                     const VT *values, const VT *x, VT *y, void *stream = nullptr) {
    uspmv_scs_gpu<VT, IT>(C, n_chunks, chunk_ptrs, chunk_lengths, col_idxs, values, x, y, stream);
}
template <typename VT, typename IT>
void uspmv_csr_gpu(const ST num_rows, const IT *row_ptrs, const IT *col_idxs, const VT *values, const VT *x, VT *y,
                   void *stream = nullptr) {
    int rc = std::is_same<VT, double>::value
                 ? uspmv_csr_gpu_f64(num_rows, row_ptrs, col_idxs, (const double *)values, (const double *)x, (double *)y, stream)
                 : uspmv_csr_gpu_f32(num_rows, row_ptrs, col_idxs, (const float *)values, (const float *)x, (float *)y, stream);
    uspmv_detail::check(rc, "uspmv_csr_gpu");
}

// execute_uspmv (code/interface.hpp:1871-1910, :2037-2046): SELL kernel when C > 1 or sigma > 1, else
// CRS.  CHUNK_SIZE / SIGMA are host-defined macros in the reference; arguments here.
template <typename VT, typename IT>
void execute_uspmv(const ST chunk_size, const ST sigma, const ST *C, const ST *n_chunks, const IT *chunk_ptrs,
                   const IT *chunk_lengths, const IT *col_idxs, const VT *values, VT *x, VT *y, void *stream = nullptr) {
    if (chunk_size > 1 || sigma > 1) uspmv_scs_gpu<VT, IT>(*C, *n_chunks, chunk_ptrs, chunk_lengths, col_idxs, values, x, y, stream);
    else uspmv_csr_gpu<VT, IT>(*n_chunks, chunk_ptrs, col_idxs, values, x, y, stream);
}

// ---- beyond interface.hpp: a device-resident matrix with the MI355X plan, for callers that keep the matrix across
// many SpMVs (what the reference's SpmvKernel object does with its cudaMalloc'ed arrays).  RAII over uspmv_dmat_t.
class DeviceScs {
public:
    DeviceScs() = default;
    DeviceScs(const DeviceScs &) = delete;
    DeviceScs &operator=(const DeviceScs &) = delete;
    DeviceScs(DeviceScs &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    ~DeviceScs() { if (h_) uspmv_dmat_free(h_); }
    // arrays that already live in HBM (owned by the caller); optimise = build the tile-local-column plan on the device
    template <typename VT, typename IT>
    static DeviceScs wrap(ST C, ST n_chunks, ST n_elements, const IT *chunk_ptrs, const IT *chunk_lengths, const IT *col_idxs,
                          const VT *values, bool optimise = true) {
        DeviceScs d;
        uspmv_detail::check(uspmv_dmat_wrap(C, n_chunks, n_elements, uspmv_detail::dtype_of<VT>(), chunk_ptrs, chunk_lengths, col_idxs,
                                            values, &d.h_), "uspmv_dmat_wrap");
        if (optimise) uspmv_detail::check(uspmv_dmat_optimize_device(d.h_, 0, nullptr, nullptr), "uspmv_dmat_optimize_device");
        return d;
    }
    // COO -> SELL-C-sigma with the O(nnz) part on the GPU (convert_to_scs + permute_scs_cols + staging in one call);
    // *perm receives old_to_new_idx for apply_permutation on x and y
    static DeviceScs from_mtx(const MtxData<double, int> &m, ST C, ST sigma, bool single_precision, std::vector<int> *perm,
                              std::vector<int> *inv_perm, bool optimise = true) {
        uspmv_coo_t *coo = nullptr;
        uspmv_detail::check(uspmv_coo_create(m.n_rows, m.n_cols, m.nnz, m.I.data(), m.J.data(), m.values.data(), &coo), "uspmv_coo_create");
        uspmv_scs_t *layout = nullptr;
        DeviceScs d;
        int rc = uspmv_convert_to_scs_device(coo, C, sigma, single_precision ? USPMV_F32 : USPMV_F64, nullptr, 1, &layout, &d.h_);
        uspmv_coo_free(coo);
        uspmv_detail::check(rc, "uspmv_convert_to_scs_device");
        const int32_t *o2n = nullptr, *n2o = nullptr;
        uspmv_scs_arrays(layout, nullptr, nullptr, nullptr, nullptr, &o2n, &n2o);
        if (perm) perm->assign(o2n, o2n + m.n_rows);
        if (inv_perm) inv_perm->assign(n2o, n2o + m.n_rows);
        int64_t meta[8];
        uspmv_scs_meta(layout, meta);
        d.n_rows_padded_ = meta[4];
        uspmv_scs_free(layout);
        if (optimise) uspmv_detail::check(uspmv_dmat_optimize_device(d.h_, 0, nullptr, nullptr), "uspmv_dmat_optimize_device");
        return d;
    }
    // the same from COO arrays that already live in HBM (d_I, d_J: int; d_V: double; entries sorted by row): nothing but O(n_rows)
    // integers leaves the device.  stable_ties = false: the reference's std::sort tie order (every array as convert_to_scs gives it);
    // true: rows of equal length keep their original order (ordering on the device; y in original row order is the same either way).
    // d_perm / d_inv_perm (device, n_rows ints each, may be null) receive old_to_new_idx / new_to_old_idx.
    static DeviceScs from_device_coo(const int *d_I, const int *d_J, const double *d_V, ST n_rows, ST n_cols, ST nnz, ST C, ST sigma,
                                     bool single_precision, int *d_perm, int *d_inv_perm, bool stable_ties = false, bool optimise = true,
                                     void *stream = nullptr) {
        DeviceScs d;
        uspmv_detail::check(uspmv_convert_to_scs_device_from_arrays(d_I, d_J, d_V, n_rows, n_cols, nnz, C, sigma, single_precision ? USPMV_F32 : USPMV_F64,
                                                                    nullptr, 1, stable_ties ? USPMV_SORT_DEVICE_STABLE : USPMV_SORT_HOST, stream, nullptr,
                                                                    d_perm, d_inv_perm, &d.h_), "uspmv_convert_to_scs_device_from_arrays");
        d.n_rows_padded_ = ((n_rows + C - 1) / C) * C;
        if (optimise) uspmv_detail::check(uspmv_dmat_optimize_device(d.h_, 0, nullptr, nullptr), "uspmv_dmat_optimize_device");
        return d;
    }
    void spmv(const void *d_x, void *d_y, void *stream = nullptr) const { uspmv_detail::check(uspmv_spmv(h_, d_x, d_y, stream), "uspmv_spmv"); }
    // a column-major X that does not change between calls (a bench loop): re-laid out once, until x_released() or another X
    void x_prepared(const void *d_X, int b, ST ld, void *stream = nullptr) const { uspmv_detail::check(uspmv_spmmv_x_prepared(h_, d_X, b, ld, stream), "uspmv_spmmv_x_prepared"); }
    void x_released() const { uspmv_detail::check(uspmv_spmmv_x_release(h_), "uspmv_spmmv_x_release"); }
    void spmmv(const void *d_X, void *d_Y, int b, ST ld, bool rowwise, void *stream = nullptr) const {
        uspmv_detail::check(uspmv_spmmv(h_, d_X, d_Y, b, ld, rowwise ? USPMV_ROWWISE : USPMV_COLWISE, stream), "uspmv_spmmv");
    }
    ST n_rows_padded() const { return n_rows_padded_; }
    uspmv_dmat_t *handle() const { return h_; }

private:
    uspmv_dmat_t *h_ = nullptr;
    ST n_rows_padded_ = 0;
};

#endif  // USPMV_INTERFACE_HPP
