// uspmv_launchers.hpp -- launchers with EXACTLY the signatures of the reference's kernel seam, on top of the C ABI (uspmv.h).
//
// The reference's SpmvKernel holds its kernel as a std::function of type OnePrecFuncPtr / MultiPrecFuncPtr
// (code/classes_structs.hpp:283-333, the __CUDACC__ form with n_thread_blocks) and fills it in its constructor with
// spmv_gpu_scs_adv_launcher / spmv_gpu_scs_launcher / spmv_gpu_csr_launcher (code/kernels.hpp:579-775) or
// spmv_gpu_ap_scs_adv_launcher (code/ap_kernels.hpp:821-953).  A HIP build of the harness assigns these instead:
//
//     one_prec_kernel_func_ptr   = uspmv_launchers::spmv_hip_scs_launcher<VT, IT>;      // scs, any C; also block vectors
//     one_prec_kernel_func_ptr   = uspmv_launchers::spmv_hip_csr_launcher<VT, IT>;      // crs
//     multi_prec_kernel_func_ptr = uspmv_launchers::spmv_hip_ap_scs_launcher<IT>;       // ap[dp_sp]
//
// Argument meaning is the reference's.  As in its GPU build, every array argument is a DEVICE pointer, and C / n_chunks may be
// device pointers too (code/utilities.hpp:3803-3811) or host pointers: they are read once per set of arrays.  The first call
// for a set of arrays wraps them (uspmv_dmat_wrap), builds the MI355X plan on the device from those very arrays
// (uspmv_dmat_optimize_device[_ap]: tile-local-column plan, internal C = 32 re-chunking of narrow chunks) and remembers the
// handle; later calls are one table look-up and one kernel launch.  The arrays must stay unchanged while cached
// (uspmv_launchers::release() before freeing or rewriting them).  Launches go to the default stream and return at once; the
// reference synchronises after the call (code/classes_structs.hpp:1032-1034: cudaDeviceSynchronize -> uspmv_stream_synchronize(NULL)).
// Error convention of the reference's launchers: message on stderr, exit(1).
#ifndef USPMV_LAUNCHERS_HPP
#define USPMV_LAUNCHERS_HPP

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "uspmv.h"

namespace uspmv_launchers {

using ST = long;   // code/classes_structs.hpp:31

namespace detail {

struct Entry {
    const void *cp, *cl, *ci, *va;        // key: the device arrays
    const void *cp2, *cl2, *ci2, *va2;    // second struct of an ap pair (nullptr otherwise)
    uspmv_dmat_t *A, *A2;
    ST C;
    int b_planned;                        // block width the handle's block plan was built for (0: none yet)
};
inline std::vector<Entry> &table() { static std::vector<Entry> t; return t; }
inline std::mutex &lock() { static std::mutex m; return m; }

[[noreturn]] inline void die(const char *who) {
    fprintf(stderr, "ERROR: %s: %s\n", who, uspmv_last_error());
    exit(1);
}
inline void ck(int rc, const char *who) { if (rc != USPMV_OK) die(who); }

inline uspmv_dmat_t *wrap(const ST *C, const ST *n_chunks, const void *cp, const void *cl, const void *ci, const void *va, int dtype, ST *C_out) {
    int64_t c = 0, nc = 0;
    int32_t n_el = 0;
    ck(uspmv_peek_i64(C, &c), "uspmv_launchers (C)");
    ck(uspmv_peek_i64(n_chunks, &nc), "uspmv_launchers (n_chunks)");
    ck(uspmv_peek_i32((const int32_t *)cp + nc, &n_el), "uspmv_launchers (chunk_ptrs)");
    uspmv_dmat_t *A = nullptr;
    ck(uspmv_dmat_wrap(c, nc, n_el, dtype, (const int32_t *)cp, (const int32_t *)cl, (const int32_t *)ci, va, &A), "uspmv_dmat_wrap");
    *C_out = (ST)c;
    return A;
}

inline Entry one(const ST *C, const ST *n_chunks, const void *cp, const void *cl, const void *ci, const void *va, int dtype, bool crs) {
    std::lock_guard<std::mutex> g(lock());
    for (const Entry &e : table())
        if (e.cp == cp && e.cl == cl && e.ci == ci && e.va == va && !e.cp2) return e;
    Entry e{cp, cl, ci, va, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
    e.A = wrap(C, n_chunks, cp, cl, ci, va, dtype, &e.C);
    if (crs) ck(uspmv_dmat_set_crs(e.A, 1), "uspmv_dmat_set_crs");
    ck(uspmv_dmat_optimize_device(e.A, 0, nullptr, nullptr), "uspmv_dmat_optimize_device");
    table().push_back(e);
    return table().back();
}

// the LDS block plan for block vectors of width b, built once per (array set, b) from the device arrays
inline void plan_block(const Entry &e, int b) {
    std::lock_guard<std::mutex> g(lock());
    for (Entry &t : table())
        if (t.A == e.A) {
            if (t.b_planned != b) { ck(uspmv_dmat_optimize_block_device(t.A, b, nullptr, nullptr), "uspmv_dmat_optimize_block_device"); t.b_planned = b; }
            return;
        }
}

inline Entry pair(const ST *dC, const ST *dn, const void *dcp, const void *dcl, const void *dci, const void *dva,
                         const ST *sC, const ST *sn, const void *scp, const void *scl, const void *sci, const void *sva) {
    std::lock_guard<std::mutex> g(lock());
    for (const Entry &e : table())
        if (e.cp == dcp && e.ci == dci && e.va == dva && e.cp2 == scp && e.ci2 == sci && e.va2 == sva) return e;
    Entry e{dcp, dcl, dci, dva, scp, scl, sci, sva, nullptr, nullptr, 0, 0};
    ST c2 = 0;
    e.A = wrap(dC, dn, dcp, dcl, dci, dva, USPMV_F64, &e.C);
    e.A2 = wrap(sC, sn, scp, scl, sci, sva, USPMV_F32, &c2);
    ck(uspmv_dmat_optimize_device_ap(e.A, e.A2, 0, nullptr, nullptr), "uspmv_dmat_optimize_device_ap");
    table().push_back(e);
    return table().back();
}

}  // namespace detail

// which MI355X plan the cached handle of a set of arrays runs on (uspmv_dmat_plan_info: 0 none, 1 tile-local-column, 2 column-window
// sweep); -1: no handle cached for these arrays.  For logging / tests: the launchers pick the plan by themselves.
inline int plan_kind(const void *chunk_ptrs) {
    std::lock_guard<std::mutex> g(detail::lock());
    for (const detail::Entry &e : detail::table())
        if (e.cp == chunk_ptrs) {
            int k = 0;
            detail::ck(uspmv_dmat_plan_info(e.A, &k, nullptr, nullptr), "uspmv_dmat_plan_info");
            return k;
        }
    return -1;
}

// forget (and free) every cached handle -- before the arrays behind them are freed or rewritten
inline void release() {
    std::lock_guard<std::mutex> g(detail::lock());
    for (detail::Entry &e : detail::table()) { uspmv_dmat_free(e.A); uspmv_dmat_free(e.A2); }
    detail::table().clear();
}

// OnePrecFuncPtr (code/classes_structs.hpp:283-299); replaces spmv_gpu_scs_adv_launcher / spmv_gpu_scs_launcher and the
// block_spmv_gpu_scs_*_launcher stubs (code/kernels.hpp:757-844): block_vec_size > 1 runs the SpMMV kernels with
// vec_length as the leading dimension (layout = the build's *_BLOCK_VECTOR_LAYOUT macro, column-wise by default).
template <typename VT, typename IT>
void spmv_hip_scs_launcher(bool /*warmup_flag*/, const ST *C, const ST *n_chunks, const IT *chunk_ptrs, const IT *chunk_lengths,
                           const IT *col_idxs, const VT *values, VT *x, VT *y, int *block_vec_size, int *vec_length,
                           const ST /*n_thread_blocks*/, const int * /*my_rank*/) {
    static_assert(sizeof(IT) == 4, "IT = int (code/main.cpp:1711-1718)");
    static_assert(sizeof(VT) == 8 || sizeof(VT) == 4, "VT = double | float");
    const detail::Entry e = detail::one(C, n_chunks, chunk_ptrs, chunk_lengths, col_idxs, values, sizeof(VT) == 8 ? USPMV_F64 : USPMV_F32, false);
    const int b = block_vec_size ? *block_vec_size : 1;
    if (b > 1) {
#ifdef ROWWISE_BLOCK_VECTOR_LAYOUT
        const int layout = USPMV_ROWWISE;
#else
        const int layout = USPMV_COLWISE;
#endif
        if (e.b_planned != b) detail::plan_block(e, b);
        detail::ck(uspmv_spmmv(e.A, x, y, b, vec_length ? *vec_length : 0, layout, nullptr), "uspmv_spmmv");
    } else {
        detail::ck(uspmv_spmv(e.A, x, y, nullptr), "uspmv_spmv");
    }
}

// OnePrecFuncPtr for `crs` (replaces spmv_gpu_csr_launcher / block_spmv_gpu_csr_launcher, code/kernels.hpp:579-640, :777-800):
// C = 1, chunk_ptrs = row pointers.
template <typename VT, typename IT>
void spmv_hip_csr_launcher(bool /*warmup_flag*/, const ST *C, const ST *n_chunks, const IT *chunk_ptrs, const IT *chunk_lengths,
                           const IT *col_idxs, const VT *values, VT *x, VT *y, int *block_vec_size, int *vec_length,
                           const ST /*n_thread_blocks*/, const int * /*my_rank*/) {
    static_assert(sizeof(IT) == 4, "IT = int");
    const detail::Entry e = detail::one(C, n_chunks, chunk_ptrs, chunk_lengths, col_idxs, values, sizeof(VT) == 8 ? USPMV_F64 : USPMV_F32, true);
    const int b = block_vec_size ? *block_vec_size : 1;
    if (b > 1) {
#ifdef ROWWISE_BLOCK_VECTOR_LAYOUT
        const int layout = USPMV_ROWWISE;
#else
        const int layout = USPMV_COLWISE;
#endif
        detail::ck(uspmv_spmmv(e.A, x, y, b, vec_length ? *vec_length : 0, layout, nullptr), "uspmv_spmmv");
    } else {
        detail::ck(uspmv_spmv(e.A, x, y, nullptr), "uspmv_spmv");
    }
}

// MultiPrecFuncPtr without HAVE_HALF_MATH (code/classes_structs.hpp:301-333); replaces spmv_gpu_ap_scs_adv_launcher /
// spmv_gpu_ap_scs_launcher (code/ap_kernels.hpp:721-953).  C in {2,4,...,128}: numerics of scs_ap_impl_cpu (both parts times the
// double x, code/ap_kernels.hpp:59-75; sp_x / sp_y unused); any other C: spmv_omp_scs_ap (sp part times the float x, :619-623).
template <typename IT>
void spmv_hip_ap_scs_launcher(bool /*warmup_flag*/, const ST *dp_C, const ST *dp_n_chunks, const IT *dp_chunk_ptrs, const IT *dp_chunk_lengths,
                              const IT *dp_col_idxs, const double *dp_values, double *dp_x, double *dp_y, const ST *sp_C,
                              const ST *sp_n_chunks, const IT *sp_chunk_ptrs, const IT *sp_chunk_lengths, const IT *sp_col_idxs,
                              const float *sp_values, float *sp_x, float * /*sp_y*/, const ST /*n_thread_blocks*/, const int * /*my_rank*/) {
    static_assert(sizeof(IT) == 4, "IT = int");
    const detail::Entry e = detail::pair(dp_C, dp_n_chunks, dp_chunk_ptrs, dp_chunk_lengths, dp_col_idxs, dp_values, sp_C, sp_n_chunks,
                                          sp_chunk_ptrs, sp_chunk_lengths, sp_col_idxs, sp_values);
    const ST C = e.C;
    const bool adv = C == 2 || C == 4 || C == 8 || C == 16 || C == 32 || C == 64 || C == 128;   // code/classes_structs.hpp:545-551, :630-636
    if (adv) detail::ck(uspmv_spmv_ap(e.A, e.A2, dp_x, dp_y, nullptr), "uspmv_spmv_ap");
    else detail::ck(uspmv_spmv_ap_generic(e.A, e.A2, dp_x, sp_x, dp_y, nullptr), "uspmv_spmv_ap_generic");
}

}  // namespace uspmv_launchers
#endif  // USPMV_LAUNCHERS_HPP
