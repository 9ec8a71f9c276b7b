// Host side of uspmv_dist_check (csrc/uspmv_dist_api.hip): the y a distributed step MUST produce for the rows of one block,
// evaluated straight from the block's COO entries -- global column ids, the x vector a closed form of the global column --
// as one FMA chain per row in entry order.  That is the summation order of every kernel of the path (column order
// inside a row = file order = order of the FMA chain, code/kernels.hpp:242-246), so the comparison is bitwise and covers
// partition, halo discovery, exchange plan, exchange and kernels at any size.  It plays the part of the reference's MKL
// validation (code/write_results.hpp:442-556); it is a checker, never a result.
#include <algorithm>
#include <cmath>

#include "uspmv_internal.hpp"

namespace uspmv {

double check_x(int64_t g) { return 1.0 + 1e-3 * (double)(g % 1000); }

// loopback: a halo column j owned by block p is served by THIS block's row j - wsa[p] (equal block heights)
int64_t check_col(int64_t j, const int32_t *wsa, int P, int rank, bool loopback) {
    if (!loopback) return j;
    const int p = (int)(std::upper_bound(wsa, wsa + P + 1, (int32_t)j) - wsa) - 1;
    return p == rank || p < 0 || p >= P ? j : (int64_t)wsa[rank] + (j - wsa[p]);
}

template <typename VT>
static void rows(const uspmv_coo *m, const int32_t *wsa, int P, int rank, bool loopback, VT *y) {
    const int64_t n = m->n_rows, nnz = m->nnz;
    bool sorted = true;
    for (int64_t k = 1; k < nnz && sorted; ++k) sorted = m->I[(size_t)k] >= m->I[(size_t)k - 1];
    auto xval = [&](int32_t j) { return (VT)check_x(check_col(j, wsa, P, rank, loopback)); };
    if (!sorted) {
        for (int64_t i = 0; i < n; ++i) y[i] = VT(0);
        for (int64_t k = 0; k < nnz; ++k) {
            VT &a = y[m->I[(size_t)k]];
            a = std::fma((VT)m->values[(size_t)k], xval(m->J[(size_t)k]), a);
        }
        return;
    }
    std::vector<int64_t> rp((size_t)n + 1, 0);
    for (int64_t k = 0; k < nnz; ++k) ++rp[(size_t)m->I[(size_t)k] + 1];
    for (int64_t i = 0; i < n; ++i) rp[(size_t)i + 1] += rp[(size_t)i];
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        VT a = VT(0);
        for (int64_t k = rp[(size_t)i]; k < rp[(size_t)i + 1]; ++k) a = std::fma((VT)m->values[(size_t)k], xval(m->J[(size_t)k]), a);
        y[i] = a;
    }
}

int dist_reference_rows(const uspmv_coo *local, const int32_t *wsa, int P, int rank, bool loopback, int dtype, void *y_ref) {
    if (!local || !wsa || !y_ref) return fail(USPMV_ERR_INVALID, "uspmv_dist_check: NULL argument");
    for (int64_t k = 0; k < local->nnz; ++k)
        if (local->I[(size_t)k] < 0 || local->I[(size_t)k] >= local->n_rows) return fail(USPMV_ERR_INVALID, "uspmv_dist_check: row id outside the block");
    if (dtype == USPMV_F64) rows<double>(local, wsa, P, rank, loopback, (double *)y_ref);
    else rows<float>(local, wsa, P, rank, loopback, (float *)y_ref);
    return USPMV_OK;
}

}  // namespace uspmv

// The rows of the check on the host alone (no GPU): y_ref[i] = the entry-ordered FMA chain of local row i over
// x_global[j] = 1 + 1e-3 (j mod 1000) -- what uspmv_dist_check compares the device's y with; for steps that do not run on a
// uspmv_dist object (the torch.distributed twin of bench.py).
extern "C" int uspmv_dist_check_reference(const uspmv_coo_t *local, const int32_t *wsa, int rank, int P, int dtype, void *y_ref) {
    if (P < 1 || rank < 0 || rank >= P) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_check_reference: bad rank / P");
    if (dtype != USPMV_F64 && dtype != USPMV_F32) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_check_reference: unknown dtype %d", dtype);
    return uspmv::dist_reference_rows(local, wsa, P, rank, /*loopback=*/false, dtype, y_ref);
}
