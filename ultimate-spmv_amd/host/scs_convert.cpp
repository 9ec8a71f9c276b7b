// COO -> SELL-C-sigma conversion and the small host-side helpers around it.
//
// Contract = reference convert_to_scs (code/utilities.hpp:1842-2104; library twin
// code/interface.hpp:401-656), permute_scs_cols (code/utilities.hpp:1802-1831),
// apply_permutation (:1768-1782), partition_precisions ap[dp_sp] (:2899-2911).
// All integer outputs (chunk_ptrs, chunk_lengths, col_idxs, old_to_new_idx, new_to_old_idx)
// are bit-identical to the reference's for the same input, including
//   * the tie order of the sigma-window sort: the reference calls the UNSTABLE std::sort on
//     std::pair<long,long>{row, length} with the comparator a.second > b.second
//     (:1892-1941); we call the same libstdc++ algorithm on the same element type and
//     comparator, window by window (windows are independent, so they are sorted in parallel);
//   * the fixed_permutation quirks (:1911-1928): row lengths are moved to their new slots but
//     `.first` stays the identity, so old_to_new_idx/new_to_old_idx of such a struct come out
//     as the identity while the entries themselves are placed with fixed_permutation (:2017-2019);
//     padded slots i >= n_rows are re-zeroed AFTER the move (loop order of :1913-1923).
// Unlike the reference the conversion is O(nnz) parallel work (OpenMP over windows / rows) and
// reports 32-bit overflow instead of continuing.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <utility>

#include "uspmv_internal.hpp"

// Everything of convert_to_scs except the O(nnz) scatter: row lengths, sigma-window sort (or the
// fixed permutation), chunk lengths / pointers, both permutations.  `s` comes back with empty
// col_idxs / values ("layout-only").  row_start (n_rows + 1 offsets into the COO arrays) is filled
// when the COO entries are sorted by row, left empty otherwise.
int uspmv_scs_layout(const uspmv_coo_t *m, int64_t C, int64_t sigma, int dtype, const int32_t *fixed_permutation,
                     uspmv_scs *s, std::vector<int64_t> *row_start, const char *who) {
    if (C < 1 || sigma < 1) return uspmv::fail(USPMV_ERR_INVALID, "%s: C and sigma must be >= 1", who);
    if (dtype != USPMV_F64 && dtype != USPMV_F32) return uspmv::fail(USPMV_ERR_INVALID, "%s: unknown dtype %d", who, dtype);
    if (m->n_rows < 1) return uspmv::fail(USPMV_ERR_INVALID, "%s: matrix has no rows", who);

    const int64_t n_rows = m->n_rows, nnz = m->nnz;
    const int64_t n_chunks = (n_rows + C - 1) / C;
    const int64_t n_pad = n_chunks * C;
    if (n_pad > INT32_MAX) return uspmv::fail(USPMV_ERR_OVERFLOW, "%s: padded rows exceed int32", who);

    using row_len = std::pair<long, long>;  // {original row, population count}
    std::vector<row_len> rl((size_t)(n_pad + sigma));
    for (int64_t i = 0; i < n_pad; ++i) rl[(size_t)i].first = i;
    // row populations and the sortedness test in one parallel sweep: every thread walks a contiguous piece of
    // the entry list and adds one count per run of equal row indices (one atomic per run, not per entry)
    bool sorted = true;
    const int32_t *I = m->I.data();
#pragma omp parallel
    {
        bool my_sorted = true;
#pragma omp for schedule(static) nowait
        for (int64_t blk = 0; blk < (nnz + 65535) / 65536; ++blk) {
            const int64_t k0 = blk * 65536, k1 = std::min(k0 + 65536, nnz);
            if (k0 > 0 && I[k0 - 1] > I[k0]) my_sorted = false;
            int64_t run_start = k0;
            for (int64_t k = k0 + 1; k <= k1; ++k) {
                if (k == k1 || I[k] != I[k - 1]) {
                    long &cnt = rl[(size_t)I[k - 1]].second;
                    const long add = (long)(k - run_start);
#pragma omp atomic
                    cnt += add;
                    run_start = k;
                }
                if (k < k1 && I[k - 1] > I[k]) my_sorted = false;
            }
        }
        if (!my_sorted) {
#pragma omp atomic write
            sorted = false;
        }
    }
    row_start->clear();
    if (sorted) {
        row_start->assign((size_t)n_rows + 1, 0);
        for (int64_t r = 0; r < n_rows; ++r) (*row_start)[(size_t)r + 1] = (*row_start)[(size_t)r] + rl[(size_t)r].second;
    }

    if (fixed_permutation) {
        for (int64_t i = 0; i < n_rows; ++i)
            if (fixed_permutation[i] < 0 || fixed_permutation[i] >= n_pad)
                return uspmv::fail(USPMV_ERR_INVALID, "%s: fixed_permutation[%lld]=%d out of range", who, (long long)i,
                                   fixed_permutation[i]);
        std::vector<row_len> tmp((size_t)n_pad);
        for (int64_t i = 0; i < n_pad; ++i) {
            tmp[(size_t)i].first = rl[(size_t)i].first;
            if (i < n_rows) tmp[(size_t)fixed_permutation[i]].second = rl[(size_t)i].second;
            else tmp[(size_t)i].second = rl[(size_t)i].second;
        }
        std::copy(tmp.begin(), tmp.end(), rl.begin());
    } else {
        const int64_t n_win = (n_pad + sigma - 1) / sigma;
#pragma omp parallel for schedule(dynamic, 64)
        for (int64_t w = 0; w < n_win; ++w) {
            int64_t b = w * sigma, e = std::min(b + sigma, n_pad);
            std::sort(rl.begin() + b, rl.begin() + e,
                      [](const row_len &a, const row_len &b2) { return a.second > b2.second; });
        }
    }

    s->C = C; s->sigma = sigma; s->n_rows = n_rows; s->n_cols = m->n_cols; s->nnz = nnz;
    s->n_chunks = n_chunks; s->n_rows_padded = n_pad; s->dtype = dtype;
    s->chunk_lengths.assign((size_t)n_chunks, 0);
    s->chunk_ptrs.assign((size_t)n_chunks + 1, 0);

    int64_t cur = 0;
    for (int64_t c = 0; c < n_chunks; ++c) {
        long mx = 0;
        for (int64_t i = 0; i < C; ++i) mx = std::max(mx, rl[(size_t)(c * C + i)].second);
        s->chunk_lengths[(size_t)c] = (int32_t)mx;
        s->chunk_ptrs[(size_t)c] = (int32_t)cur;
        cur += mx * C;
        if (cur > INT32_MAX) return uspmv::fail(USPMV_ERR_OVERFLOW, "%s: chunk_ptrs exceed the 32-bit index type", who);
    }
    s->chunk_ptrs[(size_t)n_chunks] = (int32_t)cur;
    s->n_elements = cur;

    s->old_to_new_idx.assign((size_t)n_rows, 0);
    for (int64_t i = 0; i < n_pad; ++i) {
        long old_row = rl[(size_t)i].first;
        if (old_row < n_rows) s->old_to_new_idx[(size_t)old_row] = (int32_t)i;
    }
    s->new_to_old_idx.assign((size_t)n_rows, 0);
    for (int64_t i = 0; i < n_rows; ++i) {
        int32_t p = s->old_to_new_idx[(size_t)i];
        if (p < n_rows) s->new_to_old_idx[(size_t)p] = (int32_t)i;  // (reference writes out of bounds otherwise)
    }
    if (fixed_permutation && sorted) {
        // a non-empty row mapped onto a slot of a shorter chunk would overrun it (the reference does, code/utilities.hpp:1919-1922)
        for (int64_t r = 0; r < n_rows; ++r)
            if ((*row_start)[(size_t)r + 1] - (*row_start)[(size_t)r] > s->chunk_lengths[(size_t)(fixed_permutation[r] / C)])
                return uspmv::fail(USPMV_ERR_INVALID,
                                   "%s: fixed_permutation maps a non-empty row onto a padded slot "
                                   "(the reference overruns its chunk here, code/utilities.hpp:1919-1922)", who);
    }
    return USPMV_OK;
}

extern "C" {

int uspmv_convert_to_scs(const uspmv_coo_t *m, int64_t C, int64_t sigma, int dtype,
                         const int32_t *fixed_permutation, uspmv_scs_t **out) {
    if (!m || !out) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_convert_to_scs: NULL argument");
    auto *s = new uspmv_scs;
    std::vector<int64_t> row_start;
    if (int rc = uspmv_scs_layout(m, C, sigma, dtype, fixed_permutation, s, &row_start, "uspmv_convert_to_scs")) { delete s; return rc; }
    const int64_t n_rows = m->n_rows, nnz = m->nnz, n_pad = s->n_rows_padded, cur = s->n_elements;

    // ---- fill, preserving the COO order inside every row
    s->col_idxs.assign((size_t)cur, 0);  // padding: column 0 (code/utilities.hpp:1991-2002)
    if (dtype == USPMV_F64) s->values_f64.assign((size_t)cur, 0.0);
    else s->values_f32.assign((size_t)cur, 0.0f);

    const int32_t *row_map = fixed_permutation ? fixed_permutation : s->old_to_new_idx.data();
    int bad = 0;
    auto place = [&](int64_t k, int64_t slot) {
        int64_t row = row_map[m->I[(size_t)k]];
        int64_t c = row / C;
        if (slot >= s->chunk_lengths[(size_t)c]) { bad = 1; return; }
        int64_t idx = (int64_t)s->chunk_ptrs[(size_t)c] + slot * C + row % C;
        s->col_idxs[(size_t)idx] = m->J[(size_t)k];
        if (dtype == USPMV_F64) s->values_f64[(size_t)idx] = m->values[(size_t)k];
        else s->values_f32[(size_t)idx] = (float)m->values[(size_t)k];
    };
    if (!row_start.empty()) {
#pragma omp parallel for schedule(static)
        for (int64_t r = 0; r < n_rows; ++r)
            for (int64_t k = row_start[(size_t)r]; k < row_start[(size_t)r + 1]; ++k) place(k, k - row_start[(size_t)r]);
    } else {
        std::vector<int32_t> fill((size_t)n_pad, 0);
        for (int64_t k = 0; k < nnz; ++k) {
            int64_t row = row_map[m->I[(size_t)k]];
            place(k, fill[(size_t)row]++);
        }
    }
    if (bad) {
        delete s;
        return uspmv::fail(USPMV_ERR_INVALID,
                           "uspmv_convert_to_scs: fixed_permutation maps a non-empty row onto a padded slot "
                           "(the reference overruns its chunk here, code/utilities.hpp:1919-1922)");
    }
    *out = s;
    return USPMV_OK;
}

int uspmv_scs_meta(const uspmv_scs_t *s, int64_t meta[8]) {
    if (!s || !meta) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_scs_meta: NULL argument");
    meta[0] = s->C; meta[1] = s->sigma; meta[2] = s->n_rows; meta[3] = s->n_cols;
    meta[4] = s->n_rows_padded; meta[5] = s->n_chunks; meta[6] = s->n_elements; meta[7] = s->nnz;
    return USPMV_OK;
}

int uspmv_scs_dtype(const uspmv_scs_t *s, int *dtype) {
    if (!s || !dtype) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_scs_dtype: NULL argument");
    *dtype = s->dtype;
    return USPMV_OK;
}

int uspmv_scs_arrays(const uspmv_scs_t *s, const int32_t **chunk_ptrs, const int32_t **chunk_lengths,
                     const int32_t **col_idxs, const void **values, const int32_t **old_to_new_idx,
                     const int32_t **new_to_old_idx) {
    if (!s) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_scs_arrays: NULL matrix");
    if (chunk_ptrs) *chunk_ptrs = s->chunk_ptrs.data();
    if (chunk_lengths) *chunk_lengths = s->chunk_lengths.data();
    const bool full = uspmv::scs_has_entries(s);      // layout-only structs (uspmv_convert_to_scs_device): NULL
    if (col_idxs) *col_idxs = full ? s->col_idxs.data() : nullptr;
    if (values) *values = full ? s->values_ptr() : nullptr;
    if (old_to_new_idx) *old_to_new_idx = s->old_to_new_idx.data();
    if (new_to_old_idx) *new_to_old_idx = s->new_to_old_idx.data();
    return USPMV_OK;
}

int uspmv_scs_col_idxs_mut(uspmv_scs_t *s, int32_t **col_idxs) {
    if (!s || !col_idxs) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_scs_col_idxs_mut: NULL argument");
    *col_idxs = s->col_idxs.data();
    return USPMV_OK;
}

int uspmv_permute_scs_cols(uspmv_scs_t *s, const int32_t *perm) {
    if (!s || !perm) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_permute_scs_cols: NULL argument");
    if (!uspmv::scs_has_entries(s)) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_permute_scs_cols: layout-only struct (uspmv_convert_to_scs_device)");
    const int64_t n = s->n_elements, n_rows = s->n_rows;
    int32_t *ci = s->col_idxs.data();
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i)
        if (ci[i] < n_rows) ci[i] = perm[ci[i]];
    return USPMV_OK;
}

void uspmv_scs_free(uspmv_scs_t *s) { delete s; }

int uspmv_coo_equilibrate(uspmv_coo_t *m) {
    // equilibrate_matrix (code/utilities.hpp:2667-2685, -equilibrate 1 of a one-precision run): every value divided
    // by the largest magnitude of its row, then by the largest magnitude of its column in the row-scaled matrix.
    // max and division are order-independent, so the parallel sweeps give the reference's bits.
    if (!m) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_coo_equilibrate: NULL matrix");
    const int64_t nnz = m->nnz;
    for (int pass = 0; pass < 2; ++pass) {
        const uspmv_ivec &idx = pass == 0 ? m->I : m->J;
        std::vector<double> mx((size_t)std::max<int64_t>(pass == 0 ? m->n_rows : m->n_cols, 1), 0.0);
#pragma omp parallel
        {
            std::vector<double> local(mx.size(), 0.0);
#pragma omp for schedule(static) nowait
            for (int64_t k = 0; k < nnz; ++k) {
                const double a = std::fabs(m->values[(size_t)k]);
                if (a > local[(size_t)idx[(size_t)k]]) local[(size_t)idx[(size_t)k]] = a;
            }
#pragma omp critical
            for (size_t r = 0; r < mx.size(); ++r) mx[r] = std::max(mx[r], local[r]);
        }
#pragma omp parallel for schedule(static)
        for (int64_t k = 0; k < nnz; ++k) m->values[(size_t)k] = m->values[(size_t)k] / mx[(size_t)idx[(size_t)k]];
    }
    return USPMV_OK;
}

int uspmv_apply_permutation(void *out, const void *in, const int32_t *perm, int64_t n, int dtype) {
    if (!out || !in || !perm || n < 0) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_apply_permutation: bad argument");
    if (dtype == USPMV_F64) {
        auto *o = (double *)out; auto *i = (const double *)in;
#pragma omp parallel for schedule(static)
        for (int64_t k = 0; k < n; ++k) o[k] = i[perm[k]];
    } else if (dtype == USPMV_F32) {
        auto *o = (float *)out; auto *i = (const float *)in;
#pragma omp parallel for schedule(static)
        for (int64_t k = 0; k < n; ++k) o[k] = i[perm[k]];
    } else {
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_apply_permutation: unknown dtype %d", dtype);
    }
    return USPMV_OK;
}

int uspmv_partition_precisions(const uspmv_coo_t *m, double threshold_1, uspmv_coo_t **dp, uspmv_coo_t **sp) {
    if (!m || !dp || !sp) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_partition_precisions: NULL argument");
    auto *d = new uspmv_coo; auto *s = new uspmv_coo;
    d->n_rows = s->n_rows = m->n_rows;
    d->n_cols = s->n_cols = m->n_cols;
    for (int64_t k = 0; k < m->nnz; ++k) {
        double v = m->values[(size_t)k];
        if (std::fabs(v) >= threshold_1) {
            d->I.push_back(m->I[(size_t)k]); d->J.push_back(m->J[(size_t)k]); d->values.push_back(v);
        } else if (std::fabs(v) < threshold_1) {
            s->I.push_back(m->I[(size_t)k]); s->J.push_back(m->J[(size_t)k]);
            s->values.push_back((double)(float)v);  // static_cast<float> (code/utilities.hpp:2907)
        } else {  // NaN fits neither bucket (code/utilities.hpp:2912-2915)
            delete d; delete s;
            return uspmv::fail(USPMV_ERR_INVALID, "uspmv_partition_precisions: element %lld fits neither struct",
                               (long long)k);
        }
    }
    d->nnz = (int64_t)d->values.size(); s->nnz = (int64_t)s->values.size();
    *dp = d; *sp = s;
    return USPMV_OK;
}

}  // extern "C"
