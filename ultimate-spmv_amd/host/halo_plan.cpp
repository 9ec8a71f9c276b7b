// Row-block partitioning and halo (remote x element) discovery -- pure integer set-up.
//
// Contract = reference code/mpi_funcs.hpp: seg_work_sharing_arr (:424-622, seg-rows / seg-nnz),
// seg_mtx_struct (:636-674) + localize_row_idx (:862-877), collect_local_needed_heri (:242-415).
// Outputs are bit-identical to the reference's for the same (matrix, P, C, sigma); the golden
// vectors in tests/golden/halo.npz come from the reference code itself.
//
// Numbering produced on rank r (SURVEY.md 8a "vector-layout contract"):
//   [0, n_local)                    local columns (col - wsa[r]; later permuted by permute_scs_cols)
//   [n_local, n_local + n_halo)     halo columns, grouped by owner rank ascending (lower ranks
//                                   first, then higher ranks), first-seen order inside an owner,
//                                   where "seen" scans col_idxs in SCS STORAGE order -- padding
//                                   entries (column 0) included, so every rank > 0 with padding
//                                   requests global column 0 (reference quirk, kept for parity).
// Own implementation: O(n_elements + n_cols) with a dense first-seen table instead of the
// reference's unordered_set + map + per-element owner search.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "uspmv_internal.hpp"

extern "C" {

int uspmv_seg_work_sharing_arr(const uspmv_coo_t *t, int seg_method, int P, int32_t *wsa) {
    if (!t || !wsa || P < 1) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_seg_work_sharing_arr: bad argument");
    if (t->n_rows < P)  // code/mpi_funcs.hpp:442-444
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_seg_work_sharing_arr: n_rows < number of ranks");
    if (t->nnz < 1) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_seg_work_sharing_arr: empty matrix");
    const int32_t last_row_p1 = t->I[(size_t)t->nnz - 1] + 1;
    wsa[0] = 0;
    if (seg_method == USPMV_SEG_ROWS) {
        const int64_t per = t->n_rows / P;
        for (int s = 1; s <= P; ++s) wsa[s] = (int32_t)(s * per);
        wsa[P] = last_row_p1;
    } else if (seg_method == USPMV_SEG_NNZ) {
        for (int s = 1; s <= P; ++s) wsa[s] = 0;
        const int64_t per = t->nnz / P;
        int64_t local = 0;
        int seg = 1;
        for (int64_t g = 0; g < t->nnz; ++g) {
            if (local == per) {  // cut AFTER the row in which the running count reached nnz/P
                if (seg <= P) wsa[seg] = t->I[(size_t)g] + 1;
                ++seg; local = 0;
                continue;
            }
            ++local;
        }
        wsa[P] = last_row_p1;
    } else {
        return uspmv::fail(USPMV_ERR_UNSUPPORTED, "uspmv_seg_work_sharing_arr: unknown seg method %d", seg_method);
    }
    if (P > 1 && wsa[P - 1] == wsa[P])  // last rank would be empty (code/mpi_funcs.hpp:602-606)
        for (int r = 1; r < P; ++r) wsa[r] -= 1;
    for (int i = 1; i <= P; ++i)
        if (wsa[i] < wsa[i - 1])
            return uspmv::fail(USPMV_ERR_INVALID, "uspmv_seg_work_sharing_arr: flaw in work_sharing_arr");
    return USPMV_OK;
}

// The same rule from per-row entry counts only, so that the ranks of a distributed run can agree on the partition of a
// generated matrix without anybody materialising it: in uspmv_seg_work_sharing_arr's entry loop every segment consumes
// nnz/P counted entries plus the one at which the cut is taken, so cut k falls after the row of entry k*(nnz/P + 1) - 1.
int uspmv_seg_from_row_counts(const int32_t *row_nnz, int64_t n_rows, int seg_method, int P, int32_t *wsa) {
    if (!row_nnz || !wsa || P < 1 || n_rows < 1) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_seg_from_row_counts: bad argument");
    if (n_rows < P) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_seg_from_row_counts: n_rows < number of ranks");
    if (n_rows > INT32_MAX) return uspmv::fail(USPMV_ERR_OVERFLOW, "uspmv_seg_from_row_counts: rows exceed int32");
    int64_t nnz = 0, last_row_p1 = 0;
    for (int64_t r = 0; r < n_rows; ++r) { nnz += row_nnz[r]; if (row_nnz[r] > 0) last_row_p1 = r + 1; }
    if (nnz < 1) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_seg_from_row_counts: empty matrix");
    for (int s = 0; s <= P; ++s) wsa[s] = 0;
    if (seg_method == USPMV_SEG_ROWS) {
        const int64_t per = n_rows / P;
        for (int s = 1; s <= P; ++s) wsa[s] = (int32_t)(s * per);
    } else if (seg_method == USPMV_SEG_NNZ) {
        const int64_t per = nnz / P;
        int64_t seen = 0, r = 0;
        for (int k = 1; k <= P; ++k) {
            const int64_t g = (int64_t)k * (per + 1) - 1;       // entry index at which cut k is taken
            if (g >= nnz) break;
            while (seen + row_nnz[r] <= g) seen += row_nnz[r++];
            wsa[k] = (int32_t)(r + 1);
        }
    } else {
        return uspmv::fail(USPMV_ERR_UNSUPPORTED, "uspmv_seg_from_row_counts: unknown seg method %d", seg_method);
    }
    wsa[P] = (int32_t)last_row_p1;
    if (P > 1 && wsa[P - 1] == wsa[P])
        for (int r = 1; r < P; ++r) wsa[r] -= 1;
    for (int i = 1; i <= P; ++i)
        if (wsa[i] < wsa[i - 1]) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_seg_from_row_counts: flaw in work_sharing_arr");
    return USPMV_OK;
}

int uspmv_seg_local_coo(const uspmv_coo_t *t, const int32_t *wsa, int rank, uspmv_coo_t **out) {
    if (!t || !wsa || !out || rank < 0) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_seg_local_coo: bad argument");
    const int32_t lo = wsa[rank], hi = wsa[rank + 1];
    // entries are row-sorted: [first entry with row >= lo, first entry with row >= hi)
    auto b = std::lower_bound(t->I.begin(), t->I.end(), lo);
    auto e = std::lower_bound(t->I.begin(), t->I.end(), hi);
    auto *m = new uspmv_coo;
    m->n_rows = hi - lo;  // == distinct-row count of the reference when the block has no empty row
    m->n_cols = t->n_cols;
    m->nnz = e - b;
    size_t o = (size_t)(b - t->I.begin());
    m->I.resize((size_t)m->nnz); m->J.assign(t->J.begin() + o, t->J.begin() + o + m->nnz);
    m->values.assign(t->values.begin() + o, t->values.begin() + o + m->nnz);
    for (int64_t k = 0; k < m->nnz; ++k) m->I[(size_t)k] = t->I[o + (size_t)k] - lo;
    *out = m;
    return USPMV_OK;
}

int uspmv_halo_discover(uspmv_scs_t *s, const int32_t *wsa, int rank, int P, uspmv_halo_t **out) {
    if (!s || !wsa || !out || P < 1 || rank < 0 || rank >= P)
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_halo_discover: bad argument");
    if (!uspmv::scs_has_entries(s)) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_halo_discover: layout-only struct (uspmv_convert_to_scs_device)");
    const int32_t lo = wsa[rank], hi = wsa[rank + 1];
    const int64_t n_cols = std::max<int64_t>(s->n_cols, wsa[P]);
    const int64_t n_el = s->n_elements;
    int32_t *ci = s->col_idxs.data();

    std::vector<int32_t> slot((size_t)n_cols, -1);  // first-seen position inside the owner's list
    std::vector<std::vector<int32_t>> lists((size_t)P);
    auto owner_of = [&](int32_t col) {
        return (int)(std::upper_bound(wsa, wsa + P + 1, col) - wsa) - 1;
    };
    for (int64_t k = 0; k < n_el; ++k) {
        int32_t col = ci[k];
        if (col < 0 || col >= n_cols)
            return uspmv::fail(USPMV_ERR_INVALID, "uspmv_halo_discover: column %d outside the global matrix", col);
        if (col >= lo && col < hi) continue;
        if (slot[(size_t)col] >= 0) continue;
        int p = owner_of(col);
        if (p < 0 || p >= P) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_halo_discover: column %d has no owner", col);
        slot[(size_t)col] = (int32_t)lists[(size_t)p].size();
        lists[(size_t)p].push_back(col - wsa[p]);
    }
    auto *h = new uspmv_halo;
    h->P = P; h->rank = rank; h->n_local = hi - lo;
    std::vector<int64_t> base((size_t)P + 1, 0);
    for (int p = 0; p < P; ++p) base[(size_t)p + 1] = base[(size_t)p] + (int64_t)lists[(size_t)p].size();
    h->n_halo = base[(size_t)P];
    if (h->n_local + h->n_halo > INT32_MAX) {
        delete h;
        return uspmv::fail(USPMV_ERR_OVERFLOW, "uspmv_halo_discover: local + halo columns exceed int32");
    }
    const int64_t n_local = h->n_local;
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < n_el; ++k) {
        int32_t col = ci[k];
        if (col >= lo && col < hi) { ci[k] = col - lo; continue; }
        int p = owner_of(col);
        ci[k] = (int32_t)(n_local + base[(size_t)p] + slot[(size_t)col]);
    }
    h->recv_counts.resize((size_t)P);
    for (int p = 0; p < P; ++p) {
        h->recv_counts[(size_t)p] = (int32_t)lists[(size_t)p].size();
        h->recv_idxs.insert(h->recv_idxs.end(), lists[(size_t)p].begin(), lists[(size_t)p].end());
    }
    // recv_counts_cumsum assembled exactly as code/mpi_funcs.hpp:403-414 does (entries below
    // `rank` from the lower-owner cumsum, entries rank..P from lower total + higher-owner cumsum)
    h->recv_counts_cumsum.assign((size_t)P + 1, 0);
    std::vector<int64_t> lhs((size_t)P + 1, 0), rhs((size_t)P + 1, 0);
    for (int p = 1; p <= P; ++p) {
        lhs[(size_t)p] = lhs[(size_t)p - 1] + ((p - 1) < rank ? h->recv_counts[(size_t)p - 1] : 0);
        rhs[(size_t)p] = rhs[(size_t)p - 1] + ((p - 1) > rank ? h->recv_counts[(size_t)p - 1] : 0);
    }
    for (int p = 0; p < rank; ++p) h->recv_counts_cumsum[(size_t)p] = (int32_t)lhs[(size_t)p];
    int limit = (P == 1) ? P - (rank + 1) : P - rank + 1;
    for (int i = 0; i < limit; ++i)
        h->recv_counts_cumsum[(size_t)(rank + i)] = (int32_t)(lhs[(size_t)rank] + rhs[(size_t)(rank + i)]);
    *out = h;
    return USPMV_OK;
}

int uspmv_halo_meta(const uspmv_halo_t *h, int64_t *n_halo, const int32_t **recv_counts_cumsum,
                    const int32_t **recv_idxs, const int32_t **recv_counts) {
    if (!h) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_halo_meta: NULL plan");
    if (n_halo) *n_halo = h->n_halo;
    if (recv_counts_cumsum) *recv_counts_cumsum = h->recv_counts_cumsum.data();
    if (recv_idxs) *recv_idxs = h->recv_idxs.data();
    if (recv_counts) *recv_counts = h->recv_counts.data();
    return USPMV_OK;
}

void uspmv_halo_free(uspmv_halo_t *h) { delete h; }

int uspmv_scs_split_chunks(const uspmv_scs_t *s, int64_t n_local, int32_t **interior, int64_t *n_interior,
                           int32_t **boundary, int64_t *n_boundary) {
    if (!s || !interior || !boundary || !n_interior || !n_boundary)
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_scs_split_chunks: NULL argument");
    if (!uspmv::scs_has_entries(s)) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_scs_split_chunks: layout-only struct (uspmv_convert_to_scs_device)");
    const int64_t nc = s->n_chunks, C = s->C;
    std::vector<uint8_t> is_bnd((size_t)nc, 0);
    const int32_t *ci = s->col_idxs.data();
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < nc; ++c) {
        int64_t b = s->chunk_ptrs[(size_t)c], e = b + (int64_t)s->chunk_lengths[(size_t)c] * C;
        uint8_t f = 0;
        for (int64_t k = b; k < e; ++k) f |= (ci[k] >= n_local);
        is_bnd[(size_t)c] = f;
    }
    int64_t nb = 0;
    for (int64_t c = 0; c < nc; ++c) nb += is_bnd[(size_t)c];
    auto *in = (int32_t *)malloc(sizeof(int32_t) * (size_t)std::max<int64_t>(nc - nb, 1));
    auto *bd = (int32_t *)malloc(sizeof(int32_t) * (size_t)std::max<int64_t>(nb, 1));
    if (!in || !bd) { free(in); free(bd); return uspmv::fail(USPMV_ERR_ALLOC, "uspmv_scs_split_chunks: out of memory"); }
    int64_t a = 0, b2 = 0;
    for (int64_t c = 0; c < nc; ++c) {
        if (is_bnd[(size_t)c]) bd[b2++] = (int32_t)c;
        else in[a++] = (int32_t)c;
    }
    *interior = in; *n_interior = a; *boundary = bd; *n_boundary = b2;
    return USPMV_OK;
}

void uspmv_free(void *p) { free(p); }

}  // extern "C"

// Per chunk: 0 = no halo column; 1 = halo columns only through entries that are +0.0 on ONE column (*pad_col) -- the padding the
// reference fills chunks with (value 0, column 0: code/utilities.hpp:1991-2002), which is a halo column on every rank but the first
// (code/mpi_funcs.hpp:279-306); 2 = some other halo reference.  *pad_col = -1 when no chunk is of class 1.
int uspmv_scs_classify_chunks(const uspmv_scs_t *s, int64_t n_local, std::vector<uint8_t> *cls, int32_t *pad_col) {
    if (!s || !cls || !pad_col) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_scs_classify_chunks: NULL argument");
    if (!uspmv::scs_has_entries(s)) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_scs_classify_chunks: layout-only struct");
    const int64_t nc = s->n_chunks, C = s->C;
    const int32_t *ci = s->col_idxs.data();
    const bool dp = s->dtype == USPMV_F64;
    auto pos_zero = [&](int64_t k) -> bool {
        if (dp) { uint64_t b; memcpy(&b, &s->values_f64[(size_t)k], 8); return b == 0; }
        uint32_t b; memcpy(&b, &s->values_f32[(size_t)k], 4); return b == 0;
    };
    // the padding column: the lowest halo column that carries a +0.0 entry
    int32_t h0 = INT32_MAX;
#pragma omp parallel for schedule(static) reduction(min : h0)
    for (int64_t c = 0; c < nc; ++c) {
        const int64_t b = s->chunk_ptrs[(size_t)c], e = b + (int64_t)s->chunk_lengths[(size_t)c] * C;
        for (int64_t k = b; k < e; ++k)
            if (ci[k] >= n_local && ci[k] < h0 && pos_zero(k)) h0 = ci[k];
    }
    cls->assign((size_t)nc, 0);
    int64_t n_pad = 0;
#pragma omp parallel for schedule(static) reduction(+ : n_pad)
    for (int64_t c = 0; c < nc; ++c) {
        const int64_t b = s->chunk_ptrs[(size_t)c], e = b + (int64_t)s->chunk_lengths[(size_t)c] * C;
        uint8_t f = 0;
        for (int64_t k = b; k < e && f < 2; ++k)
            if (ci[k] >= n_local) f = (ci[k] == h0 && pos_zero(k)) ? std::max<uint8_t>(f, 1) : 2;
        (*cls)[(size_t)c] = f;
        n_pad += f == 1;
    }
    *pad_col = n_pad ? h0 : -1;
    return USPMV_OK;
}

extern "C" int uspmv_scs_chunk_classes(const uspmv_scs_t *s, int64_t n_local, uint8_t *classes, int32_t *pad_col) {
    if (!classes || !pad_col) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_scs_chunk_classes: NULL argument");
    std::vector<uint8_t> cls;
    if (int rc = uspmv_scs_classify_chunks(s, n_local, &cls, pad_col)) return rc;
    if (!cls.empty()) memcpy(classes, cls.data(), cls.size());
    return USPMV_OK;
}
