// -seg_metis without METIS.
//
// The reference (code/mpi_funcs.hpp:494-598, USE_METIS builds) hands the matrix graph to METIS_PartGraphKway, sorts the rows by
// part (sortPerm = std::stable_sort on the part ids, code/utilities.hpp:1833-1840), applies that permutation symmetrically to the
// matrix (ScsData::permute, code/classes_structs.hpp:1620-1700: rows move, columns are renumbered, the order of a row's entries is
// kept) and derives work_sharing_arr from the part sizes.  METIS is not in this image, so two sources of the part vector stand in:
//   * uspmv_graph_partition -- a built-in partitioner: breadth-first level sets from a pseudo-peripheral vertex (every component in
//     turn) give an ordering in which neighbours sit close; P contiguous, equally sized pieces of it are the parts, followed by
//     boundary refinement passes (a vertex moves to the neighbouring part that holds more of its neighbours when the balance allows);
//   * uspmv_read_partition  -- a part vector from a file, one part id per row (the output format of gpmetis / kmetis), so a real
//     METIS run can be used when one is at hand.
// uspmv_coo_apply_partition is the reference's post-processing (stable sort, symmetric permutation, work_sharing_arr) for either.
#include <algorithm>
#include <cstdio>
#include <numeric>
#include <queue>

#include "uspmv_internal.hpp"

namespace {

struct Graph {   // symmetrised pattern without the diagonal
    std::vector<int64_t> ptr;
    std::vector<int32_t> adj;
};

Graph build_graph(const uspmv_coo *m) {
    const int64_t n = m->n_rows;
    Graph g;
    std::vector<int64_t> deg((size_t)n + 1, 0);
    for (int64_t k = 0; k < m->nnz; ++k) {
        const int32_t i = m->I[(size_t)k], j = m->J[(size_t)k];
        if (i == j || j >= n || j < 0 || i < 0 || i >= n) continue;   // (entries outside the square are no edges; uspmv_graph_partition refuses them up front)
        ++deg[(size_t)i + 1]; ++deg[(size_t)j + 1];
    }
    for (int64_t i = 0; i < n; ++i) deg[(size_t)i + 1] += deg[(size_t)i];
    std::vector<int32_t> raw((size_t)deg[(size_t)n]);
    std::vector<int64_t> pos(deg.begin(), deg.end() - 1);
    for (int64_t k = 0; k < m->nnz; ++k) {
        const int32_t i = m->I[(size_t)k], j = m->J[(size_t)k];
        if (i == j || j >= n || j < 0 || i < 0 || i >= n) continue;   // (entries outside the square are no edges; uspmv_graph_partition refuses them up front)
        raw[(size_t)pos[(size_t)i]++] = j; raw[(size_t)pos[(size_t)j]++] = i;
    }
    g.ptr.assign((size_t)n + 1, 0);
    g.adj.reserve(raw.size() / 2 + 1);
    for (int64_t i = 0; i < n; ++i) {     // sort + unique per vertex
        auto b = raw.begin() + deg[(size_t)i], e = raw.begin() + deg[(size_t)i + 1];
        std::sort(b, e);
        e = std::unique(b, e);
        g.adj.insert(g.adj.end(), b, e);
        g.ptr[(size_t)i + 1] = (int64_t)g.adj.size();
    }
    return g;
}

// breadth-first search from `start` over unvisited vertices; appends the visit order, returns the last vertex of the last level with
// the smallest degree (the next candidate for a pseudo-peripheral start)
int32_t bfs(const Graph &g, int32_t start, std::vector<char> &seen, std::vector<int32_t> &order, bool commit) {
    const size_t first = order.size();
    order.push_back(start);
    seen[(size_t)start] = 1;
    size_t level_begin = first, head = first;
    while (head < order.size()) {
        const size_t level_end = order.size();
        level_begin = head;
        for (; head < level_end; ++head) {
            const int32_t v = order[head];
            for (int64_t k = g.ptr[(size_t)v]; k < g.ptr[(size_t)v + 1]; ++k) {
                const int32_t w = g.adj[(size_t)k];
                if (!seen[(size_t)w]) { seen[(size_t)w] = 1; order.push_back(w); }
            }
        }
    }
    int32_t best = order[level_begin];
    for (size_t k = level_begin; k < order.size(); ++k) {
        const int32_t v = order[k];
        if (g.ptr[(size_t)v + 1] - g.ptr[(size_t)v] < g.ptr[(size_t)best + 1] - g.ptr[(size_t)best]) best = v;
    }
    if (!commit) {
        for (size_t k = first; k < order.size(); ++k) seen[(size_t)order[k]] = 0;
        order.resize(first);
    }
    return best;
}

}  // namespace

extern "C" {

int uspmv_graph_partition(const uspmv_coo_t *m, int P, int32_t *part) {
    if (!m || !part || P < 1) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_graph_partition: bad argument");
    if (m->n_rows != m->n_cols) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_graph_partition: the matrix must be square");
    const int64_t n = m->n_rows;
    if (n == 0) return USPMV_OK;
    for (int64_t k = 0; k < m->nnz; ++k)
        if (m->I[(size_t)k] < 0 || m->I[(size_t)k] >= n || m->J[(size_t)k] < 0 || m->J[(size_t)k] >= n)
            return uspmv::fail(USPMV_ERR_INVALID, "uspmv_graph_partition: entry %lld (%d, %d) lies outside the %lld x %lld matrix", (long long)k, m->I[(size_t)k],
                               m->J[(size_t)k], (long long)n, (long long)n);
    const Graph g = build_graph(m);
    std::vector<char> seen((size_t)n, 0);
    std::vector<int32_t> order;
    order.reserve((size_t)n);
    for (int64_t s = 0; s < n; ++s) {
        if (seen[(size_t)s]) continue;
        int32_t start = (int32_t)s;
        for (int sweep = 0; sweep < 2; ++sweep) start = bfs(g, start, seen, order, false);   // George-Liu: walk to a pseudo-peripheral vertex
        bfs(g, start, seen, order, true);
    }
    // P contiguous pieces of the ordering, sizes differing by at most one (METIS_PartGraphKway without weights balances vertex counts)
    std::vector<int64_t> size((size_t)P, 0);
    for (int64_t r = 0; r < n; ++r) {
        const int p = (int)std::min<int64_t>(P - 1, r * P / n);
        part[(size_t)order[(size_t)r]] = p; ++size[(size_t)p];
    }
    // boundary refinement: a vertex moves to the part that holds most of its neighbours when that lowers the cut and keeps every part
    // within 3 % (at least one vertex) of the mean
    const int64_t slack = std::max<int64_t>(1, n * 3 / (100 * (int64_t)P));
    const int64_t lo = std::max<int64_t>(1, n / P - slack), hi = (n + P - 1) / P + slack;   // (lo >= 1: refinement never empties a part)
    std::vector<int32_t> cnt((size_t)P, 0), touched;
    for (int pass = 0; pass < 4 && P > 1; ++pass) {
        int64_t moved = 0;
        for (int64_t r = 0; r < n; ++r) {
            const int32_t v = order[(size_t)r];
            const int32_t pv = part[(size_t)v];
            touched.clear();
            for (int64_t k = g.ptr[(size_t)v]; k < g.ptr[(size_t)v + 1]; ++k) {
                const int32_t q = part[(size_t)g.adj[(size_t)k]];
                if (cnt[(size_t)q]++ == 0) touched.push_back(q);
            }
            int32_t best = pv;
            for (int32_t q : touched)
                if (q != pv && cnt[(size_t)q] > cnt[(size_t)best] && size[(size_t)q] < hi && size[(size_t)pv] > lo) best = q;
            for (int32_t q : touched) cnt[(size_t)q] = 0;
            if (best != pv) { part[(size_t)v] = best; --size[(size_t)pv]; ++size[(size_t)best]; ++moved; }
        }
        if (!moved) break;
    }
    return USPMV_OK;
}

int uspmv_read_partition(const char *path, int64_t n_rows, int P, int32_t *part) {
    if (!path || !part || n_rows < 0 || P < 1) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_read_partition: bad argument");
    FILE *f = fopen(path, "r");
    if (!f) return uspmv::fail(USPMV_ERR_IO, "uspmv_read_partition: cannot open '%s'", path);
    for (int64_t i = 0; i < n_rows; ++i) {
        long v;
        if (fscanf(f, "%ld", &v) != 1) { fclose(f); return uspmv::fail(USPMV_ERR_IO, "uspmv_read_partition: '%s' holds fewer than %ld part ids", path, (long)n_rows); }
        if (v < 0 || v >= P) { fclose(f); return uspmv::fail(USPMV_ERR_INVALID, "uspmv_read_partition: part id %ld of row %ld is outside [0, %d)", v, (long)i, P); }
        part[(size_t)i] = (int32_t)v;
    }
    fclose(f);
    return USPMV_OK;
}

// the reference's post-processing of a part vector (code/mpi_funcs.hpp:529-598): perm = stable sort of the rows by part
// (new row r = old row perm[r]), symmetric permutation of the matrix (entry order inside a row kept), wsa = running part sizes,
// and the empty-last-rank fix-up of :602-606
int uspmv_coo_apply_partition(const uspmv_coo_t *m, int P, const int32_t *part, uspmv_coo_t **out, int32_t *wsa, int32_t *perm_out) {
    if (!m || !part || !out || !wsa || P < 1) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_coo_apply_partition: bad argument");
    if (m->n_rows != m->n_cols) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_coo_apply_partition: the matrix must be square");
    const int64_t n = m->n_rows, nnz = m->nnz;
    for (int64_t i = 0; i < n; ++i)
        if (part[(size_t)i] < 0 || part[(size_t)i] >= P) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_coo_apply_partition: part id %d of row %ld is outside [0, %d)", part[(size_t)i], (long)i, P);
    std::vector<int32_t> perm((size_t)n), inv((size_t)n);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) { return part[(size_t)a] < part[(size_t)b]; });
    for (int64_t r = 0; r < n; ++r) inv[(size_t)perm[(size_t)r]] = (int32_t)r;
    // rows of the COO (any order of rows, entries of a row in storage order) -> row-major under the new numbering
    std::vector<int64_t> start((size_t)n + 1, 0);
    for (int64_t k = 0; k < nnz; ++k) ++start[(size_t)inv[(size_t)m->I[(size_t)k]] + 1];
    for (int64_t r = 0; r < n; ++r) start[(size_t)r + 1] += start[(size_t)r];
    auto *o = new uspmv_coo;
    o->n_rows = n; o->n_cols = n; o->nnz = nnz;
    o->I.resize((size_t)nnz); o->J.resize((size_t)nnz); o->values.resize((size_t)nnz);
    std::vector<int64_t> pos(start.begin(), start.end() - 1);
    for (int64_t k = 0; k < nnz; ++k) {
        const int32_t r = inv[(size_t)m->I[(size_t)k]];
        const size_t d = (size_t)pos[(size_t)r]++;
        o->I[d] = r; o->J[d] = inv[(size_t)m->J[(size_t)k]]; o->values[d] = m->values[(size_t)k];
    }
    wsa[0] = 0;
    std::vector<int64_t> size((size_t)P, 0);
    for (int64_t i = 0; i < n; ++i) ++size[(size_t)part[(size_t)i]];
    for (int p = 0; p < P; ++p) wsa[p + 1] = (int32_t)(wsa[p] + size[(size_t)p]);
    if (P > 1 && wsa[P - 1] == wsa[P])       // "Protect against edge case where last process gets no work" (code/mpi_funcs.hpp:602-606)
        for (int p = 1; p < P; ++p) wsa[p] -= 1;
    if (perm_out) std::copy(perm.begin(), perm.end(), perm_out);
    *out = o;
    return USPMV_OK;
}

}  // extern "C"
