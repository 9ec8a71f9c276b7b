// Deterministic synthetic matrices standing in for the SuiteSparse inputs of the BASELINE
// configurations (nlpkkt200 / nlpkkt240 / Queen_4147 / HV15R are not available offline):
// SURVEY.md 8(d).  Not part of the reference; the reference's ScaMaC generator hook
// (code/utilities.hpp:1585-1752) plays the same role there.
//
// 27-point stencil, `dof` unknowns per node, on an nx*ny*nz grid.  Row = node*dof + d with
// node = x + nx*(y + ny*z).  Inside a row the columns ascend (that is the "file order" the kernels
// sum in).  Pattern and values are symmetric: v(i,j) = u(hash(min(i,j), max(i,j), seed)) in
// [-1,1); diagonal = 27*dof + u.  With magnitude_decades D > 0 the off-diagonal magnitudes are
// 10^(2 - D*u01) (log-uniform over D decades below 1e2) with a hashed sign, so that an
// -ap_threshold_1 split is non-trivial.
#include <algorithm>
#include <cmath>

#include "uspmv_internal.hpp"

namespace {
inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline uint64_t pair_hash(uint64_t a, uint64_t b, uint64_t seed) { return mix64(mix64(a ^ seed) + b); }
inline double u01(uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }
}  // namespace

extern "C" int uspmv_gen_stencil27(int64_t nx, int64_t ny, int64_t nz, int dof, uint64_t seed,
                                   double magnitude_decades, int64_t row_begin, int64_t row_end,
                                   uspmv_coo_t **out) {
    if (!out || nx < 1 || ny < 1 || nz < 1 || dof < 1)
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_gen_stencil27: bad argument");
    const int64_t n = nx * ny * nz * dof;
    if (n > INT32_MAX) return uspmv::fail(USPMV_ERR_OVERFLOW, "uspmv_gen_stencil27: %lld rows exceed int32", (long long)n);
    if (row_begin < 0 || row_end > n || row_begin >= row_end)
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_gen_stencil27: bad row range");
    const int64_t nloc = row_end - row_begin;
    std::vector<int64_t> start((size_t)nloc + 1, 0);
    auto span = [](int64_t c, int64_t nc) { return (int64_t)1 + (c > 0) + (c < nc - 1); };
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < nloc; ++r) {
        int64_t node = (row_begin + r) / dof;
        int64_t x = node % nx, y = (node / nx) % ny, z = node / (nx * ny);
        start[(size_t)r + 1] = span(x, nx) * span(y, ny) * span(z, nz) * dof;
    }
    for (int64_t r = 0; r < nloc; ++r) start[(size_t)r + 1] += start[(size_t)r];
    const int64_t nnz = start[(size_t)nloc];
    if (nnz > INT32_MAX)
        return uspmv::fail(USPMV_ERR_OVERFLOW, "uspmv_gen_stencil27: %lld local nnz exceed int32", (long long)nnz);
    auto *m = new uspmv_coo;
    m->n_rows = nloc; m->n_cols = n; m->nnz = nnz;
    m->I.resize((size_t)nnz); m->J.resize((size_t)nnz); m->values.resize((size_t)nnz);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < nloc; ++r) {
        const int64_t row = row_begin + r;
        const int64_t node = row / dof;
        const int64_t x = node % nx, y = (node / nx) % ny, z = node / (nx * ny);
        int64_t k = start[(size_t)r];
        for (int64_t dz = -1; dz <= 1; ++dz) {
            if (z + dz < 0 || z + dz >= nz) continue;
            for (int64_t dy = -1; dy <= 1; ++dy) {
                if (y + dy < 0 || y + dy >= ny) continue;
                for (int64_t dx = -1; dx <= 1; ++dx) {
                    if (x + dx < 0 || x + dx >= nx) continue;
                    const int64_t q = (x + dx) + nx * ((y + dy) + ny * (z + dz));
                    for (int e = 0; e < dof; ++e) {
                        const int64_t col = q * dof + e;
                        const uint64_t h = pair_hash((uint64_t)std::min(row, col), (uint64_t)std::max(row, col), seed);
                        double v;
                        if (col == row) {
                            v = 27.0 * dof + (2.0 * u01(h) - 1.0);
                        } else if (magnitude_decades > 0.0) {
                            const uint64_t h2 = mix64(h);
                            v = std::pow(10.0, 2.0 - magnitude_decades * u01(h));
                            if (h2 & 1) v = -v;
                        } else {
                            v = 2.0 * u01(h) - 1.0;
                        }
                        m->I[(size_t)k] = (int32_t)r;
                        m->J[(size_t)k] = (int32_t)col;
                        m->values[(size_t)k] = v;
                        ++k;
                    }
                }
            }
        }
    }
    *out = m;
    return USPMV_OK;
}

// entries per row of uspmv_gen_stencil27's matrix for rows [row_begin, row_end) (nothing else is generated)
extern "C" int uspmv_gen_stencil27_row_counts(int64_t nx, int64_t ny, int64_t nz, int dof, int64_t row_begin, int64_t row_end, int32_t *out) {
    if (!out || nx < 1 || ny < 1 || nz < 1 || dof < 1) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_gen_stencil27_row_counts: bad argument");
    const int64_t n = nx * ny * nz * dof;
    if (row_begin < 0 || row_end > n || row_begin > row_end) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_gen_stencil27_row_counts: bad row range");
    auto span = [](int64_t c, int64_t nc) { return (int64_t)1 + (c > 0) + (c < nc - 1); };
#pragma omp parallel for schedule(static)
    for (int64_t r = row_begin; r < row_end; ++r) {
        const int64_t node = r / dof, x = node % nx, y = (node / nx) % ny, z = node / (nx * ny);
        out[r - row_begin] = (int32_t)(span(x, nx) * span(y, ny) * span(z, nz) * dof);
    }
    return USPMV_OK;
}

// Banded-random matrix (the HV15R-class stand-in of SURVEY.md 8(d)): row i holds the diagonal plus
// nnz_per_row - 1 distinct columns drawn by hash from [i - band, i + band] (clipped), ascending inside the row;
// general (non-symmetric) pattern; off-diagonal magnitudes 10^(2 - D*u) with hashed sign when
// magnitude_decades D > 0, else uniform in [-1,1); diagonal = nnz_per_row + u.
extern "C" int uspmv_gen_banded_random(int64_t n, int nnz_per_row, int64_t band, uint64_t seed, double magnitude_decades,
                                       int64_t row_begin, int64_t row_end, uspmv_coo_t **out) {
    if (!out || n < 1 || nnz_per_row < 1 || band < 1) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_gen_banded_random: bad argument");
    if (n > INT32_MAX) return uspmv::fail(USPMV_ERR_OVERFLOW, "uspmv_gen_banded_random: %lld rows exceed int32", (long long)n);
    if (row_begin < 0 || row_end > n || row_begin >= row_end) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_gen_banded_random: bad row range");
    const int64_t nloc = row_end - row_begin;
    auto window = [&](int64_t row, int64_t &lo, int64_t &hi) { lo = std::max<int64_t>(0, row - band); hi = std::min<int64_t>(n - 1, row + band); };
    std::vector<int64_t> start((size_t)nloc + 1, 0);
    for (int64_t r = 0; r < nloc; ++r) {
        int64_t lo, hi; window(row_begin + r, lo, hi);
        start[(size_t)r + 1] = start[(size_t)r] + std::min<int64_t>(nnz_per_row, hi - lo + 1);
    }
    const int64_t nnz = start[(size_t)nloc];
    if (nnz > INT32_MAX) return uspmv::fail(USPMV_ERR_OVERFLOW, "uspmv_gen_banded_random: %lld local nnz exceed int32", (long long)nnz);
    auto *m = new uspmv_coo;
    m->n_rows = nloc; m->n_cols = n; m->nnz = nnz;
    m->I.resize((size_t)nnz); m->J.resize((size_t)nnz); m->values.resize((size_t)nnz);
#pragma omp parallel
    {
        std::vector<int64_t> cols;
#pragma omp for schedule(static)
        for (int64_t r = 0; r < nloc; ++r) {
            const int64_t row = row_begin + r;
            int64_t lo, hi; window(row, lo, hi);
            const int64_t k = start[(size_t)r + 1] - start[(size_t)r], w = hi - lo + 1;
            cols.clear(); cols.push_back(row);
            uint64_t h = pair_hash((uint64_t)row, 0x6a09e667f3bcc909ull, seed);
            while ((int64_t)cols.size() < k) {              // rejection of duplicates: k << window width in practice
                h = mix64(h);
                const int64_t c = lo + (int64_t)(h % (uint64_t)w);
                if (std::find(cols.begin(), cols.end(), c) == cols.end()) cols.push_back(c);
            }
            std::sort(cols.begin(), cols.end());
            for (int64_t j = 0; j < k; ++j) {
                const int64_t col = cols[(size_t)j];
                const uint64_t hv = pair_hash((uint64_t)row, (uint64_t)col, seed ^ 0xbb67ae8584caa73bull);
                double v;
                if (col == row) v = (double)nnz_per_row + (2.0 * u01(hv) - 1.0);
                else if (magnitude_decades > 0.0) { v = std::pow(10.0, 2.0 - magnitude_decades * u01(hv)); if (mix64(hv) & 1) v = -v; }
                else v = 2.0 * u01(hv) - 1.0;
                const size_t p = (size_t)(start[(size_t)r] + j);
                m->I[p] = (int32_t)r; m->J[p] = (int32_t)col; m->values[p] = v;
            }
        }
    }
    *out = m;
    return USPMV_OK;
}
