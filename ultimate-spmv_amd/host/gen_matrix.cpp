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

// KKT-structured matrix of the nlpkkt class ([H A^T; A 0] of a PDE-constrained optimisation problem on an N^3 grid with boundary
// control -- the structure behind SuiteSparse nlpkkt200 / nlpkkt240, n = 2 N^3 + 6 N^2: 16 240 000 for N = 200, 27 993 600 for
// N = 240).  Unknowns, in this order:   states y (N^3), multipliers lambda (N^3), boundary controls u (6 N^2: face f, in-face (a, b)).
//   state row      : H_yy  self + x-neighbours (3)                   | A^T  the 25-point stencil of its multipliers (the 27-point cube
//                    without the corners (+,+,+) and (-,-,-): symmetric under negation) | H_yu  the controls of the faces it lies on
//   multiplier row : A     the same 25-point stencil of states       | (no diagonal: the zero block)       | B  the controls of its faces
//   control row    : H_uy its state | B^T its multiplier | H_uu self + in-face neighbours (2-4)
// Interior rows: 28 / 25 entries, control rows 5-7, i.e. rows of 5-28 entries whose columns live in TWO coupled index ranges
// N^3 apart -- the opposite of the friendly 27-point stencil stand-in: a 256-row tile of states touches a window of states and a
// window of multipliers.  Symmetric pattern and values (v(i,j) = v(j,i) = hash of the sorted pair, uniform in [-1,1); H diagonals
// 8 + u).  Columns ascend inside a row.  Rows [row_begin, row_end) are generated, row ids local, column ids global.
namespace {
struct Kkt {
    int64_t N, N2, N3, n;
    explicit Kkt(int64_t N_) : N(N_), N2(N_ * N_), N3(N_ * N_ * N_), n(2 * N_ * N_ * N_ + 6 * N_ * N_) {}
    static bool in_stencil(int dx, int dy, int dz) { return !((dx == 1 && dy == 1 && dz == 1) || (dx == -1 && dy == -1 && dz == -1)); }
    int64_t node(int64_t x, int64_t y, int64_t z) const { return x + N * (y + N * z); }
    // faces: 0 x=0, 1 x=N-1 (a=y, b=z); 2 y=0, 3 y=N-1 (a=x, b=z); 4 z=0, 5 z=N-1 (a=x, b=y)
    int64_t ctrl(int f, int64_t a, int64_t b) const { return 2 * N3 + (int64_t)f * N2 + a + N * b; }
    // controls of the faces node (x,y,z) lies on, ascending
    int faces(int64_t x, int64_t y, int64_t z, int64_t *out) const {
        int k = 0;
        if (x == 0) out[k++] = ctrl(0, y, z);
        if (x == N - 1) out[k++] = ctrl(1, y, z);
        if (y == 0) out[k++] = ctrl(2, x, z);
        if (y == N - 1) out[k++] = ctrl(3, x, z);
        if (z == 0) out[k++] = ctrl(4, x, y);
        if (z == N - 1) out[k++] = ctrl(5, x, y);
        return k;
    }
    // columns of a global row, ascending; returns the count (at most 34)
    int row_cols(int64_t row, int64_t *c) const {
        int k = 0;
        if (row < 2 * N3) {
            const bool state = row < N3;
            const int64_t nd = state ? row : row - N3;
            const int64_t x = nd % N, y = (nd / N) % N, z = nd / N2;
            if (state) {                                   // H_yy: x-neighbours and self
                if (x > 0) c[k++] = nd - 1;
                c[k++] = nd;
                if (x < N - 1) c[k++] = nd + 1;
            }
            int64_t st[27];
            int ns = 0;
            for (int dz = -1; dz <= 1; ++dz) {
                if (z + dz < 0 || z + dz >= N) continue;
                for (int dy = -1; dy <= 1; ++dy) {
                    if (y + dy < 0 || y + dy >= N) continue;
                    for (int dx = -1; dx <= 1; ++dx) {
                        if (x + dx < 0 || x + dx >= N || !in_stencil(dx, dy, dz)) continue;
                        st[ns++] = node(x + dx, y + dy, z + dz);
                    }
                }
            }
            // state row: the stencil addresses multipliers (A^T); multiplier row: states (A).  States come first in the numbering.
            if (state) for (int j = 0; j < ns; ++j) c[k++] = N3 + st[j];
            else { for (int j = 0; j < ns; ++j) c[k++] = st[j]; }
            k += faces(x, y, z, c + k);
            return k;
        }
        const int64_t cc = row - 2 * N3;
        const int f = (int)(cc / N2);
        const int64_t a = (cc % N2) % N, b = (cc % N2) / N;
        int64_t x, y, z;
        switch (f) {
            case 0: x = 0; y = a; z = b; break;
            case 1: x = N - 1; y = a; z = b; break;
            case 2: x = a; y = 0; z = b; break;
            case 3: x = a; y = N - 1; z = b; break;
            case 4: x = a; y = b; z = 0; break;
            default: x = a; y = b; z = N - 1; break;
        }
        const int64_t nd = node(x, y, z);
        c[k++] = nd;                 // H_uy
        c[k++] = N3 + nd;            // B^T
        if (b > 0) c[k++] = row - N;
        if (a > 0) c[k++] = row - 1;
        c[k++] = row;
        if (a < N - 1) c[k++] = row + 1;
        if (b < N - 1) c[k++] = row + N;
        return k;
    }
};
}  // namespace

extern "C" int uspmv_gen_kkt(int64_t N, uint64_t seed, int64_t row_begin, int64_t row_end, uspmv_coo_t **out) {
    if (!out || N < 2) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_gen_kkt: bad argument");
    const Kkt g(N);
    if (g.n > INT32_MAX) return uspmv::fail(USPMV_ERR_OVERFLOW, "uspmv_gen_kkt: %lld rows exceed int32", (long long)g.n);
    if (row_begin < 0 || row_end > g.n || row_begin >= row_end) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_gen_kkt: bad row range");
    const int64_t nloc = row_end - row_begin;
    std::vector<int64_t> start((size_t)nloc + 1, 0);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < nloc; ++r) {
        int64_t c[40];
        start[(size_t)r + 1] = g.row_cols(row_begin + r, c);
    }
    for (int64_t r = 0; r < nloc; ++r) start[(size_t)r + 1] += start[(size_t)r];
    const int64_t nnz = start[(size_t)nloc];
    if (nnz > INT32_MAX) return uspmv::fail(USPMV_ERR_OVERFLOW, "uspmv_gen_kkt: %lld local nnz exceed int32", (long long)nnz);
    auto *m = new uspmv_coo;
    m->n_rows = nloc; m->n_cols = g.n; m->nnz = nnz;
    m->I.resize((size_t)nnz); m->J.resize((size_t)nnz); m->values.resize((size_t)nnz);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < nloc; ++r) {
        const int64_t row = row_begin + r;
        int64_t c[40];
        const int k = g.row_cols(row, c);
        for (int j = 0; j < k; ++j) {
            const int64_t col = c[j];
            const uint64_t h = pair_hash((uint64_t)std::min(row, col), (uint64_t)std::max(row, col), seed);
            const size_t p = (size_t)(start[(size_t)r] + j);
            m->I[p] = (int32_t)r; m->J[p] = (int32_t)col;
            m->values[p] = col == row ? 8.0 + (2.0 * u01(h) - 1.0) : 2.0 * u01(h) - 1.0;
        }
    }
    *out = m;
    return USPMV_OK;
}

extern "C" int uspmv_gen_kkt_row_counts(int64_t N, int64_t row_begin, int64_t row_end, int32_t *out) {
    if (!out || N < 2) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_gen_kkt_row_counts: bad argument");
    const Kkt g(N);
    if (row_begin < 0 || row_end > g.n || row_begin > row_end) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_gen_kkt_row_counts: bad row range");
#pragma omp parallel for schedule(static)
    for (int64_t r = row_begin; r < row_end; ++r) {
        int64_t c[40];
        out[r - row_begin] = (int32_t)g.row_cols(r, c);
    }
    return USPMV_OK;
}
