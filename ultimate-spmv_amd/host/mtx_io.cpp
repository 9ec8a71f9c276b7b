// MatrixMarket reader and COO container of libuspmv.so.
//
// Behavioural contract = reference read_mtx (code/utilities.hpp:2148-2309) on top of
// mm_read_unsymmetric_sparse (code/mmio.h:132-263) and mm_read_banner / mm_read_mtx_crd_size
// (code/mmio.cpp): coordinate files of field real | integer | pattern and symmetry general |
// symmetric; pattern entries get the value 0.01 (code/mmio.h:195-203); a symmetric file is
// expanded entry by entry -- (r,c,v) immediately followed by (c,r,v) when r != c
// (code/utilities.hpp:2213-2267); finally entries are STABLE-sorted by row only
// (code/utilities.hpp:2139-2146, :2278), so the column order inside a row is file order.
// That order is the floating-point summation order of every kernel.
//
// Own implementation: the file is slurped once and parsed with strtol/strtod (same correctly
// rounded decimal->binary conversion as the reference's fscanf("%lg")); the stable row sort is a
// counting sort.  Errors are returned, never exit()ed.
#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <memory>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "uspmv_internal.hpp"

namespace {

thread_local std::string g_last_error;

struct Slurp {
    char *p = nullptr;
    size_t n = 0;
    ~Slurp() { free(p); }
};

bool slurp(const char *path, Slurp &s) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz < 0) { fclose(f); return false; }
    s.p = (char *)malloc((size_t)sz + 1);
    if (!s.p) { fclose(f); return false; }
    s.n = fread(s.p, 1, (size_t)sz, f);
    s.p[s.n] = 0;
    fclose(f);
    return true;
}

std::string lower(std::string s) {
    for (auto &c : s) c = (char)tolower((unsigned char)c);
    return s;
}

}  // namespace

namespace uspmv {
int fail(int status, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return status;
}
}  // namespace uspmv

extern "C" {

const char *uspmv_last_error(void) { return g_last_error.c_str(); }

const char *uspmv_version(void) { return "uspmv-mi355x 0.1 (gfx950)"; }

const char *uspmv_status_string(int s) {
    switch (s) {
        case USPMV_OK: return "ok";
        case USPMV_ERR_INVALID: return "invalid argument";
        case USPMV_ERR_IO: return "i/o error";
        case USPMV_ERR_UNSUPPORTED: return "unsupported";
        case USPMV_ERR_OVERFLOW: return "32-bit index overflow";
        case USPMV_ERR_NO_DEVICE: return "no HIP device";
        case USPMV_ERR_HIP: return "HIP runtime error";
        case USPMV_ERR_ALLOC: return "allocation failed";
        case USPMV_ERR_COMM: return "communication failure (peer rank)";
    }
    return "unknown status";
}

int uspmv_coo_create(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t *I, const int32_t *J,
                     const double *values, uspmv_coo_t **out) {
    if (!out || n_rows < 0 || n_cols < 0 || nnz < 0 || (nnz > 0 && (!I || !J || !values)))
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_coo_create: bad argument");
    for (int64_t k = 0; k < nnz; ++k)
        if (I[k] < 0 || I[k] >= n_rows || J[k] < 0 || J[k] >= n_cols)
            return uspmv::fail(USPMV_ERR_INVALID, "uspmv_coo_create: entry %lld (%d,%d) outside %lldx%lld",
                               (long long)k, I[k], J[k], (long long)n_rows, (long long)n_cols);
    auto *m = new uspmv_coo;
    m->n_rows = n_rows; m->n_cols = n_cols; m->nnz = nnz;
    m->I.assign(I, I + nnz); m->J.assign(J, J + nnz); m->values.assign(values, values + nnz);
    *out = m;
    return USPMV_OK;
}

int uspmv_coo_dims(const uspmv_coo_t *m, int64_t *n_rows, int64_t *n_cols, int64_t *nnz) {
    if (!m) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_coo_dims: NULL matrix");
    if (n_rows) *n_rows = m->n_rows;
    if (n_cols) *n_cols = m->n_cols;
    if (nnz) *nnz = m->nnz;
    return USPMV_OK;
}

int uspmv_coo_arrays(const uspmv_coo_t *m, const int32_t **I, const int32_t **J, const double **values) {
    if (!m) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_coo_arrays: NULL matrix");
    if (I) *I = m->I.data();
    if (J) *J = m->J.data();
    if (values) *values = m->values.data();
    return USPMV_OK;
}

void uspmv_coo_free(uspmv_coo_t *m) { delete m; }

// Binary COO cache: header {magic, version, n_rows, n_cols, nnz} + I, J, values as stored in
// the handle (already expanded and row-sorted), so that a second run skips the text parse.
namespace {
constexpr uint64_t COO_MAGIC = 0x4f4f43564d505355ull;   // "USPMVCOO"
struct CooHeader { uint64_t magic, version; int64_t n_rows, n_cols, nnz; };
}

int uspmv_coo_save(const uspmv_coo_t *m, const char *path) {
    if (!m || !path) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_coo_save: NULL argument");
    FILE *f = fopen(path, "wb");
    if (!f) return uspmv::fail(USPMV_ERR_IO, "uspmv_coo_save: cannot create '%s'", path);
    const CooHeader h{COO_MAGIC, 1, m->n_rows, m->n_cols, m->nnz};
    const size_t nz = (size_t)m->nnz;
    bool ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(m->I.data(), 4, nz, f) == nz && fwrite(m->J.data(), 4, nz, f) == nz &&
              fwrite(m->values.data(), 8, nz, f) == nz;
    ok = (fclose(f) == 0) && ok;
    if (!ok) { remove(path); return uspmv::fail(USPMV_ERR_IO, "uspmv_coo_save: short write to '%s'", path); }
    return USPMV_OK;
}

int uspmv_coo_load(const char *path, uspmv_coo_t **out) {
    if (!path || !out) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_coo_load: NULL argument");
    FILE *f = fopen(path, "rb");
    if (!f) return uspmv::fail(USPMV_ERR_IO, "uspmv_coo_load: cannot open '%s'", path);
    CooHeader h{};
    if (fread(&h, sizeof h, 1, f) != 1 || h.magic != COO_MAGIC || h.version != 1 || h.n_rows < 0 || h.n_cols < 0 || h.nnz < 0 ||
        h.n_rows > INT32_MAX || h.n_cols > INT32_MAX) {
        fclose(f);
        return uspmv::fail(USPMV_ERR_IO, "uspmv_coo_load: '%s' is not a uspmv COO cache", path);
    }
    auto *m = new uspmv_coo;
    m->n_rows = h.n_rows; m->n_cols = h.n_cols; m->nnz = h.nnz;
    const size_t nz = (size_t)h.nnz;
    m->I.resize(nz); m->J.resize(nz); m->values.resize(nz);
    bool ok = fread(m->I.data(), 4, nz, f) == nz && fread(m->J.data(), 4, nz, f) == nz && fread(m->values.data(), 8, nz, f) == nz;
    fclose(f);
    for (size_t k = 0; ok && k < nz; ++k) ok = m->I[k] >= 0 && m->I[k] < h.n_rows && m->J[k] >= 0 && m->J[k] < h.n_cols;
    if (!ok) { delete m; return uspmv::fail(USPMV_ERR_IO, "uspmv_coo_load: '%s' is truncated or corrupt", path); }
    *out = m;
    return USPMV_OK;
}

int uspmv_read_mtx(const char *path, uspmv_coo_t **out) {
    if (!path || !out) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_read_mtx: NULL argument");
    Slurp s;
    if (!slurp(path, s)) return uspmv::fail(USPMV_ERR_IO, "uspmv_read_mtx: cannot open '%s'", path);
    char *p = s.p, *end = s.p + s.n;

    // ---- banner: %%MatrixMarket matrix coordinate <field> <symmetry>   (code/mmio.cpp mm_read_banner)
    char *eol = (char *)memchr(p, '\n', (size_t)(end - p));
    std::string banner(p, eol ? (size_t)(eol - p) : (size_t)(end - p));
    char t0[64] = "", t1[64] = "", t2[64] = "", t3[64] = "", t4[64] = "";
    if (sscanf(banner.c_str(), "%63s %63s %63s %63s %63s", t0, t1, t2, t3, t4) != 5 ||
        strncmp(t0, "%%MatrixMarket", 14) != 0)
        return uspmv::fail(USPMV_ERR_IO, "uspmv_read_mtx: '%s' has no MatrixMarket banner", path);
    std::string obj = lower(t1), fmt = lower(t2), field = lower(t3), sym = lower(t4);
    if (obj != "matrix" || fmt != "coordinate")
        return uspmv::fail(USPMV_ERR_UNSUPPORTED, "uspmv_read_mtx: matrix has to be sparse (coordinate)");
    bool pattern = field == "pattern";
    if (!(field == "real" || field == "integer" || pattern))
        return uspmv::fail(USPMV_ERR_UNSUPPORTED, "uspmv_read_mtx: matrix has to be real, integer or pattern");
    bool symmetric = sym == "symmetric";
    if (!(symmetric || sym == "general"))
        return uspmv::fail(USPMV_ERR_UNSUPPORTED, "uspmv_read_mtx: matrix has to be either general or symmetric");
    p = eol ? eol + 1 : end;

    // ---- comments, then the size line
    long M = 0, N = 0, NZ = 0;
    for (;;) {
        if (p >= end) return uspmv::fail(USPMV_ERR_IO, "uspmv_read_mtx: could not parse matrix size");
        eol = (char *)memchr(p, '\n', (size_t)(end - p));
        char *next = eol ? eol + 1 : end;
        if (*p != '%') {
            char *q = p;
            errno = 0;
            M = strtol(q, &q, 10); N = strtol(q, &q, 10); NZ = strtol(q, &q, 10);
            if (M > 0 || N > 0 || NZ > 0) { p = next; break; }
        }
        p = next;
    }
    if (M != N)
        return uspmv::fail(USPMV_ERR_UNSUPPORTED,
                           "uspmv_read_mtx: matrix not square (%ldx%ld); only square matrices are supported", M, N);
    if (M > INT32_MAX || NZ > INT32_MAX)
        return uspmv::fail(USPMV_ERR_OVERFLOW, "uspmv_read_mtx: dimensions exceed the 32-bit index type");

    // ---- entries (1-based -> 0-based), symmetric expansion interleaved.  The entry section is cut
    // into one piece per thread at line boundaries; pieces are parsed in parallel and concatenated in
    // file order, so the result is identical to a sequential fscanf loop (the reference
    // parses 7.6e8 entries with one fscanf each on rank 0).
    std::vector<int32_t> ru, cu;
    std::vector<double> vu;
    {
        int nth = 1;
#ifdef _OPENMP
        nth = omp_get_max_threads();
#endif
        const size_t body = (size_t)(end - p);
        if (body < (1u << 20)) nth = 1;
        std::vector<char *> cut((size_t)nth + 1);
        cut[0] = p; cut[(size_t)nth] = end;
        for (int t = 1; t < nth; ++t) {
            char *q = p + body * (size_t)t / (size_t)nth;
            while (q < end && *q != '\n') ++q;
            cut[(size_t)t] = q < end ? q + 1 : end;
        }
        std::vector<std::vector<int32_t>> pr((size_t)nth), pc((size_t)nth);
        std::vector<std::vector<double>> pv((size_t)nth);
        std::vector<long> n_parsed((size_t)nth, 0);
        std::vector<int> err((size_t)nth, 0);
#pragma omp parallel for schedule(static, 1) num_threads(nth)
        for (int t = 0; t < nth; ++t) {
            char *q = cut[(size_t)t], *e = cut[(size_t)t + 1];
            auto &R = pr[(size_t)t]; auto &Cc = pc[(size_t)t]; auto &V = pv[(size_t)t];
            const size_t guess = (size_t)(e - q) / 12 + 16;
            R.reserve(guess * (symmetric ? 2 : 1)); Cc.reserve(guess * (symmetric ? 2 : 1)); V.reserve(guess * (symmetric ? 2 : 1));
            while (q < e) {
                while (q < e && (*q == ' ' || *q == '\t' || *q == '\r' || *q == '\n')) ++q;
                if (q >= e) break;
                char *r2;
                long r = strtol(q, &r2, 10);
                if (r2 == q) { err[(size_t)t] = 1; break; }
                q = r2;
                long c = strtol(q, &r2, 10);
                if (r2 == q) { err[(size_t)t] = 1; break; }
                q = r2;
                double v = 0.01;  // pattern matrices (code/mmio.h:195-203)
                if (!pattern) {
                    v = strtod(q, &r2);
                    if (r2 == q) { err[(size_t)t] = 1; break; }
                    q = r2;
                }
                if (r < 1 || r > M || c < 1 || c > N) { err[(size_t)t] = 2; break; }
                R.push_back((int32_t)(r - 1)); Cc.push_back((int32_t)(c - 1)); V.push_back(v);
                if (symmetric && r != c) { R.push_back((int32_t)(c - 1)); Cc.push_back((int32_t)(r - 1)); V.push_back(v); }
                ++n_parsed[(size_t)t];
            }
        }
        long total = 0;
        for (int t = 0; t < nth; ++t) {
            if (err[(size_t)t] == 2) return uspmv::fail(USPMV_ERR_IO, "uspmv_read_mtx: an entry lies outside the %ldx%ld matrix", M, N);
            if (err[(size_t)t]) return uspmv::fail(USPMV_ERR_IO, "uspmv_read_mtx: malformed entry line");
            total += n_parsed[(size_t)t];
        }
        if (total < NZ) return uspmv::fail(USPMV_ERR_IO, "uspmv_read_mtx: premature end of file (%ld of %ld entries)", total, NZ);
        // keep exactly the first NZ entries of the file (what the reference's counted loop reads)
        ru.reserve((size_t)NZ * (symmetric ? 2 : 1)); cu.reserve(ru.capacity()); vu.reserve(ru.capacity());
        long seen = 0;
        for (int t = 0; t < nth; ++t) {
            const auto &R = pr[(size_t)t]; const auto &Cc = pc[(size_t)t]; const auto &V = pv[(size_t)t];
            if (seen + n_parsed[(size_t)t] <= NZ) {
                ru.insert(ru.end(), R.begin(), R.end()); cu.insert(cu.end(), Cc.begin(), Cc.end()); vu.insert(vu.end(), V.begin(), V.end());
                seen += n_parsed[(size_t)t];
            } else {  // file holds more lines than the header announces: take entries until NZ is reached
                size_t k = 0;
                while (seen < NZ && k < R.size()) {
                    const bool pair = symmetric && k + 1 < R.size() && R[k] == Cc[k + 1] && Cc[k] == R[k + 1] && R[k] != Cc[k];
                    ru.push_back(R[k]); cu.push_back(Cc[k]); vu.push_back(V[k]); ++k;
                    if (pair) { ru.push_back(R[k]); cu.push_back(Cc[k]); vu.push_back(V[k]); ++k; }
                    ++seen;
                }
                break;
            }
        }
    }
    if (ru.size() > (size_t)INT32_MAX)
        return uspmv::fail(USPMV_ERR_OVERFLOW, "uspmv_read_mtx: expanded nnz exceeds the 32-bit index type");

    // ---- stable sort by row == counting sort
    int64_t nnz = (int64_t)ru.size();
    std::vector<int64_t> start((size_t)M + 1, 0);
    for (int64_t k = 0; k < nnz; ++k) start[(size_t)ru[k] + 1]++;
    for (long r = 0; r < M; ++r) start[(size_t)r + 1] += start[(size_t)r];
    auto *m = new uspmv_coo;
    m->n_rows = M; m->n_cols = N; m->nnz = nnz;
    m->I.resize((size_t)nnz); m->J.resize((size_t)nnz); m->values.resize((size_t)nnz);
    for (int64_t k = 0; k < nnz; ++k) {
        int64_t d = start[(size_t)ru[k]]++;
        m->I[(size_t)d] = ru[k]; m->J[(size_t)d] = cu[k]; m->values[(size_t)d] = vu[k];
    }
    *out = m;
    return USPMV_OK;
}

}  // extern "C"
