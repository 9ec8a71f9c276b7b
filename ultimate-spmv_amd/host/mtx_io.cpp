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
#include <charconv>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <memory>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "uspmv_internal.hpp"

namespace {

thread_local std::string g_last_error;

// scratch array on anonymous memory with transparent huge pages asked for: 2 MiB pages take 512 times fewer page faults than 4 KiB ones
// (eight parsing threads faulting in one address space serialise on the kernel's mm lock) and keep hundreds of write streams in the TLB
template <typename T> struct HugeBuf {
    T *p = nullptr;
    size_t cap = 0, n = 0;
    HugeBuf() = default;
    HugeBuf(const HugeBuf &) = delete;
    HugeBuf &operator=(const HugeBuf &) = delete;
    ~HugeBuf() { release(); }
    void release() { if (p) munmap(p, bytes()); p = nullptr; cap = n = 0; }
    size_t bytes() const { return ((cap * sizeof(T) + (2u << 20) - 1) / (2u << 20)) * (2u << 20); }
    bool alloc(size_t elems) {
        release();
        cap = elems ? elems : 1;
        void *m = mmap(nullptr, bytes(), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (m == MAP_FAILED) { cap = 0; return false; }
        (void)madvise(m, bytes(), MADV_HUGEPAGE);
        p = (T *)m;
        return true;
    }
    void push(T v) { p[n++] = v; }
};

// The whole file in one anonymous huge-page buffer, read by all threads at once (pread of disjoint ranges: page-cache copies run in
// parallel; faulting a file MAPPING in 4 KiB steps from eight threads measured 5 x slower than the parse itself).  One zero byte follows
// the data -- the terminator strtol / strtod need at the end of the last line.
struct Slurp {
    HugeBuf<char> buf;
    char *p = nullptr;
    size_t n = 0;
};

bool slurp(const char *path, Slurp &s) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return false;
    struct stat st{};
    if (fstat(fd, &st) != 0 || st.st_size < 0) { close(fd); return false; }
    const size_t n = (size_t)st.st_size;
    if (!s.buf.alloc(n + 1)) { close(fd); return false; }
    int nth = 1;
#ifdef _OPENMP
    nth = omp_get_max_threads();
#endif
    if (n < (64u << 20)) nth = 1;
    bool ok = true;
#pragma omp parallel for schedule(static, 1) num_threads(nth)
    for (int t = 0; t < nth; ++t) {
        size_t o = n * (size_t)t / (size_t)nth;
        const size_t o1 = n * (size_t)(t + 1) / (size_t)nth;
        while (o < o1) {
            const ssize_t r = pread(fd, s.buf.p + o, std::min<size_t>(o1 - o, (size_t)64 << 20), (off_t)o);
            if (r <= 0) {
#pragma omp atomic write
                ok = false;
                break;
            }
            o += (size_t)r;
        }
    }
    close(fd);
    if (!ok) return false;
    s.p = s.buf.p; s.n = n;
    s.p[n] = 0;
    return true;
}

// decimal -> double, correctly rounded like the reference's fscanf("%lg") / strtod.  Fast path for plain decimal tokens "[-]ddd[.ddd][e[+-]dd]"
// with at most 19 significant digits: the digits as one 64-bit integer m (exact in x87 extended precision, 64-bit significand), times or
// divided by an exactly representable power of ten 10^k, k <= 27 -- ONE rounding to 64 bits, the true value within half a unit of the
// last place -- then rounded to double.  That second rounding can only differ from a direct one when the 64-bit value sits at a midpoint
// between two doubles (low 11 bits 0x400; 0x3ff and 0x401 are refused as well): those tokens, and everything else (more digits, larger
// exponents, hex floats, '+', inf / nan), go to strtod.  (libstdc++ 11's std::from_chars<double> is strtod behind a locale switch: slower.)
inline bool parse_double(char *&q, char *e, double &v) {
    while (q < e && (*q == ' ' || *q == '\t')) ++q;
    const char *s = q;
    bool neg = false;
    if (s < e && *s == '-') { neg = true; ++s; }
    uint64_t m = 0;
    int nd = 0, frac = 0;
    bool any = false, ok = true;
    while (s < e && *s >= '0' && *s <= '9') { any = true; if (nd < 19) { m = m * 10 + (uint64_t)(*s - '0'); nd += (m != 0); } else ok = false; ++s; }
    if (s < e && *s == '.') {
        ++s;
        while (s < e && *s >= '0' && *s <= '9') {
            any = true;
            if (nd < 19) { m = m * 10 + (uint64_t)(*s - '0'); nd += (m != 0); ++frac; }
            else if (*s != '0') ok = false;                   // (zeros behind the 19th significant digit change nothing)
            ++s;
        }
    }
    int ex = 0;
    if (any && s < e && (*s == 'e' || *s == 'E')) {
        const char *t = s + 1;
        bool eneg = false;
        if (t < e && (*t == '+' || *t == '-')) { eneg = *t == '-'; ++t; }
        if (t < e && *t >= '0' && *t <= '9') {
            int x = 0;
            while (t < e && *t >= '0' && *t <= '9') { if (x < 100000) x = x * 10 + (*t - '0'); ++t; }
            ex = eneg ? -x : x;
            s = t;
        }
    }
    // the token must end here: anything glued to it ('x' of a hex float, a 'd' exponent, "nan(...)") is strtod's business
    if (any && ok && (s >= e || *s == ' ' || *s == '\t' || *s == '\n' || *s == '\r' || *s == 0)) {
        const int k = ex - frac;
        if (m == 0) { v = neg ? -0.0 : 0.0; q = (char *)s; return true; }
        if (k >= -27 && k <= 27) {
            static const long double P10[28] = {1e0L, 1e1L, 1e2L, 1e3L, 1e4L, 1e5L, 1e6L, 1e7L, 1e8L, 1e9L, 1e10L, 1e11L, 1e12L, 1e13L, 1e14L, 1e15L, 1e16L,
                                                1e17L, 1e18L, 1e19L, 1e20L, 1e21L, 1e22L, 1e23L, 1e24L, 1e25L, 1e26L, 1e27L};
            long double x = (long double)m;                   // exact: m < 2^64
            bool ambiguous = false;
            if (k != 0) {
                x = k > 0 ? x * P10[k] : x / P10[-k];         // one rounding to 64 bits
                uint64_t sig;
                memcpy(&sig, &x, 8);                          // x87 extended: the low 8 bytes are the explicit 64-bit significand
                const unsigned low = (unsigned)(sig & 0x7FF);
                ambiguous = low >= 0x3FF && low <= 0x401;
            }
            if (!ambiguous) {
                v = (double)x;                                // (k == 0: the one and only rounding, ties to even like strtod)
                if (neg) v = -v;
                q = (char *)s;
                return true;
            }
        }
    }
    char *r2;
    v = strtod(q, &r2);
    if (r2 == q) return false;
    q = r2;
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

// MatrixMarket writer (no reference counterpart: the reference only reads; tools/mtx_scale_probe.py and the tests write inputs with it).
// symmetric != 0: only entries with column <= row are written under a "symmetric" banner -- the reader expands them again.
// Values are printed with 17 significant digits, so a general file read back gives the same doubles.
int uspmv_coo_write_mtx(const uspmv_coo_t *m, const char *path, int symmetric) {
    if (!m || !path) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_coo_write_mtx: NULL argument");
    FILE *f = fopen(path, "wb");
    if (!f) return uspmv::fail(USPMV_ERR_IO, "uspmv_coo_write_mtx: cannot create '%s'", path);
    int64_t stored = m->nnz;
    if (symmetric) {
        stored = 0;
#pragma omp parallel for reduction(+ : stored) schedule(static)
        for (int64_t k = 0; k < m->nnz; ++k) stored += m->J[(size_t)k] <= m->I[(size_t)k];
    }
    bool ok = fprintf(f, "%%%%MatrixMarket matrix coordinate real %s\n%lld %lld %lld\n", symmetric ? "symmetric" : "general", (long long)m->n_rows,
                      (long long)m->n_cols, (long long)stored) > 0;
    constexpr int64_t BLK = 1 << 22;                        // entries formatted per round (every thread its piece, written in order)
    for (int64_t b0 = 0; b0 < m->nnz && ok; b0 += BLK) {
        const int64_t b1 = std::min(b0 + BLK, m->nnz);
        int nt = 1;
#ifdef _OPENMP
        nt = omp_get_max_threads();
#endif
        std::vector<std::string> piece((size_t)nt);
#pragma omp parallel num_threads(nt)
        {
            int t = 0;
#ifdef _OPENMP
            t = omp_get_thread_num();
#endif
            const int64_t per = (b1 - b0 + nt - 1) / nt, k0 = b0 + t * per, k1 = std::min(k0 + per, b1);
            std::string &out = piece[(size_t)t];
            out.reserve((size_t)std::max<int64_t>(k1 - k0, 0) * 40);
            char line[96];
            for (int64_t k = k0; k < k1; ++k) {
                if (symmetric && m->J[(size_t)k] > m->I[(size_t)k]) continue;
                const int n = snprintf(line, sizeof line, "%d %d %.17g\n", m->I[(size_t)k] + 1, m->J[(size_t)k] + 1, m->values[(size_t)k]);
                out.append(line, (size_t)n);
            }
        }
        for (const std::string &o : piece) ok = ok && (o.empty() || fwrite(o.data(), 1, o.size(), f) == o.size());
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok) return uspmv::fail(USPMV_ERR_IO, "uspmv_coo_write_mtx: write to '%s' failed", path);
    return USPMV_OK;
}

int uspmv_read_mtx(const char *path, uspmv_coo_t **out) {
    if (!path || !out) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_read_mtx: NULL argument");
    const bool verbose = getenv("USPMV_VERBOSE") != nullptr;
    double t_last = omp_get_wtime();
    auto lap = [&](const char *what) { if (verbose) { const double t = omp_get_wtime(); fprintf(stderr, "[uspmv] read_mtx: %-28s %.2f s\n", what, t - t_last); t_last = t; } };
    Slurp s;
    if (!slurp(path, s)) return uspmv::fail(USPMV_ERR_IO, "uspmv_read_mtx: cannot open '%s'", path);
    lap("map the file");
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
        std::vector<HugeBuf<int32_t>> pr((size_t)nth), pc((size_t)nth);
        std::vector<HugeBuf<double>> pv((size_t)nth);
        std::vector<long> n_parsed((size_t)nth, 0);
        std::vector<int> err((size_t)nth, 0);
#pragma omp parallel for schedule(static, 1) num_threads(nth)
        for (int t = 0; t < nth; ++t) {
            char *q = cut[(size_t)t], *e = cut[(size_t)t + 1];
            auto &R = pr[(size_t)t]; auto &Cc = pc[(size_t)t]; auto &V = pv[(size_t)t];
            // upper bound of the entries of this piece: the shortest line is "1 1\n" (pattern) / "1 1 1\n" (only touched pages become real)
            const size_t bound = ((size_t)(e - q) / (pattern ? 4 : 6) + 2) * (symmetric ? 2 : 1);
            if (!R.alloc(bound) || !Cc.alloc(bound) || !V.alloc(bound)) { err[(size_t)t] = 3; continue; }
            // (everything the loop updates lives in locals: the per-thread counters of neighbouring threads share cache lines)
            int32_t *rp = R.p, *cp = Cc.p;
            double *vp = V.p;
            size_t w = 0;
            long lines = 0;
            int bad = 0;
            while (q < e) {
                while (q < e && (*q == ' ' || *q == '\t' || *q == '\r' || *q == '\n')) ++q;
                if (q >= e) break;
                char *r2;
                long r = strtol(q, &r2, 10);
                if (r2 == q) { bad = 1; break; }
                q = r2;
                long c = strtol(q, &r2, 10);
                if (r2 == q) { bad = 1; break; }
                q = r2;
                double v = 0.01;  // pattern matrices (code/mmio.h:195-203)
                if (!pattern && !parse_double(q, e, v)) { bad = 1; break; }
                if (r < 1 || r > M || c < 1 || c > N) { bad = 2; break; }
                rp[w] = (int32_t)(r - 1); cp[w] = (int32_t)(c - 1); vp[w] = v; ++w;
                if (symmetric && r != c) { rp[w] = (int32_t)(c - 1); cp[w] = (int32_t)(r - 1); vp[w] = v; ++w; }
                ++lines;
            }
            R.n = Cc.n = V.n = w;
            n_parsed[(size_t)t] = lines;
            err[(size_t)t] = bad;
        }
        lap("parse");
        long total = 0;
        for (int t = 0; t < nth; ++t) {
            if (err[(size_t)t] == 3) return uspmv::fail(USPMV_ERR_ALLOC, "uspmv_read_mtx: out of memory");
            if (err[(size_t)t] == 2) return uspmv::fail(USPMV_ERR_IO, "uspmv_read_mtx: an entry lies outside the %ldx%ld matrix", M, N);
            if (err[(size_t)t]) return uspmv::fail(USPMV_ERR_IO, "uspmv_read_mtx: malformed entry line");
            total += n_parsed[(size_t)t];
        }
        if (total < NZ) return uspmv::fail(USPMV_ERR_IO, "uspmv_read_mtx: premature end of file (%ld of %ld entries)", total, NZ);
        // keep exactly the first NZ entries of the file (what the reference's counted loop reads): per piece, how many of its (expanded)
        // elements count.  A file with more lines than the header announces is cut inside the piece that crosses NZ.
        std::vector<size_t> keep((size_t)nth, 0);
        long seen = 0;
        for (int t = 0; t < nth; ++t) {
            const int32_t *R = pr[(size_t)t].p, *Cc = pc[(size_t)t].p;
            const size_t rn = pr[(size_t)t].n;
            if (seen + n_parsed[(size_t)t] <= NZ) { keep[(size_t)t] = rn; seen += n_parsed[(size_t)t]; continue; }
            size_t k = 0;
            while (seen < NZ && k < rn) {
                const bool pair = symmetric && k + 1 < rn && R[k] == Cc[k + 1] && Cc[k] == R[k + 1] && R[k] != Cc[k];
                k += pair ? 2 : 1;
                ++seen;
            }
            keep[(size_t)t] = k;
            break;
        }
        size_t tot = 0;
        for (int t = 0; t < nth; ++t) tot += keep[(size_t)t];
        if (tot > (size_t)INT32_MAX)
            return uspmv::fail(USPMV_ERR_OVERFLOW, "uspmv_read_mtx: expanded nnz exceeds the 32-bit index type");

        // ---- stable sort by row (code/utilities.hpp:2139-2146, :2278), in parallel and in two passes instead of one serial counting sort
        //      with 2e8 cache-missing scatters: (1) the pieces, in file order, are distributed stably over NB buckets of consecutive rows
        //      (per-piece bucket histograms -> every piece knows where its share of a bucket starts: sequential writes into NB streams);
        //      (2) every bucket -- a few thousand rows, cache-resident -- is counting-sorted by row on its own, stably, into its final place.
        const int64_t nnz = (int64_t)tot;
        int shift = 0;
        while (((int64_t)(M - 1) >> shift) >= 512) ++shift;      // (<= 512 buckets: 3 x 512 write streams stay inside the second-level TLB)
        const int64_t NB = ((int64_t)(M - 1) >> shift) + 1;
        std::vector<std::vector<int64_t>> hist((size_t)nth, std::vector<int64_t>((size_t)NB, 0));
#pragma omp parallel for schedule(static, 1) num_threads(nth)
        for (int t = 0; t < nth; ++t) {
            const int32_t *R = pr[(size_t)t].p;
            int64_t *h = hist[(size_t)t].data();
            for (size_t k = 0; k < keep[(size_t)t]; ++k) h[R[k] >> shift]++;
        }
        lap("bucket histograms");
        std::vector<int64_t> bstart((size_t)NB + 1, 0);
        {
            int64_t run = 0;
            for (int64_t b = 0; b < NB; ++b) {
                bstart[(size_t)b] = run;
                for (int t = 0; t < nth; ++t) { const int64_t c = hist[(size_t)t][(size_t)b]; hist[(size_t)t][(size_t)b] = run; run += c; }
            }
            bstart[(size_t)NB] = run;
        }
        HugeBuf<int32_t> trb, tcb;
        HugeBuf<double> tvb;
        if (!trb.alloc((size_t)nnz) || !tcb.alloc((size_t)nnz) || !tvb.alloc((size_t)nnz)) return uspmv::fail(USPMV_ERR_ALLOC, "uspmv_read_mtx: out of memory");
        int32_t *tr = trb.p, *tc = tcb.p;
        double *tv = tvb.p;
#pragma omp parallel for schedule(static, 1) num_threads(nth)
        for (int t = 0; t < nth; ++t) {
            const int32_t *R = pr[(size_t)t].p, *Cc = pc[(size_t)t].p;
            const double *V = pv[(size_t)t].p;
            int64_t *h = hist[(size_t)t].data();
            for (size_t k = 0; k < keep[(size_t)t]; ++k) {
                const int64_t d = h[R[k] >> shift]++;
                tr[(size_t)d] = R[k]; tc[(size_t)d] = Cc[k]; tv[(size_t)d] = V[k];
            }
            pr[(size_t)t].release(); pc[(size_t)t].release(); pv[(size_t)t].release();
        }
        lap("distribute over buckets");
        auto *m = new uspmv_coo;
        m->n_rows = M; m->n_cols = N; m->nnz = nnz;
        m->I.resize((size_t)nnz); m->J.resize((size_t)nnz); m->values.resize((size_t)nnz);
        lap("allocate the result");
#pragma omp parallel
        {
            std::vector<int64_t> cnt((size_t)1 << shift);
#pragma omp for schedule(dynamic, 8)
            for (int64_t b = 0; b < NB; ++b) {
                const int64_t k0 = bstart[(size_t)b], k1 = bstart[(size_t)b + 1];
                const int32_t r0 = (int32_t)(b << shift);
                const int64_t nr = std::min<int64_t>((int64_t)1 << shift, M - r0);
                std::fill(cnt.begin(), cnt.begin() + nr, 0);
                for (int64_t k = k0; k < k1; ++k) cnt[(size_t)(tr[(size_t)k] - r0)]++;
                int64_t run = k0;
                for (int64_t r = 0; r < nr; ++r) { const int64_t c = cnt[(size_t)r]; cnt[(size_t)r] = run; run += c; }
                for (int64_t k = k0; k < k1; ++k) {
                    const int64_t d = cnt[(size_t)(tr[(size_t)k] - r0)]++;
                    m->I[(size_t)d] = tr[(size_t)k]; m->J[(size_t)d] = tc[(size_t)k]; m->values[(size_t)d] = tv[(size_t)k];
                }
            }
        }
        lap("sort inside the buckets");
        *out = m;
        return USPMV_OK;
    }
}

}  // extern "C"
