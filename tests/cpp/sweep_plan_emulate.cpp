// Host-side check of the column-window sweep plan (ultimate-spmv_amd/host/sweep_plan.cpp): replays the plan exactly the way
// scs_spmv_sweep consumes it (windows ascending, rounds, active lanes ascending, compacted stream, trailing padding once)
// and compares y bit for bit with the plain slot-ordered FMA chain over the SCS arrays (the reference's summation,
// code/kernels.hpp:237-252 / code/ap_kernels.hpp:59-75).  Test infrastructure only; built by tests/test_sweep_plan.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "uspmv_internal.hpp"

static int fails = 0;
#define REQUIRE(c) do { if (!(c)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

template <typename VT>
static std::vector<VT> chain(const uspmv_scs *s, const std::vector<VT> &x, const VT *vals) {
    std::vector<VT> y((size_t)(s->n_chunks * s->C));
    for (int64_t c = 0; c < s->n_chunks; ++c)
        for (int64_t i = 0; i < s->C; ++i) {
            VT acc = 0;
            for (int64_t j = 0; j < s->chunk_lengths[c]; ++j) {
                const int64_t k = s->chunk_ptrs[c] + j * s->C + i;
                acc = std::fma(vals[k], x[(size_t)s->col_idxs[k]], acc);
            }
            y[(size_t)(c * s->C + i)] = acc;
        }
    return y;
}

static bool same_bits(double a, double b) { return std::memcmp(&a, &b, 8) == 0 || (std::isnan(a) && std::isnan(b)); }

static void run(int64_t n, int nnz_row, int64_t band, int C, int sigma, int wlog, int tile_rows, bool ap, bool special) {
    uspmv_coo_t *coo = nullptr;
    REQUIRE(uspmv_gen_banded_random(n, nnz_row, band, 7, ap ? 10.0 : 0.0, 0, n, &coo) == 0);
    uspmv_coo_t *dpc = coo, *spc = nullptr;
    if (ap) REQUIRE(uspmv_partition_precisions(coo, 1e-3, &dpc, &spc) == 0);
    uspmv_scs_t *s = nullptr, *s2 = nullptr;
    REQUIRE(uspmv_convert_to_scs(dpc, C, sigma, USPMV_F64, nullptr, &s) == 0);
    REQUIRE(uspmv_permute_scs_cols(s, s->old_to_new_idx.data()) == 0);
    if (ap) {
        REQUIRE(uspmv_convert_to_scs(spc, C, sigma, USPMV_F32, s->old_to_new_idx.data(), &s2) == 0);
        REQUIRE(uspmv_permute_scs_cols(s2, s->old_to_new_idx.data()) == 0);
    }
    std::vector<double> x((size_t)std::max<int64_t>(s->n_rows_padded, n));
    for (size_t i = 0; i < x.size(); ++i) x[i] = 1.0 + 1e-3 * (double)(i % 1000);
    if (special) { x[0] = -INFINITY; x[5] = -0.0; x[17] = NAN; }
    uspmv_sweep_plan p;
    REQUIRE(uspmv_build_sweep_plan(s, s2, wlog, tile_rows, 1e9, &p) == 0);
    REQUIRE(p.valid);
    const int64_t R = p.tile_rows, wpt = R / 64, n_pad = s->n_chunks * s->C;
    std::vector<double> y((size_t)n_pad, 12345.0);
    std::vector<char> covered((size_t)n_pad, 0);
    for (int64_t k = 0; k < p.n_sweep_tiles; ++k) {
        const int64_t t = p.tile_ids[(size_t)k];
        for (int64_t v = 0; v < wpt; ++v) {
            double acc[64], accb[64];
            for (int l = 0; l < 64; ++l) acc[l] = accb[l] = 0.0;
            uint32_t base = p.wave_off[(size_t)(k * wpt + v)], base_b = ap ? p.wave_off_b[(size_t)(k * wpt + v)] : 0;
            for (int64_t sw = 0; sw < p.t_S[(size_t)k]; ++sw) {
                const int64_t g0 = (int64_t)(p.t_smin[(size_t)k] + sw) << wlog;
                for (int part = 0; part < (ap ? 2 : 1); ++part) {
                    const uint8_t *cnt = (part ? p.cnt_b.data() : p.cnt.data()) + p.t_cnt_off[(size_t)k] + sw * R + v * 64;
                    for (int kk = 0;; ++kk) {
                        bool any = false;
                        for (int l = 0; l < 64; ++l) {
                            if (cnt[l] <= kk) continue;
                            any = true;
                            if (part == 0) { acc[l] = std::fma(p.vals_f64[base], x[(size_t)(g0 + p.idx[base])], acc[l]); ++base; }
                            else { accb[l] = std::fma((double)p.vals_b_f32[base_b], x[(size_t)(g0 + p.idx_b[base_b])], accb[l]); ++base_b; }
                        }
                        if (!any) break;
                    }
                }
            }
            for (int l = 0; l < 64; ++l) {
                const int64_t row = t * R + v * 64 + l;
                if (row >= n_pad) continue;
                const int32_t pc = p.pad_col[(size_t)(k * R + v * 64 + l)];
                if (pc >= 0) acc[l] = std::fma(0.0, x[(size_t)pc], acc[l]);
                if (ap) {
                    const int32_t pcb = p.pad_col_b[(size_t)(k * R + v * 64 + l)];
                    if (pcb >= 0) accb[l] = std::fma((double)0.0f, x[(size_t)pcb], accb[l]);
                    acc[l] += accb[l];
                }
                y[(size_t)row] = acc[l]; covered[(size_t)row] = 1;
            }
        }
    }
    // rows of the chunks the plan leaves to the gather kernel
    for (int32_t c : p.rest_chunks) for (int64_t i = 0; i < s->C; ++i) covered[(size_t)(c * s->C + i)] = 2;
    std::vector<double> ref = chain<double>(s, x, s->values_f64.data());
    if (ap) {
        std::vector<double> refb((size_t)n_pad);
        for (int64_t c = 0; c < s2->n_chunks; ++c)
            for (int64_t i = 0; i < C; ++i) {
                double acc = 0;
                for (int64_t j = 0; j < s2->chunk_lengths[c]; ++j) {
                    const int64_t q = s2->chunk_ptrs[c] + j * C + i;
                    acc = std::fma((double)s2->values_f32[q], x[(size_t)s2->col_idxs[q]], acc);
                }
                refb[(size_t)(c * C + i)] = acc;
            }
        for (int64_t r = 0; r < n_pad; ++r) ref[(size_t)r] += refb[(size_t)r];
    }
    int64_t bad = 0, nsw = 0;
    for (int64_t r = 0; r < n_pad; ++r) {
        REQUIRE(covered[(size_t)r] != 0);
        if (covered[(size_t)r] == 1) { ++nsw; if (!same_bits(y[(size_t)r], ref[(size_t)r])) ++bad; }
    }
    printf("n=%ld C=%d sigma=%d wlog=%d tile=%d ap=%d special=%d: sweep tiles %ld/%ld, rows checked %ld, mismatches %ld\n", (long)n, C, sigma, wlog,
           tile_rows, (int)ap, (int)special, (long)p.n_sweep_tiles, (long)p.n_tiles, (long)nsw, (long)bad);
    REQUIRE(bad == 0);
    REQUIRE(nsw > 0);
    uspmv_scs_free(s); if (s2) uspmv_scs_free(s2);
    if (ap) { uspmv_coo_free(dpc); uspmv_coo_free(spc); }
    uspmv_coo_free(coo);
}

int main() {
    run(20000, 40, 3000, 32, 512, 9, 256, false, false);
    run(20000, 40, 3000, 32, 512, 10, 1024, false, true);
    run(9999, 23, 700, 16, 64, 8, 512, false, false);
    run(5000, 300, 2400, 32, 1, 11, 256, false, false);   // > 255 entries per window possible: such tiles must not sweep
    run(20000, 40, 3000, 32, 512, 9, 1024, true, false);
    run(20000, 40, 3000, 32, 512, 10, 2048, true, false);   // several rows per lane in the kernel: tiles above 1 024 rows
    run(9999, 23, 700, 16, 64, 9, 4096, false, true);
    run(7777, 31, 900, 64, 128, 8, 256, true, true);
    printf(fails ? "FAILED\n" : "OK\n");
    return fails ? 1 : 0;
}
