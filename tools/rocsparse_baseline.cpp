// Vendor baseline on the same matrix (measurement tool, not product code): rocSPARSE CSR (default,
// adaptive) and sliced-ELL SpMV -- the ROCm twin of the reference's optional cuSPARSE path
// (USE_CUSPARSE: cusparseCreateCsr / cusparseCreateSlicedEll + cusparseSpMV, code/utilities.hpp:3380-3550,
// code/classes_structs.hpp:998-1011).  Matrix and SELL-C-sigma arrays come from libuspmv's host layer.
//   hipcc -O2 -std=c++17 -Iinclude tools/rocsparse_baseline.cpp -o tools/rocsparse_baseline \
//         -Lultimate-spmv_amd -luspmv -lrocsparse -Wl,-rpath,$PWD/ultimate-spmv_amd
#include <hip/hip_runtime.h>
#include <rocsparse/rocsparse.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "uspmv.h"
#pragma clang diagnostic ignored "-Wdeprecated-declarations"
#define HK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define RK(x) do { rocsparse_status s = (x); if (s != rocsparse_status_success) { printf("%s: status %d\n", #x, (int)s); return 1; } } while (0)
#define UK(x) do { int r = (x); if (r) { printf("%s: %s\n", #x, uspmv_last_error()); return 1; } } while (0)

template <typename T> T *to_dev(const void *h, size_t n) {
    void *d = nullptr;
    if (hipMalloc(&d, n * sizeof(T) + 16) != hipSuccess) return nullptr;
    hipMemcpy(d, h, n * sizeof(T), hipMemcpyHostToDevice);
    return (T *)d;
}

#include <string>
static std::string g_json;   // one {"name": ms, ...} object, printed last when --json is given (bench.py's "vendor_baseline")

int main(int argc, char **argv) {
    const long g = argc > 1 ? atol(argv[1]) : 253;
    const int dof = argc > 2 ? atoi(argv[2]) : 1;
    bool json = false;
    for (int i = 1; i < argc; ++i) if (std::string(argv[i]) == "--json") json = true;
    uspmv_coo_t *coo;
    UK(uspmv_gen_stencil27(g, g, g, dof, 0x5EED, 0.0, 0, g * g * g * dof, &coo));
    int64_t n, nc, nnz;
    uspmv_coo_dims(coo, &n, &nc, &nnz);
    rocsparse_handle h;
    RK(rocsparse_create_handle(&h));
    std::vector<double> hx((size_t)n + 512, 5.0);
    double *dx = to_dev<double>(hx.data(), hx.size()), *dy = to_dev<double>(hx.data(), hx.size());
    rocsparse_dnvec_descr vx, vy;
    const double alpha = 1.0, beta = 0.0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // every algorithm must reproduce the row sums of the first one (x is constant, so y does not depend on the row order): a routine
    // that returns success without computing (seen with csr lrb here) is reported as such and left out of the JSON
    double ref_sum = 0, ref_abs = 0;
    bool have_ref = false;
    std::vector<double> hy;
    auto check_y = [&](const char *name, long rows) -> bool {
        hy.resize((size_t)rows);
        if (hipMemcpy(hy.data(), dy, sizeof(double) * (size_t)rows, hipMemcpyDeviceToHost) != hipSuccess) return false;
        double sum = 0, sabs = 0;
        for (double v : hy) { sum += v; sabs += v < 0 ? -v : v; }
        if (!have_ref) { ref_sum = sum; ref_abs = sabs; have_ref = true; return true; }
        const double d1 = sum - ref_sum, d2 = sabs - ref_abs;
        if ((d1 < 0 ? -d1 : d1) > 1e-9 * ref_abs || (d2 < 0 ? -d2 : d2) > 1e-9 * ref_abs) {
            printf("%-28s WRONG RESULT (sum %.6e vs %.6e): not reported\n", name, sum, ref_sum);
            return false;
        }
        return true;
    };
    auto run = [&](const char *name, rocsparse_spmat_descr A, rocsparse_spmv_alg alg, long rows) -> int {
        RK(rocsparse_create_dnvec_descr(&vx, rows, dx, rocsparse_datatype_f64_r));      // (square: the sliced-ELL struct is padded to a multiple of 32)
        RK(rocsparse_create_dnvec_descr(&vy, rows, dy, rocsparse_datatype_f64_r));
        size_t bs = 0;
        rocsparse_status st = rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, vx, &beta, vy, rocsparse_datatype_f64_r, alg,
                                             rocsparse_spmv_stage_buffer_size, &bs, nullptr);
        if (st != rocsparse_status_success) { printf("%-28s not supported (status %d)\n", name, (int)st); return 0; }
        void *buf = nullptr;
        HK(hipMalloc(&buf, bs + 16));
        RK(rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, vx, &beta, vy, rocsparse_datatype_f64_r, alg, rocsparse_spmv_stage_preprocess, &bs, buf));
        HK(hipMemset(dy, 0, sizeof(double) * (size_t)rows));
        for (int k = 0; k < 5; ++k)
            RK(rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, vx, &beta, vy, rocsparse_datatype_f64_r, alg, rocsparse_spmv_stage_compute, &bs, buf));
        hipEventRecord(e0);
        const int reps = 50;
        for (int k = 0; k < reps; ++k)
            RK(rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, vx, &beta, vy, rocsparse_datatype_f64_r, alg, rocsparse_spmv_stage_compute, &bs, buf));
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        if (!check_y(name, rows)) { hipFree(buf); return 0; }
        printf("%-28s %8.4f ms  %8.1f GF/s  %7.0f GB/s (12 B/nnz + vectors)\n", name, ms, 2.0 * nnz / ms / 1e6, (12.0 * nnz + 16.0 * n) / ms / 1e6);
        char jb[128];
        snprintf(jb, sizeof jb, "%s\"%s\": %.5f", g_json.empty() ? "" : ", ", name, ms);
        g_json += jb;
        fflush(stdout);
        hipFree(buf);
        return 0;
    };
    {   // CSR = SELL-1-1
        uspmv_scs_t *s;
        UK(uspmv_convert_to_scs(coo, 1, 1, USPMV_F64, nullptr, &s));
        const int32_t *rp, *ci; const void *va;
        uspmv_scs_arrays(s, &rp, nullptr, &ci, &va, nullptr, nullptr);
        int32_t *drp = to_dev<int32_t>(rp, (size_t)n + 1), *dci = to_dev<int32_t>(ci, (size_t)nnz);
        double *dva = to_dev<double>(va, (size_t)nnz);
        rocsparse_spmat_descr A;
        RK(rocsparse_create_csr_descr(&A, n, n, nnz, drp, dci, dva, rocsparse_indextype_i32, rocsparse_indextype_i32, rocsparse_index_base_zero, rocsparse_datatype_f64_r));
        if (run("rocsparse csr default", A, rocsparse_spmv_alg_default, n)) return 1;
        if (run("rocsparse csr adaptive", A, rocsparse_spmv_alg_csr_adaptive, n)) return 1;
        if (run("rocsparse csr rowsplit", A, rocsparse_spmv_alg_csr_rowsplit, n)) return 1;
        if (run("rocsparse csr lrb", A, rocsparse_spmv_alg_csr_lrb, n)) return 1;
        rocsparse_destroy_spmat_descr(A);
        hipFree(drp); hipFree(dci); hipFree(dva); uspmv_scs_free(s);
    }
    for (int variant = 0; variant < 5; ++variant) {   // sliced ELL, slice = 32, from the SELL-32-512 struct (rows in sigma-sorted order; x is constant)
        // the descriptor's conventions are not documented beyond the argument list, so the forms it may expect are tried in turn:
        // 0: nnz = true non-zeros, padding as the reference stores it (value 0, column 0)   1: nnz = size of the col / val arrays
        // 2: as 1 with padding columns = -1 (the convention of rocSPARSE's ELL format)   3 / 4: as 0 / 2 with rows = cols = the TRUE row count
        //    (the last slice then holds fewer than 32 rows)
        uspmv_scs_t *s;
        UK(uspmv_convert_to_scs(coo, 32, 512, USPMV_F64, nullptr, &s));
        int64_t meta[8]; uspmv_scs_meta(s, meta);
        const int32_t *cp, *ci, *o2n; const void *va;
        uspmv_scs_arrays(s, &cp, nullptr, &ci, &va, &o2n, nullptr);
        uspmv_permute_scs_cols(s, o2n);
        uspmv_scs_arrays(s, &cp, nullptr, &ci, &va, nullptr, nullptr);
        std::vector<int32_t> cim(ci, ci + meta[6]);
        if (variant == 2 || variant == 4) { const double *vv = (const double *)va; for (int64_t k = 0; k < meta[6]; ++k) if (vv[k] == 0.0 && cim[(size_t)k] == 0) cim[(size_t)k] = -1; }
        int32_t *dcp = to_dev<int32_t>(cp, (size_t)meta[5] + 1), *dci = to_dev<int32_t>(cim.data(), (size_t)meta[6]);
        double *dva = to_dev<double>(va, (size_t)meta[6]);
        rocsparse_spmat_descr A;
        rocsparse_status st = rocsparse_create_sell_descr(&A, variant >= 3 ? n : meta[4], variant >= 3 ? n : meta[4], (variant == 0 || variant == 3) ? nnz : meta[6], 32, meta[6], dcp, dci, dva, rocsparse_indextype_i32,
                                                          rocsparse_indextype_i32, rocsparse_index_base_zero, rocsparse_datatype_f64_r);
        bool done = false;
        if (st != rocsparse_status_success) printf("rocsparse sliced-ELL descriptor (form %d): status %d\n", variant, (int)st);
        else {   // sliced ELL is served by the descriptor-based rocsparse_v2_spmv only (the staged rocsparse_spmv answers not_implemented)
            const char *name = "rocsparse sliced-ELL (32)";
            const long rows = variant >= 3 ? n : meta[4];
            rocsparse_spmv_descr sd;
            RK(rocsparse_create_spmv_descr(&sd));
            const rocsparse_spmv_alg alg = rocsparse_spmv_alg_sell;
            const rocsparse_operation op = rocsparse_operation_none;
            const rocsparse_datatype dt = rocsparse_datatype_f64_r;
            RK(rocsparse_spmv_set_input(h, sd, rocsparse_spmv_input_alg, &alg, sizeof alg, nullptr));
            RK(rocsparse_spmv_set_input(h, sd, rocsparse_spmv_input_operation, &op, sizeof op, nullptr));
            RK(rocsparse_spmv_set_input(h, sd, rocsparse_spmv_input_scalar_datatype, &dt, sizeof dt, nullptr));
            RK(rocsparse_spmv_set_input(h, sd, rocsparse_spmv_input_compute_datatype, &dt, sizeof dt, nullptr));
            RK(rocsparse_create_dnvec_descr(&vx, rows, dx, rocsparse_datatype_f64_r));
            RK(rocsparse_create_dnvec_descr(&vy, rows, dy, rocsparse_datatype_f64_r));
            size_t bs = 0;
            rocsparse_status s2 = rocsparse_v2_spmv_buffer_size(h, sd, A, vx, vy, rocsparse_v2_spmv_stage_analysis, &bs, nullptr);
            void *buf = nullptr;
            if (s2 == rocsparse_status_success) { HK(hipMalloc(&buf, bs + 16)); s2 = rocsparse_v2_spmv(h, sd, &alpha, A, vx, &beta, vy, rocsparse_v2_spmv_stage_analysis, bs, buf, nullptr); }
            size_t bc = 0;
            if (s2 == rocsparse_status_success) s2 = rocsparse_v2_spmv_buffer_size(h, sd, A, vx, vy, rocsparse_v2_spmv_stage_compute, &bc, nullptr);
            void *bufc = nullptr;
            if (s2 == rocsparse_status_success) { HK(hipMalloc(&bufc, bc + 16)); HK(hipMemset(dy, 0, sizeof(double) * (size_t)rows)); s2 = rocsparse_v2_spmv(h, sd, &alpha, A, vx, &beta, vy, rocsparse_v2_spmv_stage_compute, bc, bufc, nullptr); }
            if (s2 != rocsparse_status_success) printf("%-28s form %d: status %d\n", name, variant, (int)s2);
            else {
                for (int k = 0; k < 5; ++k) RK(rocsparse_v2_spmv(h, sd, &alpha, A, vx, &beta, vy, rocsparse_v2_spmv_stage_compute, bc, bufc, nullptr));
                hipEventRecord(e0);
                const int reps = 50;
                for (int k = 0; k < reps; ++k) RK(rocsparse_v2_spmv(h, sd, &alpha, A, vx, &beta, vy, rocsparse_v2_spmv_stage_compute, bc, bufc, nullptr));
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
                if (check_y(name, n)) {
                    printf("%-28s %8.4f ms  %8.1f GF/s  %7.0f GB/s (12 B/nnz + vectors)   [descriptor form %d]\n", name, ms, 2.0 * nnz / ms / 1e6, (12.0 * nnz + 16.0 * n) / ms / 1e6, variant);
                    char jb[128];
                    snprintf(jb, sizeof jb, "%s\"%s\": %.5f", g_json.empty() ? "" : ", ", name, ms);
                    g_json += jb;
                    done = true;
                }
            }
            if (buf) hipFree(buf);
            if (bufc) hipFree(bufc);
        }
        hipFree(dcp); hipFree(dci); hipFree(dva); uspmv_scs_free(s);
        if (done) break;
    }
    const bool sell_ran = g_json.find("sliced-ELL") != std::string::npos;
    if (json) {
        int ver = 0;
        rocsparse_get_version(h, &ver);
        printf("{\"library\": \"rocSPARSE %d.%d.%d\", \"n\": %ld, \"nnz\": %ld, \"ms\": {%s}, \"sliced_ell\": \"%s\"}\n", ver / 100000, ver / 100 % 1000, ver % 100, (long)n, (long)nnz,
               g_json.c_str(), sell_ran ? "ran" : "rocsparse_create_sell_descr succeeds, but rocsparse_v2_spmv answers invalid_size (and the staged rocsparse_spmv not_implemented) for every descriptor form tried: no sliced-ELL number from this rocSPARSE");
    }
    return 0;
}
