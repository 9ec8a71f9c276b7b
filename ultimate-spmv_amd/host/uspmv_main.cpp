// `uspmv` -- command-line harness with the reference's interface:
//     uspmv <matrix>.mtx <crs|csr|scs> [options]
// Flag names, defaults and cross-option checks follow parse_cli_inputs (reference
// code/utilities.hpp:1047-1545, README.md:12-35); the benchmark protocol follows bench_spmv
// (code/main.cpp:380-527): 100 warm-up SpMVs, then batches of n_iter = 2,4,8,... until one batch
// lasts >= -bench_time; GF/s = 2*nnz*block_vec_size / t_iter / 1e9; the result block is appended
// to spmv_bench.txt in the format of write_bench_to_file (code/write_results.hpp:42-157).
// Everything runs through the C ABI of libuspmv.so; there is no CPU kernel in this binary.
//
// Differences, all deliberate: errors return a non-zero exit status (the reference exits 0 on
// CUDA errors, code/classes_structs.hpp:33-41); -block_vec_size > 1 works on the GPU (the reference
// refuses it, code/utilities.hpp:1395-1402); `-mode s` cross-checks the selected kernel against the
// CRS kernel on the device instead of MKL (which the reference needs, code/utilities.hpp:1404-1411);
// achieved HBM GB/s and the roofline fraction are reported next to GF/s.  `<matrix>` may also be
// `gen:NXxNYxNZ[:dof[:decades]]` to use the built-in 27-point-stencil generator.
#include <hip/hip_runtime_api.h>
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <random>
#include <string>
#include <vector>

#include "uspmv.h"
#include "uspmv_dist.hpp"

#define WARM_UP_REPS 100  // code/main.cpp:22
#define HBM_PEAK_GBS 8000.0

namespace {

struct Config {  // subset of the reference's Config (code/classes_structs.hpp:47-153), same defaults
    long chunk_size = 1, sigma = 1;
    char random_init_x = '0';
    unsigned long n_repetitions = 1;
    int validate_result = 1, verbose = 0, block_vec_size = 1, comm_halos = 0, ba_synch = 1, par_pack = 0, no_pack = 0;
    int print_comm_vol = 0, equilibrate = 0, dropout = 0;
    char mode = 'b';
    double bench_time = 5.0, ap_threshold_1 = 0.0, ap_threshold_2 = 0.0, dropout_threshold = 0.0;
    std::string matrix_file_name, seg_method = "seg-rows", value_type = "dp", kernel_format = "scs";
    std::string output_filename_bench = "spmv_bench.txt";
    std::string dump_y;          // -dump_y <file>: solve mode writes its result vector (original row order, raw VT; block vectors column after column) there
    int layout = USPMV_COLWISE;  // run-time here; a make knob in the reference (Makefile:26-31)
    int tlc = 1;                 // build the tile-local-column plan (MI355X-specific, results unchanged)
    int use_graph = 1;           // bench loop replays hipGraphs of 64 launches
    int vec_mode = USPMV_BULKVEC;  // -mpi_mode: message pattern of the block-vector halo exchange (a make knob in the reference, Makefile / config.mk)
    std::string step_form = "auto";
    int bench_steps = 0, bench_warmup = -1, check_y = 0;   // multi-rank: -bench_steps K / -bench_warmup W (fixed-count protocol), -check_y 1
    std::string json;                                       // -json <file|->
    std::string convert = "host";                           // -convert host|device|device_stable: where convert_to_scs runs (device: from the COO arrays in HBM)
    int x_prepared = 1;                                     // -x_prepared 0|1: bench mode, column-major block vectors: re-lay X out once (1) or per call (0)
    std::string part_file;                                  // -seg_metis: part ids from this file (one per row, gpmetis format) instead of the built-in partitioner
};

[[noreturn]] void die(const std::string &msg) {
    fprintf(stderr, "ERROR: %s\n", msg.c_str());
    exit(1);
}
void ck(int rc, const char *what) {
    if (rc != USPMV_OK) die(std::string(what) + ": " + uspmv_last_error());
}
void hk(hipError_t e, const char *what) {
    if (e != hipSuccess) die(std::string(what) + ": " + hipGetErrorString(e));
}

void usage() {
    fprintf(stderr,
            "Usage: uspmv <matrix>.mtx <crs|scs> [options]\n"
            "  -c <int> -s <int> -block_vec_size <int> -rev <int> -rand_x <0|1|m> -dp|-sp|-ap[dp_sp]\n"
            "  -seg_rows|-seg_nnz -validate <0|1> -verbose <0|1> -mode <s|b> -bench_time <float>\n"
            "  -ba_synch <0|1> -comm_halos <0|1> -par_pack <0|1> -no_pack <0|1> -print_comm_vol <0|1>\n"
            "  -equilibrate <0|1> -ap_threshold_1 <float> -ap_threshold_2 <float> -dropout <0|1>\n"
            "  -dropout_threshold <float> -block_vec_layout <colwise|rowwise> -tlc <0|1> -graph <0|1> -dump_y <file>\n"
            "  -mpi_mode <singlevec|multivec|bulkvec>\n"
            "  -bench_steps <int> -bench_warmup <int> -json <file|-> (fixed-count protocol, JSON report)  -x_prepared <0|1> (bench mode, column-major block vectors: X re-laid out once)\n"
            "  -convert <host|device|device_stable> (where convert_to_scs runs; device_stable: ties of the sigma sort in original order)\n"
            "  -tune <key>=<int> (a tuning key of the library, e.g. tlc_idx12=0, spmmv_reorder=1; repeatable)\n"
            "  multi-rank runs: -check_y <0|1> -step_form <auto|auto_all|overlap|plain|pad|fused>\n"
            "  -seg_metis [-part_file <file>]: graph partition (built-in level-set partitioner, or part ids from a gpmetis-style file)\n");
}

Config parse(int argc, char **argv) {
    if (argc < 3) { usage(); exit(1); }
    Config c;
    c.matrix_file_name = argv[1];
    c.kernel_format = argv[2];
    auto need = [&](int &i) -> const char * { if (i + 1 >= argc) die(std::string("missing value after ") + argv[i]); return argv[++i]; };
    for (int i = 3; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "-c") { c.chunk_size = atol(need(i)); if (c.chunk_size < 1) die("chunk size must be >= 1."); }
        else if (a == "-s") { c.sigma = atol(need(i)); if (c.sigma < 1) die("sigma must be >= 1."); }
        else if (a == "-block_vec_size") { c.block_vec_size = atoi(need(i)); if (c.block_vec_size < 1) die("block vector size must be >= 1."); }
        else if (a == "-bench_time" || a == "-bench-time") { c.bench_time = atof(need(i)); if (c.bench_time < 0) die("bench time must be > 0."); }
        else if (a == "-rev") { long v = atol(need(i)); if (v < 1) die("revisions must be >= 1."); c.n_repetitions = (unsigned long)v; }
        else if (a == "-verbose") { c.verbose = atoi(need(i)); if (c.verbose != 0 && c.verbose != 1) die("Only validation verbosity levels 0 and 1 are supported."); }
        else if (a == "-validate") { c.validate_result = atoi(need(i)); if (c.validate_result != 0 && c.validate_result != 1) die("You can only choose to validate result (1, i.e. yes) or not (0, i.e. no)."); }
        else if (a == "-mode") { c.mode = need(i)[0]; if (c.mode != 'b' && c.mode != 's') die("Only bench (b) and solve (s) modes are supported."); }
        else if (a == "-rand_x" || a == "-rand-x") { c.random_init_x = need(i)[0]; if (c.random_init_x != '0' && c.random_init_x != '1' && c.random_init_x != 'm') die("You can only choose to initialize x randomly (1), with the matrix mean (m), or with the default value (0)."); }
        else if (a == "-comm_halos" || a == "-comm-halos") c.comm_halos = atoi(need(i));
        else if (a == "-ba_synch" || a == "-ba-synch") c.ba_synch = atoi(need(i));
        else if (a == "-par_pack" || a == "-par-pack") c.par_pack = atoi(need(i));
        else if (a == "-no_pack" || a == "-no-pack") c.no_pack = atoi(need(i));
        else if (a == "-print_comm_vol" || a == "-print-comm-vol") c.print_comm_vol = atoi(need(i));
        else if (a == "-ap_threshold_1" || a == "-apt1") { c.ap_threshold_1 = atof(need(i)); if (c.ap_threshold_1 < 0) die("ap threshold must be nonnegative."); }
        else if (a == "-ap_threshold_2" || a == "-apt2") { c.ap_threshold_2 = atof(need(i)); if (c.ap_threshold_2 < 0) die("ap threshold must be nonnegative."); }
        else if (a == "-dropout" || a == "-do") c.dropout = atoi(need(i));
        else if (a == "-dropout_threshold" || a == "-dt") c.dropout_threshold = atof(need(i));
        else if (a == "-equilibrate") c.equilibrate = atoi(need(i));
        else if (a == "-dp") c.value_type = "dp";
        else if (a == "-sp") c.value_type = "sp";
        else if (a == "-hp") c.value_type = "hp";
        else if (a == "-ap[dp_sp]" || a == "-ap[sp_hp]" || a == "-ap[dp_hp]" || a == "-ap[dp_sp_hp]") c.value_type = a.substr(1);
        else if (a == "-seg_rows" || a == "-seg-rows") c.seg_method = "seg-rows";
        else if (a == "-seg_nnz" || a == "-seg-nnz") c.seg_method = "seg-nnz";
        else if (a == "-seg_metis" || a == "-seg-metis") c.seg_method = "seg-metis";
        else if (a == "-part_file") c.part_file = need(i);
        else if (a == "-tlc") c.tlc = atoi(need(i));
        else if (a == "-dump_y") c.dump_y = need(i);
        else if (a == "-mpi_mode") { std::string v = need(i); if (v == "singlevec") c.vec_mode = USPMV_SINGLEVEC; else if (v == "multivec") c.vec_mode = USPMV_MULTIVEC; else if (v == "bulkvec") c.vec_mode = USPMV_BULKVEC; else die("mpi_mode must be singlevec, multivec or bulkvec."); }
        else if (a == "-graph") c.use_graph = atoi(need(i));
        else if (a == "-bench_steps") { c.bench_steps = atoi(need(i)); if (c.bench_steps < 1) die("bench_steps must be >= 1."); }
        else if (a == "-bench_warmup") { c.bench_warmup = atoi(need(i)); if (c.bench_warmup < 0) die("bench_warmup must be >= 0."); }
        else if (a == "-check_y") c.check_y = atoi(need(i));
        else if (a == "-json") c.json = need(i);
        else if (a == "-x_prepared") c.x_prepared = atoi(need(i));
        else if (a == "-tune") {                                // -tune key=value: a library tuning key (uspmv_set_tuning), applied at once; repeatable
            const std::string kv = need(i);
            const size_t eq = kv.find('=');
            if (eq == std::string::npos || eq == 0 || eq + 1 >= kv.size()) die("tune takes key=value.");
            if (uspmv_set_tuning(kv.substr(0, eq).c_str(), atoi(kv.c_str() + eq + 1)) != USPMV_OK) die(std::string("tune: ") + uspmv_last_error());
        }
        else if (a == "-convert") { c.convert = need(i); if (c.convert != "host" && c.convert != "device" && c.convert != "device_stable") die("convert must be host, device or device_stable."); }
        else if (a == "-step_form") { c.step_form = need(i); if (c.step_form != "auto" && c.step_form != "auto_all" && c.step_form != "overlap" && c.step_form != "plain" && c.step_form != "pad" && c.step_form != "fused") die("step_form must be auto, auto_all, overlap, plain, pad or fused."); }
        else if (a == "-block_vec_layout") { std::string v = need(i); if (v == "colwise") c.layout = USPMV_COLWISE; else if (v == "rowwise") c.layout = USPMV_ROWWISE; else die("block_vec_layout must be colwise or rowwise."); }
        else { usage(); die("unknown argument: " + a); }
    }
    // cross-option checks (code/utilities.hpp:1371-1545)
    if (c.layout == USPMV_ROWWISE && c.block_vec_size == 1)
        die("Row-wise block vector layout selected, but block vector width is 1.\n Please choose colwise block vector layout if using SpMV.");
    bool ap = c.value_type.rfind("ap[", 0) == 0;
    if (c.block_vec_size > 1 && ap) die("SpMMV is not yet implemented for AP kernels.");
    if (c.convert != "host" && ap) die("-convert device needs a one-precision run (-dp or -sp).");
    if (c.convert != "host" && uspmv_dist_requested()) die("-convert device is a single-rank option (every rank of a multi-rank run converts its block on the host).");
    if (c.seg_method == "seg-metis" && !uspmv_dist_requested()) die("seg-metis selected, but this is a single-rank run (the partition only matters across ranks).");
    if (c.seg_method == "seg-metis" && c.matrix_file_name.rfind("gen:", 0) == 0) die("seg-metis needs the whole matrix on rank 0: use a .mtx file (generated matrices are built per rank).");
    if (c.value_type == "hp" || c.value_type == "ap[sp_hp]" || c.value_type == "ap[dp_hp]" || c.value_type == "ap[dp_sp_hp]")
        die("Half precision selected, but HAVE_HALF_MATH not defined.");
    if (!ap && c.ap_threshold_1 > 0.0) fprintf(stderr, "WARNING: First adaptive precision threshold entered, but not used.\n");
    if (c.ap_threshold_2 > 0.0) fprintf(stderr, "WARNING: Second adaptive precision threshold entered, but three-way partitioning is not used.\n");
    if (ap && c.ap_threshold_1 == 0.0) fprintf(stderr, "WARNING: Two-way adaptive precision used, but the first threshold is not entered.\n");
    if (c.dropout && c.dropout_threshold == 0.0) fprintf(stderr, "WARNING: Dropout selected, but dropout_threshold is 0.\n");
    if (c.kernel_format != "crs" && c.kernel_format != "csr" && c.kernel_format != "scs") die("kernel format not recognized.");
    if (c.comm_halos && !uspmv_dist_requested()) {
        printf("single rank (WORLD_SIZE not set), forcing comm_halos = 0.\n");  // reference: "USE_MPI not defined, forcing comm_halos = 0."
        c.comm_halos = 0;
    }
    if (c.equilibrate != 0 && c.equilibrate != 1) die("You can only choose to equilibrate data (1, i.e. yes) or not (0, i.e. no).");
    if (c.equilibrate && ap) die("-equilibrate with ap[dp_sp] runs into undefined behaviour in the reference (empty scratch vectors, code/main.cpp:1143-1153); not offered.");
    // (-dropout / -dropout_threshold are parsed and printed by the reference but never applied, code/utilities.hpp:1281-1301)
    if (c.kernel_format != "scs") { c.chunk_size = 1; c.sigma = 1; }
    return c;
}

uspmv_coo_t *load_matrix(const Config &c) {
    uspmv_coo_t *m = nullptr;
    if (c.matrix_file_name.rfind("gen:", 0) == 0) {
        long nx = 0, ny = 0, nz = 0; int dof = 1; double dec = 0.0;
        int n = sscanf(c.matrix_file_name.c_str() + 4, "%ldx%ldx%ld:%d:%lf", &nx, &ny, &nz, &dof, &dec);
        if (n < 3) die("generator syntax: gen:NXxNYxNZ[:dof[:decades]]");
        ck(uspmv_gen_stencil27(nx, ny, nz, dof, 0x5EED, dec, 0, nx * ny * nz * dof, &m), "uspmv_gen_stencil27");
    } else {
        // USPMV_MTX_CACHE=1: keep / reuse a binary copy next to the .mtx file (<file>.uspmvcoo), newer than the text
        const char *use_cache = getenv("USPMV_MTX_CACHE");
        const std::string cache = c.matrix_file_name + ".uspmvcoo";
        struct stat st_m{}, st_c{};
        if (use_cache && atoi(use_cache) && stat(c.matrix_file_name.c_str(), &st_m) == 0 && stat(cache.c_str(), &st_c) == 0 &&
            st_c.st_mtime >= st_m.st_mtime && uspmv_coo_load(cache.c_str(), &m) == USPMV_OK) {
            // (binary copy is current)
        } else {
            ck(uspmv_read_mtx(c.matrix_file_name.c_str(), &m), "uspmv_read_mtx");
            if (use_cache && atoi(use_cache) && uspmv_coo_save(m, cache.c_str()) != USPMV_OK)
                fprintf(stderr, "warning: %s\n", uspmv_last_error());
        }
    }
    if (c.equilibrate) ck(uspmv_coo_equilibrate(m), "uspmv_coo_equilibrate");   // code/main.cpp:1117-1125
    return m;
}

template <typename T> T *dev_alloc(size_t n) {
    void *p = nullptr;
    hk(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)), "hipMalloc");
    hk(hipMemset(p, 0, std::max<size_t>(n, 1) * sizeof(T)), "hipMemset");
    return (T *)p;
}

struct Run {  // everything one kernel invocation needs
    uspmv_dmat_t *A = nullptr, *A_sp = nullptr;
    void *x = nullptr, *y = nullptr;
    int b = 1; long ld = 0; int layout = USPMV_COLWISE; bool ap = false;
    void exec_on(void *stream) const {
        int rc;
        if (ap) rc = uspmv_spmv_ap(A, A_sp, (const double *)x, (double *)y, stream);
        else if (b > 1) rc = uspmv_spmmv(A, x, y, b, ld, layout, stream);
        else rc = uspmv_spmv(A, x, y, stream);
        ck(rc, "kernel launch");
    }
    void exec() const { exec_on(nullptr); }
};

template <typename VT>
int run(const Config &c, uspmv_coo_t *coo) {
    const int dtype = sizeof(VT) == 8 ? USPMV_F64 : USPMV_F32;
    const bool ap = c.value_type == "ap[dp_sp]";
    const auto t_start = std::chrono::steady_clock::now();
    int64_t n_rows, n_cols, nnz;
    ck(uspmv_coo_dims(coo, &n_rows, &n_cols, &nnz), "uspmv_coo_dims");
    const double *vals; ck(uspmv_coo_arrays(coo, nullptr, nullptr, &vals), "uspmv_coo_arrays");
    double vmax = 0, vmin = 1e308;  // extract_matrix_min_mean_max (code/utilities.hpp:2502-2540)
    for (int64_t k = 0; k < nnz; ++k) { double a = std::fabs(vals[k]); vmax = std::max(vmax, a); vmin = std::min(vmin, a); }
    const double vmean = vmin + (vmax - vmin) / 2.0;

    // ---- format conversion (init_local_structs, code/main.cpp:1128-1221, :1308)
    uspmv_scs_t *scs = nullptr, *scs_sp = nullptr;
    uspmv_dmat_t *device_A = nullptr;          // -convert device*: the handle the device-side conversion made
    uspmv_coo_t *coo_dp = nullptr, *coo_sp = nullptr;
    int64_t meta[8], meta_sp[8] = {0};
    const int32_t *o2n, *n2o;
    if (ap) {
        ck(uspmv_partition_precisions(coo, c.ap_threshold_1, &coo_dp, &coo_sp), "uspmv_partition_precisions");
        ck(uspmv_convert_to_scs(coo_dp, c.chunk_size, c.sigma, USPMV_F64, nullptr, &scs), "convert dp struct");
        ck(uspmv_scs_arrays(scs, nullptr, nullptr, nullptr, nullptr, &o2n, &n2o), "uspmv_scs_arrays");
        ck(uspmv_convert_to_scs(coo_sp, c.chunk_size, c.sigma, USPMV_F32, o2n, &scs_sp), "convert sp struct");
        // symmetric permutation of BOTH structs with the dp permutation (the reference leaves this
        // as a TODO, code/main.cpp:1310-1332, and is only right for uniform x)
        ck(uspmv_permute_scs_cols(scs, o2n), "uspmv_permute_scs_cols");
        ck(uspmv_permute_scs_cols(scs_sp, o2n), "uspmv_permute_scs_cols");
        ck(uspmv_scs_meta(scs_sp, meta_sp), "uspmv_scs_meta");
    } else if (c.convert != "host") {
        // -convert device | device_stable: the COO arrays go to HBM once and convert_to_scs + permute_scs_cols run there
        // (uspmv_convert_to_scs_device_from_arrays; device: the reference's std::sort tie order, computed on the host from the row counts;
        // device_stable: ties in original order, nothing but the layout comes back).  `scs` is then a struct WITHOUT entries.
        const int32_t *ci_ = nullptr, *cj_ = nullptr;
        ck(uspmv_coo_arrays(coo, &ci_, &cj_, nullptr), "uspmv_coo_arrays");
        int32_t *dI = dev_alloc<int32_t>((size_t)nnz), *dJ = dev_alloc<int32_t>((size_t)nnz);
        double *dV = dev_alloc<double>((size_t)nnz);
        hk(hipMemcpy(dI, ci_, 4 * (size_t)nnz, hipMemcpyHostToDevice), "hipMemcpy I");
        hk(hipMemcpy(dJ, cj_, 4 * (size_t)nnz, hipMemcpyHostToDevice), "hipMemcpy J");
        hk(hipMemcpy(dV, vals, 8 * (size_t)nnz, hipMemcpyHostToDevice), "hipMemcpy V");
        ck(uspmv_convert_to_scs_device_from_arrays(dI, dJ, dV, n_rows, n_cols, nnz, c.chunk_size, c.sigma, dtype, nullptr, 1,
                                                   c.convert == "device_stable" ? USPMV_SORT_DEVICE_STABLE : USPMV_SORT_HOST, nullptr, &scs, nullptr, nullptr, &device_A),
           "uspmv_convert_to_scs_device_from_arrays");
        (void)hipFree(dI); (void)hipFree(dJ); (void)hipFree(dV);
        ck(uspmv_scs_arrays(scs, nullptr, nullptr, nullptr, nullptr, &o2n, &n2o), "uspmv_scs_arrays");
    } else {
        ck(uspmv_convert_to_scs(coo, c.chunk_size, c.sigma, dtype, nullptr, &scs), "uspmv_convert_to_scs");
        ck(uspmv_scs_arrays(scs, nullptr, nullptr, nullptr, nullptr, &o2n, &n2o), "uspmv_scs_arrays");
        ck(uspmv_permute_scs_cols(scs, o2n), "uspmv_permute_scs_cols");
    }
    ck(uspmv_scs_meta(scs, meta), "uspmv_scs_meta");
    const int64_t n_pad = meta[4], n_chunks = meta[5], n_el = meta[6];
    const int b = c.block_vec_size;
    const long ld = n_pad;  // padded_vec_size = n_rows + scs_padding without halos (code/main.cpp:1406-1412)

    // ---- x, y (DefaultValues x = 5.0, code/classes_structs.hpp:1799-1800; random: default-seeded mt19937)
    std::vector<VT> hx((size_t)b * ld, VT(0)), xo((size_t)n_rows), xp((size_t)n_rows);
    std::mt19937 engine;
    // random_init (code/utilities.hpp:880-912) draws dist(engine) = canonical * (max - min) + min.  The reference is built -O3
    // -march=native (Makefile), which contracts that into ONE fused multiply-add; this file may be compiled without FMA code
    // generation, so the fused form is spelled out -- the -rand_x 1 vector is then bit-identical to the reference's.
    auto draw = [&]() { return std::fma(std::generate_canonical<double, 53>(engine), vmax - vmin, vmin); };
    for (int v = 0; v < b; ++v) {
        // random_init draws for every element of the padded vector, padding is zeroed afterwards
        // (code/utilities.hpp:880-912, :955-980): consume ld draws per vector, keep the first n_rows
        for (int64_t i = 0; i < ld; ++i) {
            const VT val = c.random_init_x == '1' ? (VT)draw() : c.random_init_x == 'm' ? (VT)vmean : (VT)5.0;
            if (i < n_rows) xo[(size_t)i] = val;
        }
        ck(uspmv_apply_permutation(xp.data(), xo.data(), n2o, n_rows, dtype), "uspmv_apply_permutation");
        for (int64_t i = 0; i < n_rows; ++i) {
            if (c.layout == USPMV_ROWWISE) hx[(size_t)(i * b + v)] = xp[(size_t)i];
            else hx[(size_t)(v * ld + i)] = xp[(size_t)i];
        }
    }
    Run r;
    r.b = b; r.ld = ld; r.layout = c.layout; r.ap = ap;
    if (device_A) { r.A = device_A; printf("convert_to_scs on the device from the COO arrays in HBM (%s)\n", c.convert == "device_stable" ? "stable tie order" : "the reference's tie order"); }
    else ck(uspmv_dmat_upload(scs, &r.A), "uspmv_dmat_upload");
    if (ap) ck(uspmv_dmat_upload(scs_sp, &r.A_sp), "uspmv_dmat_upload");
    if (c.kernel_format != "scs") ck(uspmv_dmat_set_crs(r.A, 1), "uspmv_dmat_set_crs");
    if (c.tlc && (b == 1 || (!ap && c.chunk_size < 32 && 32 % c.chunk_size == 0))) {   // b > 1: for the internal C = 32 re-chunking of narrow chunks / crs
        int64_t nt = 0, ns = 0;
        if (ap) ck(uspmv_dmat_optimize_ap(r.A, r.A_sp, scs, scs_sp, 0, &nt, &ns), "uspmv_dmat_optimize_ap");
        else if (device_A) ck(uspmv_dmat_optimize_device(r.A, 0, &nt, &ns), "uspmv_dmat_optimize_device");   // (no host entries: the plan is built on the device too)
        else ck(uspmv_dmat_optimize(r.A, scs, 0, &nt, &ns), "uspmv_dmat_optimize");
        int gran = 0, dealt = 0;
        (void)uspmv_dmat_plan_granularity(r.A, &gran); (void)uspmv_dmat_plan_rows_dealt(r.A, &dealt);
        printf("tile-local-column plan: %ld of %ld tiles staged in LDS%s%s\n", (long)ns, (long)nt, gran == 1 ? " (lists of single x elements)" : "",
               dealt ? ", rows dealt to the tiles by the matrix graph" : "");
    } else if (c.tlc && b > 1 && !ap && (size_t)b * sizeof(VT) <= 32) {   // block plan pays for rows of <= 32 bytes
        int64_t nt = 0, ns = 0;
        if (device_A) ck(uspmv_dmat_optimize_block_device(r.A, b, &nt, &ns), "uspmv_dmat_optimize_block_device");
        else ck(uspmv_dmat_optimize_block(r.A, scs, b, &nt, &ns), "uspmv_dmat_optimize_block");
        if (nt) printf("block plan: %ld of %ld tiles staged in LDS\n", (long)ns, (long)nt);
    }
    r.x = dev_alloc<VT>((size_t)b * ld);
    r.y = dev_alloc<VT>((size_t)b * ld);
    hk(hipMemcpy(r.x, hx.data(), sizeof(VT) * hx.size(), hipMemcpyHostToDevice), "hipMemcpy x");

    const auto t_setup_end = std::chrono::steady_clock::now();
    if (c.kernel_format == "scs") printf("C = %ld => %s SCS Sp%sV kernel selected (gfx950)\n", c.chunk_size, ap ? "ap[dp_sp]" : "one-precision", b > 1 ? "MM" : "M");
    else printf("CRS Sp%sV kernel selected (gfx950)\n", b > 1 ? "MM" : "M");

    double perf = 0, runtime = 0; int n_iter = 0;
    float kernel_ms = 0.f;
    // bench mode multiplies the SAME block vector in every iteration (code/main.cpp:458-519): a column-major X is re-laid out once
    // (uspmv_spmmv_x_prepared) instead of per call; -x_prepared 0 keeps the per-call pass.  Solve mode writes x every revision: never.
    if (c.mode == 'b' && b > 1 && c.layout == USPMV_COLWISE && c.x_prepared) ck(uspmv_spmmv_x_prepared(r.A, r.x, b, ld, nullptr), "uspmv_spmmv_x_prepared");
    if (c.mode == 'b' && c.bench_steps > 0) {
        // -bench_steps K [-bench_warmup W]: exactly K launches between two synchronisations, wall clock and HIP events (the protocol
        // bench.py asks the multi-rank harness for; here it yields the one-GPU time of the same matrix for the strong-scaling line)
        const int warm = c.bench_warmup >= 0 ? c.bench_warmup : WARM_UP_REPS;
        for (int k = 0; k < warm; ++k) r.exec();
        hk(hipDeviceSynchronize(), "hipDeviceSynchronize");
        hipEvent_t e0, e1;
        hk(hipEventCreate(&e0), "hipEventCreate"); hk(hipEventCreate(&e1), "hipEventCreate");
        const auto t0 = std::chrono::steady_clock::now();
        hk(hipEventRecord(e0, nullptr), "hipEventRecord");
        for (int k = 0; k < c.bench_steps; ++k) r.exec();
        hk(hipEventRecord(e1, nullptr), "hipEventRecord");
        hk(hipDeviceSynchronize(), "hipDeviceSynchronize");
        runtime = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        hk(hipEventElapsedTime(&kernel_ms, e0, e1), "hipEventElapsedTime");
        kernel_ms /= (float)c.bench_steps;
        n_iter = c.bench_steps;
        perf = (double)nnz * 2.0 * b / (runtime / n_iter) / 1e9;
    } else if (c.mode == 'b') {
        for (int k = 0; k < WARM_UP_REPS; ++k) r.exec();
        hk(hipDeviceSynchronize(), "hipDeviceSynchronize");
        hipEvent_t e0, e1;
        hk(hipEventCreate(&e0), "hipEventCreate"); hk(hipEventCreate(&e1), "hipEventCreate");
        n_iter = 2;
        float ms = 0.f;
        // launch-bound matrices: replay a captured graph of GRAPH_BATCH kernel launches instead of
        // GRAPH_BATCH host launches (same kernels, same count; only the host-side launch cost changes)
        constexpr int GRAPH_BATCH = 64;
        hipStream_t gs = nullptr;
        hipGraph_t graph = nullptr;
        hipGraphExec_t gexec = nullptr;
        if (c.use_graph) {
            hk(hipStreamCreate(&gs), "hipStreamCreate");
            hk(hipStreamBeginCapture(gs, hipStreamCaptureModeGlobal), "hipStreamBeginCapture");
            for (int k = 0; k < GRAPH_BATCH; ++k) r.exec_on(gs);
            hk(hipStreamEndCapture(gs, &graph), "hipStreamEndCapture");
            hk(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0), "hipGraphInstantiate");
        }
        do {  // code/main.cpp:483-518
            hk(hipEventRecord(e0, gs), "hipEventRecord");
            if (gexec && n_iter >= GRAPH_BATCH) for (int k = 0; k < n_iter / GRAPH_BATCH; ++k) hk(hipGraphLaunch(gexec, gs), "hipGraphLaunch");
            else for (int k = 0; k < n_iter; ++k) r.exec_on(gs);
            hk(hipEventRecord(e1, gs), "hipEventRecord");
            hk(hipEventSynchronize(e1), "hipEventSynchronize");
            hk(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime");
            n_iter *= 2;
            runtime = ms * 1e-3;
        } while (runtime < c.bench_time);
        n_iter /= 2;
        perf = (double)nnz * 2.0 * b / (runtime / n_iter) / 1e9;
    } else {  // solve mode: repeated y = A x with x <- y, then cross-check against the CRS kernel on the device
        std::vector<VT> hy((size_t)b * ld);
        for (unsigned long i = 0; i < c.n_repetitions; ++i) { r.exec(); if (i + 1 < c.n_repetitions) std::swap(r.x, r.y); }
        hk(hipMemcpy(hy.data(), r.y, sizeof(VT) * hy.size(), hipMemcpyDeviceToHost), "hipMemcpy y");
        if (!c.dump_y.empty()) {   // copy_back_result (code/utilities.hpp:3862): y_orig[i] = y[old_to_new[i]]; block vectors: column after column
            std::vector<VT> yo((size_t)n_rows * (size_t)b);
            for (int v = 0; v < b; ++v)
                for (int64_t i = 0; i < n_rows; ++i)
                    yo[(size_t)v * (size_t)n_rows + (size_t)i] = c.layout == USPMV_ROWWISE ? hy[(size_t)o2n[i] * (size_t)b + (size_t)v] : hy[(size_t)v * (size_t)ld + (size_t)o2n[i]];
            std::ofstream fy(c.dump_y, std::ios::binary);
            fy.write((const char *)yo.data(), (std::streamsize)(sizeof(VT) * yo.size()));
            if (!fy) die("cannot write " + c.dump_y);
        }
        if (c.validate_result && b == 1 && !ap) {
            uspmv_scs_t *crs = nullptr; uspmv_dmat_t *Ac = nullptr;
            ck(uspmv_convert_to_scs(coo, 1, 1, dtype, nullptr, &crs), "convert crs");
            ck(uspmv_dmat_upload(crs, &Ac), "upload crs");
            ck(uspmv_dmat_set_crs(Ac, 1), "set crs");
            std::vector<VT> xr((size_t)n_rows, c.random_init_x == 'm' ? (VT)vmean : (VT)5.0), yr((size_t)n_rows);
            if (c.random_init_x == '1') { std::mt19937 e2; for (auto &v : xr) v = (VT)std::fma(std::generate_canonical<double, 53>(e2), vmax - vmin, vmin); }
            VT *dxr = dev_alloc<VT>((size_t)n_rows), *dyr = dev_alloc<VT>((size_t)n_rows);
            hk(hipMemcpy(dxr, xr.data(), sizeof(VT) * xr.size(), hipMemcpyHostToDevice), "hipMemcpy");
            for (unsigned long i = 0; i < c.n_repetitions; ++i) { ck(uspmv_spmv(Ac, dxr, dyr, nullptr), "crs spmv"); if (i + 1 < c.n_repetitions) std::swap(dxr, dyr); }
            hk(hipMemcpy(yr.data(), dyr, sizeof(VT) * yr.size(), hipMemcpyDeviceToHost), "hipMemcpy");
            double max_rel = 0, ymax = 0;  // copy_back_result: y_orig[i] = y[old_to_new[i]] (code/utilities.hpp:3862)
            for (int64_t i = 0; i < n_rows; ++i) ymax = std::max(ymax, (double)std::fabs(yr[(size_t)i]));
            // relative difference per element as in the reference (code/write_results.hpp:350-383), except that
            // elements which cancel to (almost) nothing are measured against 1e-8 of the largest |y| -- the two kernels
            // sum in different orders, so such elements differ by rounding noise that is huge relative to themselves
            const double floor_ = std::max(1e-8 * ymax, 1e-300);
            for (int64_t i = 0; i < n_rows; ++i) {
                double a = hy[(size_t)o2n[i]], bb = yr[(size_t)i];
                double rel = std::fabs(a - bb) / std::max(std::fabs(bb), floor_);
                max_rel = std::max(max_rel, rel);
            }
            // thresholds of write_result_to_file (code/write_results.hpp:378-383, :422-428)
            const char *verdict = max_rel > 1e-2 ? "ERROR" : max_rel > 1e-4 ? "WARNING" : "OK";
            printf("validation vs CRS kernel: max relative difference %.3e -> %s\n", max_rel, verdict);
            std::ofstream f(c.value_type == "sp" ? "spmv_mkl_compare_sp.txt" : "spmv_mkl_compare_dp.txt", std::ios::app);
            f << c.matrix_file_name << " kernel: " << c.kernel_format << " C: " << c.chunk_size << " sigma: " << c.sigma
              << " revisions: " << c.n_repetitions << " max_rel_diff_vs_crs: " << std::setprecision(16) << max_rel << " " << verdict << "\n";
            if (max_rel > 1e-2) return 2;
        }
        printf("solve mode: %lu revision(s) done\n", c.n_repetitions);
        return 0;
    }

    // ---- report (write_bench_to_file, code/write_results.hpp:42-157)
    const double beta = (double)nnz / (double)n_el;
    const double vsz = sizeof(VT);
    double bytes = ap ? 12.0 * n_el + 8.0 * meta_sp[6] + 16.0 * n_chunks + 8.0 * (n_cols + n_pad)
                      : n_el * (vsz + 4) + 8.0 * n_chunks + b * vsz * n_cols + b * vsz * n_pad;
    const double t_iter = runtime / n_iter, gbs = bytes / t_iter / 1e9;
    int tpb = 0; uspmv_get_tuning("block", &tpb);
    const long blocks = (n_pad + tpb - 1) / tpb;
    std::ofstream f(c.output_filename_bench, std::ios::app);
    const int w = 32;
    f << c.matrix_file_name << " with " << blocks << " block(s), and " << tpb << " thread(s) per block" << std::endl;
    f << "kernel: " << c.kernel_format << ", block_vec_size: " << b;
    if (c.kernel_format == "scs") {
        f << ", C: " << c.chunk_size << " sigma: " << c.sigma;
        if (ap) f << std::fixed << std::setprecision(2) << ", dp_beta: " << (double)meta[7] / n_el << ", sp_beta: " << (meta_sp[6] ? (double)meta_sp[7] / meta_sp[6] : 0.0);
        else f << std::fixed << std::setprecision(8) << ", beta: " << beta;
    }
    f << ", block_vec_layout: " << (c.layout == USPMV_ROWWISE ? "rowwise" : "colwise");
    if (ap) f << ", data_type: ap[dp_sp]" << ", threshold: " << std::fixed << std::setprecision(2) << c.ap_threshold_1
              << ", % dp elems: " << 100.0 * meta[7] / nnz << ", % sp elems: " << 100.0 * meta_sp[7] / nnz;
    else f << ", data_type: " << (c.value_type == "dp" ? "double" : "float");
    f << ", revisions: " << n_iter << std::endl << std::endl;
    f << std::left << std::setw(w) << "Total Gflops:" << std::left << std::setw(w) << "Total Walltime:" << std::endl;
    f << std::left << std::setw(w) << "-------------" << std::left << std::setw(w) << "-------------" << std::endl;
    f << std::left << std::setprecision(16) << std::left << std::setw(w) << perf << std::left << std::setw(w) << runtime << std::endl << std::endl;
    f << std::left << std::setw(w) << "Achieved GB/s:" << std::left << std::setw(w) << "Fraction of 8.0 TB/s:" << std::endl;
    f << std::left << std::setw(w) << "-------------" << std::left << std::setw(w) << "-------------" << std::endl;
    f << std::left << std::setprecision(6) << std::setw(w) << gbs << std::left << std::setw(w) << gbs / HBM_PEAK_GBS << std::endl << std::endl;
    if (ap) printf("n_rows = %ld, nnz = %ld, dp: %ld nnz in %ld elements, sp: %ld nnz in %ld elements\n", (long)n_rows, (long)nnz,
                   (long)meta[7], (long)n_el, (long)meta_sp[7], (long)meta_sp[6]);
    else printf("n_rows = %ld, nnz = %ld, n_elements = %ld, beta = %.8f\n", (long)n_rows, (long)nnz, (long)n_el, beta);
    printf("Total Gflops: %.4f   (%d iterations in %.4f s, %.6f ms per SpMV)\n", perf, n_iter, runtime, t_iter * 1e3);
    printf("Achieved GB/s: %.1f   (%.1f %% of the %.0f GB/s HBM3E roofline; algorithmic bytes %.0f per SpMV)\n", gbs,
           100.0 * gbs / HBM_PEAK_GBS, HBM_PEAK_GBS, bytes);
    if (!c.json.empty()) {   // -json <file|->: the single-rank twin of the multi-rank report
        int64_t pi[3] = {0, 0, 0};
        int pk = 0;
        (void)uspmv_dmat_plan_info(r.A, &pk, &pi[1], &pi[2]);
        pi[0] = pk;
        char js[1024];
        snprintf(js, sizeof js,
                 "{\"gflops\": %.4f, \"ms_per_step\": %.6f, \"kernel_ms\": %.6f, \"steps\": %d, \"warmup\": %d, \"runtime_s\": %.6f, \"ranks\": 1, "
                 "\"n_rows\": %ld, \"nnz\": %ld, \"n_elements\": %ld, \"n_chunks\": %ld, \"n_rows_padded\": %ld, \"algorithmic_bytes\": %.0f, "
                 "\"algorithmic_GBs\": %.1f, \"plan_kind\": %ld, \"plan_tiles\": %ld, \"plan_tiles_planned\": %ld, \"setup_s\": %.2f}",
                 perf, t_iter * 1e3, (double)kernel_ms, n_iter, c.bench_steps > 0 ? (c.bench_warmup >= 0 ? c.bench_warmup : WARM_UP_REPS) : WARM_UP_REPS, runtime,
                 (long)n_rows, (long)nnz, (long)n_el, (long)n_chunks, (long)n_pad, bytes, gbs, (long)pi[0], (long)pi[1], (long)pi[2],
                 std::chrono::duration<double>(t_setup_end - t_start).count());
        if (c.json == "-") printf("%s\n", js);
        else { std::ofstream jf(c.json); jf << js << std::endl; }
    }
    return 0;
}

}  // namespace

int main(int argc, char **argv) {
    Config c = parse(argc, argv);
    if (uspmv_dist_requested()) {  // one process per GPU, halo exchange on RCCL (uspmv_dist.cpp)
        if (c.value_type != "dp" && c.value_type != "sp")
            die("multi-rank runs are -dp or -sp (single vector or -block_vec_size b, crs or scs, -mode b or s) "
                "(the reference also refuses ap with MPI, code/utilities.hpp:1443-1450)");
        if (c.mode == 's' && c.block_vec_size > 1) die("multi-rank solve mode takes a single vector.");
        // crs across ranks: the reference's C = 1, sigma = 1 struct (code/utilities.hpp:1420-1424) on the SELL kernels -- every row still the
        // entry-ordered FMA chain (the single-rank crs kernel reassociates like the reference's omp simd loop)
        if (c.kernel_format == "crs") { c.chunk_size = 1; c.sigma = 1; }
        if (c.layout == USPMV_ROWWISE && c.vec_mode != USPMV_BULKVEC) die("row-wise block vectors are exchanged in bulkvec mode only.");
        DistConfig d;
        d.C = c.chunk_size; d.sigma = c.sigma; d.seg_nnz = c.seg_method == "seg-nnz"; d.comm_halos = c.comm_halos != 0;
        d.seg_metis = c.seg_method == "seg-metis"; d.part_file = c.part_file;
        d.ba_synch = c.ba_synch != 0; d.tlc = c.tlc != 0; d.verbose = c.verbose != 0; d.bench_time = c.bench_time;
        d.matrix_name = c.matrix_file_name;
        d.block_vec_size = c.block_vec_size; d.layout = c.layout; d.vec_mode = c.vec_mode;
        d.use_graph = c.use_graph != 0; d.print_comm_vol = c.print_comm_vol != 0; d.no_pack = c.no_pack != 0;
        d.no_overlap = getenv("USPMV_NO_OVERLAP") != nullptr;
        d.step_form = c.step_form;
        d.mode = c.mode; d.n_repetitions = c.n_repetitions; d.dump_y = c.dump_y; d.sp = c.value_type == "sp"; d.random_init_x = c.random_init_x;
        d.bench_steps = c.bench_steps; d.bench_warmup = c.bench_warmup; d.json = c.json;
        // -validate (default 1) in solve mode: the reference gathers y and compares with MKL (code/write_results.hpp:442-556); across ranks
        // here the bitwise self-check of one step plays that role, as -check_y 1 does in bench mode
        d.check_y = c.check_y != 0 || (c.mode == 's' && c.validate_result != 0 && c.comm_halos != 0);
        d.equilibrate = c.equilibrate != 0;
        // (-par_pack: on the device the send buffer is packed by one kernel either way, as in the reference's device branch, code/classes_structs.hpp:787-806)
        return uspmv_run_distributed(d);   // every rank generates / receives only its row block
    }
    int ndev = 0;
    ck(uspmv_device_count(&ndev), "uspmv_device_count");
    if (ndev < 1) die("no HIP device visible: uspmv has no CPU path");
    ck(uspmv_set_device(0), "uspmv_set_device");
    uspmv_coo_t *coo = load_matrix(c);
    int rc = c.value_type == "sp" ? run<float>(c, coo) : run<double>(c, coo);
    uspmv_coo_free(coo);
    return rc;
}
