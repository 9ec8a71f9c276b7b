// Multi-rank path of the `uspmv` harness: one process per GPU, halo exchange on RCCL (xGMI).
//
// Replaces, for the SINGLEVEC colwise one-precision path, the reference's MPI flow:
//   init_local_structs            code/main.cpp:1075-1334   (partition, convert, halo discovery)
//   collect_comm_info             code/mpi_funcs.hpp:1061-1124 (who sends what to whom)
//   init/finalize_halo_exchange   code/classes_structs.hpp:857-995 (per-iteration exchange)
//   bench loop with barriers      code/main.cpp:458-474
// All of it lives behind the C ABI (uspmv_dist_* in include/uspmv.h, csrc/uspmv_dist_api.hip); this file is the harness around it:
// rank discovery, the RCCL id hand-off, the per-rank matrix block, the bench protocol and the report.
//
// Launch: any launcher that exports RANK / WORLD_SIZE / LOCAL_RANK (torchrun, srun with a wrapper,
// `for r in ...; do RANK=$r WORLD_SIZE=$N LOCAL_RANK=$r uspmv ... & done`).  The ranks of one node meet in a host communicator
// (uspmv_hostcomm_*, host/hostcomm.cpp) keyed by $USPMV_JOB_ID / $MASTER_PORT: the RCCL unique id, the partition and the set-up
// exchanges of collect_comm_info travel through it (USPMV_SETUP_TRANSPORT=rccl moves the latter onto the RCCL communicator); the
// per-step halo exchange is RCCL.  A leftover of a crashed job cannot be picked up (dead creator / fresh nonce per segment).
//
// Every rank holds ONLY its row block: generated matrices (gen:...) are generated per block (the partition comes from the
// analytic row counts); a .mtx file is read by rank 0 alone, which writes one binary block per rank (file names carry the
// segment's nonce) -- the reference's root-reads-and-scatters (code/mpi_funcs.hpp:739-860) without MPI.
//
// Protocols: the reference's (100 warm-ups, doubling batches until -bench_time, code/main.cpp:408-474) by default;
// `-bench_steps K [-bench_warmup W]` times EXACTLY K steps between barriers (what bench.py --gpus N asks for).  -ba_synch 1
// (default, as in the reference) puts a barrier behind every step (a stream-ordered all-reduce, part of the captured graph);
// `-json <file|->` adds one JSON line with everything measured, `-check_y 1` the bitwise self-check of uspmv_dist_check.
//
// Single-GPU rehearsal of the whole path: USPMV_LOOPBACK=P (with WORLD_SIZE unset) makes this process block
// $USPMV_LOOPBACK_RANK (default 0) of a P-way partition whose neighbours are itself (RCCL self send/recv); USPMV_EXCHANGE=host
// with WORLD_SIZE = P real processes stages the halo exchange through the host communicator instead of RCCL, so that the
// ranks may share one GPU (unequal seg-nnz blocks included).
// USPMV_DIST_X=ramp sets x_local[i] = 1 + 1e-3 * (i mod 1000) in ORIGINAL local row order on every rank (the default is the
// reference's constant 5.0); USPMV_DUMP_Y=<prefix> writes y of the local rows in original order to <prefix>.<rank> (raw
// doubles) after one step (with -seg_metis also <prefix>.perm: permuted row r = original row perm[r]).  tests/test_dist_native_gpu.py
// checks that output against the oracle.
#include <hip/hip_runtime_api.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <random>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "uspmv.h"
#include "uspmv_dist.hpp"

namespace {

struct Ctx { int rank = 0; uspmv_hostcomm_t *hc = nullptr; };
Ctx g;

[[noreturn]] void die(const std::string &msg) {
    fprintf(stderr, "[rank %d] ERROR: %s\n", g.rank, msg.c_str());
    if (g.hc) uspmv_hostcomm_abort(g.hc);     // the peers fail at their next wait instead of sitting in it
    exit(1);
}
#define CK(call) do { int rc_ = (call); if (rc_ != USPMV_OK) die(std::string(#call) + ": " + uspmv_last_error()); } while (0)
#define HK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) die(std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

// USPMV_STAGES=1: one line on stderr per stage a rank reaches -- a launcher that has to kill a hung run (bench.py --gpus N) can say
// WHERE it hung (communicator creation, set-up, the timed steps ...), and whether a later tier is worth trying
void stage(const char *what) {
    static const bool on = getenv("USPMV_STAGES") != nullptr;
    if (on) { fprintf(stderr, "[uspmv stage] %s (rank %d)\n", what, g.rank); fflush(stderr); }
}

int env_int(const char *a, const char *b, int dflt) {
    const char *v = getenv(a);
    if (!v && b) v = getenv(b);
    return v ? atoi(v) : dflt;
}

std::string job_key() {
    const char *job = getenv("USPMV_JOB_ID") ? getenv("USPMV_JOB_ID") : getenv("MASTER_PORT") ? getenv("MASTER_PORT") : "0";
    return std::string("cli_") + job;
}

std::string block_path(uint64_t nonce, int r) {
    const char *dir = getenv("USPMV_ID_DIR");
    char buf[64];
    snprintf(buf, sizeof buf, "/uspmv_%016llx_block%d.uspmvcoo", (unsigned long long)nonce, r);
    return std::string(dir ? dir : "/tmp") + buf;
}

void publish(const std::string &path, const void *data, size_t bytes) {
    const std::string tmp = path + ".tmp";
    { std::ofstream f(tmp, std::ios::binary); f.write((const char *)data, (std::streamsize)bytes); }
    if (rename(tmp.c_str(), path.c_str()) != 0) die("cannot publish " + path);
}

}  // namespace

bool uspmv_dist_requested() {
    return env_int("WORLD_SIZE", "USPMV_WORLD_SIZE", 1) > 1 || getenv("USPMV_FORCE_DIST") || env_int("USPMV_LOOPBACK", nullptr, 0) > 1;
}

int uspmv_run_distributed(const DistConfig &c) {
    const int comm_rank = env_int("RANK", "USPMV_RANK", 0), comm_size = env_int("WORLD_SIZE", "USPMV_WORLD_SIZE", 1);
    const int local_rank = env_int("LOCAL_RANK", "USPMV_LOCAL_RANK", comm_rank);
    const int loop_P = env_int("USPMV_LOOPBACK", nullptr, 0);
    if (loop_P > 1 && comm_size != 1) die("USPMV_LOOPBACK needs WORLD_SIZE = 1");
    const int P = loop_P > 1 ? loop_P : comm_size;
    const int rank = loop_P > 1 ? env_int("USPMV_LOOPBACK_RANK", nullptr, 0) : comm_rank;
    if (rank < 0 || rank >= P) die("bad rank");
    g.rank = rank;
    const char *exk = getenv("USPMV_EXCHANGE");
    const bool host_exchange = exk && !strcmp(exk, "host");
    if (host_exchange && loop_P > 1) die("USPMV_EXCHANGE=host needs real ranks (WORLD_SIZE = P), not USPMV_LOOPBACK");
    if (getenv("USPMV_BACKTRACE")) uspmv_debug_backtrace_on_crash(1);
    int ndev = 0;
    CK(uspmv_device_count(&ndev));
    if (ndev < 1) die("no HIP device visible");
    CK(uspmv_set_device(local_rank % ndev));  // device = my_rank % num_devices (code/main.cpp:1838-1842)

    // ---- the ranks meet (MPI_Init's part); the RCCL id travels through the host communicator
    stage("device set, waiting for the other ranks");
    CK(uspmv_hostcomm_create(job_key().c_str(), comm_rank, comm_size, (double)env_int("USPMV_HC_TIMEOUT", nullptr, 3600), &g.hc));
    stage("ranks met");
    uint64_t nonce = 0;
    CK(uspmv_hostcomm_info(g.hc, nullptr, nullptr, &nonce));
    unsigned char id[USPMV_COMM_ID_BYTES] = {0};
    if (!host_exchange) {
        if (comm_rank == 0) CK(uspmv_comm_unique_id(id));
        CK(uspmv_hostcomm_bcast(g.hc, id, sizeof id, 0));
    }

    // ---- this rank's row block (and nothing else)
    std::vector<int32_t> wsa((size_t)P + 1, 0), metis_perm;
    uspmv_coo_t *local = nullptr;
    int64_t n_rows_g = 0, nnz_g = 0;
    const int seg = c.seg_nnz ? USPMV_SEG_NNZ : USPMV_SEG_ROWS;
    if (c.matrix_name.rfind("gen:", 0) == 0) {
        long nx = 0, ny = 0, nz = 0; int dof = 1; double dec = 0.0;
        if (sscanf(c.matrix_name.c_str() + 4, "%ldx%ldx%ld:%d:%lf", &nx, &ny, &nz, &dof, &dec) < 3) die("generator syntax: gen:NXxNYxNZ[:dof[:decades]]");
        n_rows_g = (int64_t)nx * ny * nz * dof;
        {
            std::vector<int32_t> counts((size_t)n_rows_g);
            CK(uspmv_gen_stencil27_row_counts(nx, ny, nz, dof, 0, n_rows_g, counts.data()));
            for (int32_t v : counts) nnz_g += v;
            CK(uspmv_seg_from_row_counts(counts.data(), n_rows_g, seg, P, wsa.data()));
        }
        CK(uspmv_gen_stencil27(nx, ny, nz, dof, 0x5EED, dec, wsa[(size_t)rank], wsa[(size_t)rank + 1], &local));
    } else {
        std::vector<int64_t> meta((size_t)P + 3, 0);
        if (comm_rank == 0) {
            uspmv_coo_t *total = nullptr;
            CK(uspmv_read_mtx(c.matrix_name.c_str(), &total));
            int64_t nc;
            CK(uspmv_coo_dims(total, &n_rows_g, &nc, &nnz_g));
            if (c.seg_metis) {   // graph partition, rows sorted by part, matrix permuted symmetrically (code/mpi_funcs.hpp:494-598)
                std::vector<int32_t> part((size_t)n_rows_g);
                if (!c.part_file.empty()) CK(uspmv_read_partition(c.part_file.c_str(), n_rows_g, P, part.data()));
                else CK(uspmv_graph_partition(total, P, part.data()));
                uspmv_coo_t *permuted = nullptr;
                metis_perm.resize((size_t)n_rows_g);             // new row r = old row perm[r]: kept, so that dumped results can be mapped back
                CK(uspmv_coo_apply_partition(total, P, part.data(), &permuted, wsa.data(), metis_perm.data()));
                uspmv_coo_free(total);
                total = permuted;
                printf("seg-metis: METIS is not linked; rows partitioned by %s\n", c.part_file.empty() ? "the built-in level-set partitioner" : c.part_file.c_str());
            } else CK(uspmv_seg_work_sharing_arr(total, seg, P, wsa.data()));
            for (int r = 0; r < P; ++r) {
                uspmv_coo_t *blk = nullptr;
                CK(uspmv_seg_local_coo(total, wsa.data(), r, &blk));
                if (r == rank) local = blk;
                else {
                    if (comm_size > 1) CK(uspmv_coo_save(blk, block_path(nonce, r).c_str()));
                    uspmv_coo_free(blk);
                }
            }
            uspmv_coo_free(total);
            meta[0] = n_rows_g; meta[1] = nnz_g;
            for (int r = 0; r <= P; ++r) meta[(size_t)r + 2] = wsa[(size_t)r];
        }
        CK(uspmv_hostcomm_bcast(g.hc, meta.data(), (int64_t)(meta.size() * 8), 0));   // (also: the blocks are on disk now)
        if (comm_rank != 0) {
            n_rows_g = meta[0]; nnz_g = meta[1];
            for (int r = 0; r <= P; ++r) wsa[(size_t)r] = (int32_t)meta[(size_t)r + 2];
            CK(uspmv_coo_load(block_path(nonce, rank).c_str(), &local));
            unlink(block_path(nonce, rank).c_str());
        }
    }

    stage("matrix block built");
    // ---- the distributed object: convert, halo discovery, upload, plan, exchange plan, communicator (init_local_structs + collect_comm_info)
    uspmv_dist_t *D = nullptr;
    uspmv_transport_t tr{};
    uspmv_dist_options_t opt{nullptr, host_exchange ? USPMV_EXCHANGE_HOST : USPMV_EXCHANGE_RCCL};
    const char *stk = getenv("USPMV_SETUP_TRANSPORT");
    const bool setup_on_rccl = stk && !strcmp(stk, "rccl") && !host_exchange;
    if (comm_size == P && P > 1 && !setup_on_rccl) { CK(uspmv_hostcomm_transport(g.hc, &tr)); opt.transport = &tr; }
    // -rand_x 1 | m: min / max of |values| over the WHOLE matrix, taken before any scaling (extract_matrix_min_mean_max on rank 0 +
    // MPI_Bcast in the reference, code/main.cpp:1096, code/utilities.hpp:2502-2540; here every rank contributes its block)
    double vmin = 1e308, vmax = 0;
    if (c.random_init_x != '0') {
        int64_t nr_ = 0, nc_ = 0, nz = 0;
        const int32_t *ci = nullptr, *cj = nullptr;
        const double *cv = nullptr;
        CK(uspmv_coo_dims(local, &nr_, &nc_, &nz));
        CK(uspmv_coo_arrays(local, &ci, &cj, &cv));
        for (int64_t k = 0; k < nz; ++k) { const double a = std::fabs(cv[k]); vmax = std::max(vmax, a); vmin = std::min(vmin, a); }
        double neg_min = -vmin;
        CK(uspmv_hostcomm_allreduce_max_f64(g.hc, &vmax));
        CK(uspmv_hostcomm_allreduce_max_f64(g.hc, &neg_min));
        vmin = -neg_min;
    }
    // -equilibrate 1: every rank scales ITS block (rows, then columns of the row-scaled block), as the reference does after the
    // segmentation (code/main.cpp:1117-1125 on local_mtx)
    if (c.equilibrate) CK(uspmv_coo_equilibrate(local));
    CK(uspmv_dist_create_from_coo_ex(host_exchange ? nullptr : id, comm_rank, comm_size, rank, P, local, wsa.data(), c.C, c.sigma, c.sp ? USPMV_F32 : USPMV_F64, c.tlc ? 1 : 0, &opt, &D));
    // (the block's COO stays until the end: -rand_x reads its values, -step_form auto and -check_y run the self-check against it)
    stage("step object created (communicator up)");
    hipStream_t st = nullptr;
    HK(hipStreamCreate(&st));
    CK(uspmv_dist_barrier(D, st));
    stage("first collective done");
    if (c.no_overlap) CK(uspmv_dist_set_option(D, "overlap", 0));
    if (const char *fs = getenv("USPMV_FUSED_STEP")) CK(uspmv_dist_set_option(D, "fused_step", atoi(fs) != 0));
    if (const char *ps = getenv("USPMV_PAD_SPLIT")) CK(uspmv_dist_set_option(D, "pad_split", atoi(ps) != 0));   // (A/B of the padding tiles, tools/ab_dist_step.sh)
    if (c.no_pack) CK(uspmv_dist_set_option(D, "no_pack", 1));   // -no_pack 1: the exchange sends a stale buffer (code/classes_structs.hpp:941)
    if (c.block_vec_size > 1 && c.tlc) CK(uspmv_dist_set_option(D, "block_plan", c.block_vec_size));   // (64-byte X rows: the phased block plan + its interior / boundary tiles)
    CK(uspmv_dist_set_option(D, "ba_synch", c.ba_synch && c.comm_halos ? 1 : 0));   // -ba_synch (code/main.cpp:467; default 1, code/classes_structs.hpp:90)
    int64_t meta[12];
    CK(uspmv_dist_info(D, meta));
    const int64_t n_local = meta[0], n_halo = meta[1], vec_len = meta[2], n_send = meta[3];
    const uspmv_scs_t *scs = nullptr;
    CK(uspmv_dist_parts(D, &scs, nullptr, nullptr));
    int64_t sm[8];
    CK(uspmv_scs_meta(scs, sm));
    const int64_t n_pad = sm[4], n_chunks = sm[5], n_el = sm[6];
    const int32_t *o2n, *n2o;
    CK(uspmv_scs_arrays(scs, nullptr, nullptr, nullptr, nullptr, &o2n, &n2o));

    // ---- vectors: x = DefaultValues::x = 5.0 on the local rows (or the test ramp, scaled by 1 + v/8 for vector v), permuted
    //      (code/main.cpp:86-93); 0 elsewhere.  Block vectors: b * padded_vec_size elements, column- or row-wise.
    const int b = c.block_vec_size;
    const int dtype = c.sp ? USPMV_F32 : USPMV_F64;          // -sp across ranks: float matrix, vectors and exchange (code/main.cpp:1710-1720)
    const size_t vsz = c.sp ? 4 : 8;
    char *d_x = nullptr, *d_y = nullptr;
    HK(hipMalloc((void **)&d_x, vsz * (size_t)vec_len * b));
    HK(hipMalloc((void **)&d_y, vsz * (size_t)vec_len * b));
    HK(hipMemset(d_y, 0, vsz * (size_t)vec_len * b));
    std::vector<char> hx(vsz * (size_t)vec_len * b, 0);
    auto put = [&](std::vector<char> &buf, size_t k, double v) { if (c.sp) ((float *)buf.data())[k] = (float)v; else ((double *)buf.data())[k] = v; };
    auto get = [&](const std::vector<char> &buf, size_t k) -> double { return c.sp ? (double)((const float *)buf.data())[k] : ((const double *)buf.data())[k]; };
    {
        std::vector<char> xo(vsz * (size_t)std::max<int64_t>(n_local, 1)), xp(vsz * (size_t)std::max<int64_t>(n_local, 1));
        const char *xk = getenv("USPMV_DIST_X");
        const bool ramp = xk && !strcmp(xk, "ramp");
        // -rand_x 1 | m: min / max of |values| over the WHOLE matrix (every rank contributes its block; extract_matrix_min_mean_max +
        // MPI_Bcast, code/utilities.hpp:2502-2540), then random_init's default-seeded engine -- the same sequence on every rank (:880-912),
        // one draw per element of the padded local vector
        const double vmean = vmin + (vmax - vmin) / 2.0;
        std::mt19937 engine;
        auto draw = [&]() { return std::fma(std::generate_canonical<double, 53>(engine), vmax - vmin, vmin); };   // (fused like the reference's -O3 -march=native build, see uspmv_main.cpp)
        for (int v = 0; v < b; ++v) {
            for (int64_t i = 0; i < vec_len; ++i) {
                double val = ramp ? (1.0 + 1e-3 * (double)(i % 1000)) * (1.0 + v / 8.0) : 5.0;
                if (c.random_init_x == '1') val = c.sp ? (double)(float)draw() : draw();
                else if (c.random_init_x == 'm') val = vmean;
                if (i < n_local) put(xo, (size_t)i, val);
            }
            CK(uspmv_apply_permutation(xp.data(), xo.data(), n2o, n_local, dtype));
            for (int64_t i = 0; i < n_local; ++i) put(hx, (size_t)(c.layout == USPMV_ROWWISE ? i * b + v : (int64_t)v * vec_len + i), get(xp, (size_t)i));
        }
        HK(hipMemcpy(d_x, hx.data(), hx.size(), hipMemcpyHostToDevice));
    }
    const int comm_halos = c.comm_halos ? 1 : 0;
    auto steps = [&](int n) {
        if (b > 1) for (int k = 0; k < n; ++k) CK(uspmv_dist_spmmv(D, d_x, d_y, b, c.layout, c.vec_mode, comm_halos, st));
        else if (comm_halos && c.use_graph) CK(uspmv_dist_run(D, d_x, d_y, n, 1, st));
        else for (int k = 0; k < n; ++k) CK(uspmv_dist_spmv(D, d_x, d_y, comm_halos, st));
    };

    if (const char *dump = getenv("USPMV_DUMP_Y")) {   // one step, y of the local rows back in original order (copy_back_result), vector after vector
        steps(1);
        HK(hipStreamSynchronize(st));
        std::vector<char> hy(vsz * (size_t)vec_len * b), col(vsz * (size_t)std::max<int64_t>(n_pad, 1)), yo(vsz * (size_t)std::max<int64_t>(n_local * b, 1));
        HK(hipMemcpy(hy.data(), d_y, hy.size(), hipMemcpyDeviceToHost));
        for (int v = 0; v < b; ++v) {
            for (int64_t i = 0; i < n_pad; ++i) put(col, (size_t)i, get(hy, (size_t)(c.layout == USPMV_ROWWISE ? i * b + v : (int64_t)v * vec_len + i)));
            CK(uspmv_apply_permutation(yo.data() + vsz * (size_t)v * n_local, col.data(), o2n, n_local, dtype));
        }
        publish(std::string(dump) + "." + std::to_string(rank), yo.data(), vsz * (size_t)n_local * b);
        // -seg_metis: the blocks are blocks of the PERMUTED matrix; <prefix>.perm (int32 per row: permuted row r = original row perm[r])
        // lets the reader put the gathered y back into the original numbering
        if (!metis_perm.empty()) publish(std::string(dump) + ".perm", metis_perm.data(), 4 * metis_perm.size());
    }

    // ---- solve mode (-mode s): COMM - spmv - SWAP, -rev times (code/main.cpp:528-607; the last swap is undone by reading y), y of the
    //      local rows in original order to <dump_y>.<rank>, the optional bitwise self-check, no timing
    if (c.mode == 's') {
        char *sx = d_x, *sy = d_y;
        for (unsigned long i = 0; i < c.n_repetitions; ++i) {
            CK(uspmv_dist_spmv(D, sx, sy, comm_halos, st));
            if (i + 1 < c.n_repetitions) std::swap(sx, sy);
        }
        HK(hipStreamSynchronize(st));
        if (!c.dump_y.empty()) {
            std::vector<char> hy(vsz * (size_t)std::max<int64_t>(n_pad, 1)), yo(vsz * (size_t)std::max<int64_t>(n_local, 1));
            HK(hipMemcpy(hy.data(), sy, vsz * (size_t)n_pad, hipMemcpyDeviceToHost));
            CK(uspmv_apply_permutation(yo.data(), hy.data(), o2n, n_local, dtype));
            publish(c.dump_y + "." + std::to_string(rank), yo.data(), vsz * (size_t)n_local);
            if (!metis_perm.empty()) publish(c.dump_y + ".perm", metis_perm.data(), 4 * metis_perm.size());
        }
        int64_t mm = -1, tot = -1;
        if (c.check_y && comm_halos) {
            CK(uspmv_dist_check(D, local, wsa.data(), d_x, d_y, 0, st, &mm, nullptr));
            std::vector<int64_t> all((size_t)std::max(P, comm_size), 0);
            CK(uspmv_dist_allgather_i64(D, mm, all.data(), st));
            tot = 0;
            for (int p = 0; p < (meta[8] ? 1 : comm_size); ++p) tot += all[(size_t)p];
        }
        uspmv_coo_free(local);
        if (rank == 0)
            printf("solve mode: %lu revision(s) done on %d ranks%s%s\n", c.n_repetitions, P, meta[8] ? " (loopback)" : "",
                   tot < 0 ? "" : tot == 0 ? ", y checked bitwise on every rank: ok" : ", y CHECK FAILED");
        CK(uspmv_dist_barrier(D, st));
        (void)hipFree(d_x); (void)hipFree(d_y);
        uspmv_dist_free(D);
        (void)hipStreamDestroy(st);
        uspmv_hostcomm_t *hcs = g.hc;
        g.hc = nullptr;
        uspmv_hostcomm_free(hcs);
        return tot > 0 ? 3 : 0;
    }

    // ---- one EAGER step before anything is captured or timed: whoever has to kill a hung run can tell "the exchange itself does not work"
    //      (no point in an eager retry) from "the exchange works, its capture / replay does not" (USPMV_STAGES, bench.py's tiers)
    if (c.mode == 'b' && P > 1 && comm_halos && b == 1) {
        CK(uspmv_dist_spmv(D, d_x, d_y, comm_halos, st));
        HK(hipStreamSynchronize(st));
        stage("first eager step done");
    }

    // ---- the arrangement of the step (-step_form): fixed, or the fastest of the candidates on this machine
    std::string form = c.step_form, form_report;
    const bool legacy_knobs = c.no_overlap || getenv("USPMV_PAD_SPLIT") || getenv("USPMV_FUSED_STEP");
    if (!(P > 1 && comm_halos && b == 1) || legacy_knobs) form = c.no_overlap ? "plain" : "overlap";
    auto apply_form = [&](const std::string &f) -> int {
        if (int rc = uspmv_dist_set_option(D, "overlap", f == "plain" ? 0 : 1)) return rc;
        if (int rc = uspmv_dist_set_option(D, "pad_split", f == "pad" || f == "fused" ? 1 : 0)) return rc;
        return uspmv_dist_set_option(D, "fused_step", f == "fused" ? 1 : 0);
    };
    if (P > 1 && comm_halos && b == 1 && !legacy_knobs) {
        stage("timing the step forms");                       // (the first captured / eager steps run in here)
        if (form == "auto" || form == "auto_all") {
            static const char *names[4] = {"overlap", "plain", "pad", "fused"};
            int best = USPMV_STEP_OVERLAP;
            double tms[4];
            CK(uspmv_dist_set_option(D, "autotune_all", form == "auto_all"));   // (pad / fused: timed on request only, see uspmv_dist_autotune)
            CK(uspmv_dist_autotune(D, d_x, d_y, c.use_graph ? 1 : 0, local, wsa.data(), st, &best, tms));
            for (int k = 0; k < 4; ++k) {
                if (tms[k] == 0) continue;
                char buf[64];
                snprintf(buf, sizeof buf, "%s\"%s\": %.6f", form_report.empty() ? "" : ", ", names[k], std::fabs(tms[k]));
                form_report += buf;
                if (tms[k] < 0) { form_report += ", \"rejected\": 1"; if (rank == 0) fprintf(stderr, "step form %s failed the self-check on this machine: not used\n", names[k]); }
            }
            form = names[best];
            HK(hipMemcpy(d_x, hx.data(), hx.size(), hipMemcpyHostToDevice));
        }
        CK(apply_form(form));
    }

    stage("step form chosen");
    // ---- timed region
    int n_iter = 2;
    double runtime = 0, runtime_other = 0;
    const int warm = c.bench_warmup >= 0 ? c.bench_warmup : 100;
    steps(warm);                                       // WARM_UP_REPS (code/main.cpp:22, :408-419)
    HK(hipStreamSynchronize(st));
    CK(uspmv_dist_barrier(D, st));
    if (c.bench_steps > 0) {   // exactly K steps between barriers, the slowest rank's clock
        n_iter = c.bench_steps;
        HK(hipDeviceSynchronize());
        CK(uspmv_dist_barrier(D, st));
        auto t0 = std::chrono::steady_clock::now();
        steps(n_iter);
        HK(hipStreamSynchronize(st));
        CK(uspmv_dist_barrier(D, st));
        runtime = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        CK(uspmv_dist_allreduce_max(D, &runtime, st));
        if (comm_halos && P > 1) {   // the same K steps under the OTHER per-step barrier setting (reported next to the headline)
            CK(uspmv_dist_set_option(D, "ba_synch", c.ba_synch ? 0 : 1));
            steps(std::min(warm, 10) + 1);
            HK(hipDeviceSynchronize());
            CK(uspmv_dist_barrier(D, st));
            auto t1 = std::chrono::steady_clock::now();
            steps(n_iter);
            HK(hipStreamSynchronize(st));
            CK(uspmv_dist_barrier(D, st));
            runtime_other = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
            CK(uspmv_dist_allreduce_max(D, &runtime_other, st));
            CK(uspmv_dist_set_option(D, "ba_synch", c.ba_synch ? 1 : 0));
        }
    } else {                   // bench loop of the reference (code/main.cpp:449-523): doubling batches, barriers around each batch
        do {
            CK(uspmv_dist_barrier(D, st));
            auto t0 = std::chrono::steady_clock::now();
            steps(n_iter);
            CK(uspmv_dist_barrier(D, st));
            runtime = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            n_iter *= 2;
            CK(uspmv_dist_allreduce_max(D, &runtime, st));   // every rank must take the same decision: the slowest rank's clock
        } while (runtime < c.bench_time);
        n_iter /= 2;
    }
    const double perf = (double)nnz_g * 2.0 * b / (runtime / n_iter) / 1e9;
    stage("timed region done");

    // ---- this rank's kernel alone (interior + boundary without the exchange), HIP events on the step's stream
    double kernel_ms = 0;
    if (b == 1) {
        hipEvent_t e0, e1;
        HK(hipEventCreate(&e0)); HK(hipEventCreate(&e1));
        CK(uspmv_dist_spmv(D, d_x, d_y, 0, st));
        HK(hipEventRecord(e0, st));
        for (int k = 0; k < 20; ++k) CK(uspmv_dist_spmv(D, d_x, d_y, 0, st));
        HK(hipEventRecord(e1, st));
        HK(hipEventSynchronize(e1));
        float ms = 0;
        HK(hipEventElapsedTime(&ms, e0, e1));
        kernel_ms = ms / 20.0;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }

    // ---- self-check (-check_y 1): one step on x_global[j] = 1 + 1e-3 (j mod 1000), bitwise against the block's entry-ordered FMA chains
    int64_t mism = -1, mism_total = -1;
    double checksum = 0;
    if (c.check_y && b == 1 && comm_halos) {
        CK(uspmv_dist_check(D, local, wsa.data(), d_x, d_y, c.use_graph ? 1 : 0, st, &mism, &checksum));
        std::vector<int64_t> all((size_t)std::max(P, comm_size), 0);
        CK(uspmv_dist_allgather_i64(D, mism, all.data(), st));
        mism_total = 0;
        for (int p = 0; p < (meta[8] ? 1 : comm_size); ++p) mism_total += all[(size_t)p];
        HK(hipMemcpy(d_x, hx.data(), hx.size(), hipMemcpyHostToDevice));
        if (mism) fprintf(stderr, "[rank %d] CHECK FAILED: %ld of %ld local rows differ from the entry-ordered FMA chains\n", rank, (long)mism, (long)n_local);
    }
    uspmv_coo_free(local);

    // ---- report
    std::vector<int64_t> halos((size_t)std::max(P, comm_size), 0), sends((size_t)std::max(P, comm_size), 0);
    CK(uspmv_dist_allgather_i64(D, n_halo, halos.data(), st));
    CK(uspmv_dist_allgather_i64(D, n_send, sends.data(), st));
    CK(uspmv_dist_info(D, meta));
    // every rank's row of the report (the reference gathers its per-rank numbers on rank 0 too, code/main.cpp:809-1062)
    const double my_bytes = n_el * (vsz + 4.0) + 8.0 * n_chunks + (double)vsz * b * (n_local + n_halo) + (double)vsz * b * n_pad;
    const int n_rep = meta[8] ? 1 : comm_size;
    const char *rank_keys[9] = {"n_local", "n_halo", "n_send", "interior", "boundary", "n_elements", "nnz", "algorithmic_bytes", "local_kernel_ns"};
    const int64_t rank_vals[9] = {n_local, n_halo, n_send, meta[4], meta[5], n_el, sm[7], (int64_t)my_bytes, (int64_t)(kernel_ms * 1e6)};
    std::vector<std::vector<int64_t>> rank_rows(9, std::vector<int64_t>((size_t)std::max(P, comm_size), 0));
    for (int k = 0; k < 9; ++k) CK(uspmv_dist_allgather_i64(D, rank_vals[k], rank_rows[(size_t)k].data(), st));
    int rccl_nranks = 0;
    CK(uspmv_dist_comm_count(D, &rccl_nranks));
    int ver[4] = {0, 0, 0, 0};
    CK(uspmv_runtime_versions(ver));
    const char *protocol = c.bench_steps > 0 ? "fixed steps between barriers" : "reference bench loop (doubling batches)";
    if (comm_rank == 0) {
        const double bytes = n_el * (vsz + 4.0) + 8.0 * n_chunks + (double)vsz * b * (n_local + n_halo) + (double)vsz * b * n_pad;  // this rank's share
        std::ofstream f("spmv_bench.txt", std::ios::app);
        f << c.matrix_name << " with " << P << " RCCL ranks (one per GPU), halo exchange " << (c.comm_halos ? "on" : "off") << std::endl;
        f << "kernel: scs, block_vec_size: " << b << ", C: " << c.C << " sigma: " << c.sigma << ", data_type: " << (c.sp ? "float" : "double") << ", revisions: " << n_iter
          << ", seg_method: " << (c.seg_metis ? "seg-metis" : c.seg_nnz ? "seg-nnz" : "seg-rows") << ", MPI_mode: " << (b == 1 || c.vec_mode == USPMV_SINGLEVEC ? "singlevec" : c.vec_mode == USPMV_MULTIVEC ? "multivec" : "bulkvec")
          << ", ba_synch: " << (c.ba_synch && c.comm_halos ? 1 : 0) << std::endl << std::endl;
        char buf[256];
        snprintf(buf, sizeof buf, "%-32s%-32s\n%-32s%-32s\n%-32.16g%-32.16g\n\n", "Total Gflops:", "Total Walltime:", "-------------",
                 "-------------", perf, runtime);
        f << buf;
        if (c.verbose || c.print_comm_vol) {   // -print_comm_vol: elements received / sent per rank and step (code/classes_structs.hpp:941)
            f << "Rank Idx:                       Per rank Elems Recvd:           Per rank Elems Sent:\n---------                       -------------                   -------------\n";
            for (int p = 0; p < P; ++p) {
                snprintf(buf, sizeof buf, "%-32d%-32ld%-32ld\n", p, (long)halos[(size_t)p], (long)sends[(size_t)p]);
                f << buf;
            }
            f << std::endl;
        }
        printf("%d ranks%s, n = %ld, nnz = %ld: Total Gflops: %.4f (%d iterations in %.4f s, %.6f ms per SpMV); rank %d: %.1f GB/s algorithmic, "
               "%ld halo elements, %ld interior + %ld boundary %s, %s, ba_synch %d%s\n", P, meta[8] ? " (loopback)" : host_exchange ? " (host-staged exchange)" : "", (long)n_rows_g, (long)nnz_g, perf, n_iter, runtime,
               runtime / n_iter * 1e3, rank, bytes / (runtime / n_iter) / 1e9, (long)n_halo, (long)meta[4], (long)meta[5], meta[6] ? "tiles" : "chunks",
               meta[9] ? "hipGraph replay" : "eager steps", c.ba_synch && c.comm_halos ? 1 : 0,
               mism_total < 0 ? "" : mism_total == 0 ? ", y checked bitwise on every rank: ok" : ", y CHECK FAILED");
        if (P > 1 && comm_halos && b == 1)
            printf("step form: %s%s%s%s\n", form.c_str(), form_report.empty() ? "" : " (ms per step of the candidates, slowest rank: {", form_report.c_str(), form_report.empty() ? "" : "})");
        if (b > 1) {
            int64_t bm[6];
            CK(uspmv_dist_spmmv_info(D, bm));
            printf("block vectors (b = %d): %ld steps in two parts (interior chunks during the exchange), %ld exchange-then-compute; phased block plan: %s (%ld of %ld tiles boundary)\n",
                   b, (long)bm[0], (long)bm[1], bm[2] ? "yes" : "no", (long)bm[4], (long)bm[3]);
        }
        if (!c.json.empty()) {
            std::string per_rank = "[";
            for (int p = 0; p < n_rep; ++p) {
                per_rank += p ? ", {" : "{";
                char kv[96];
                snprintf(kv, sizeof kv, "\"rank\": %d", p);
                per_rank += kv;
                for (int k = 0; k < 8; ++k) { snprintf(kv, sizeof kv, ", \"%s\": %ld", rank_keys[k], (long)rank_rows[(size_t)k][(size_t)p]); per_rank += kv; }
                snprintf(kv, sizeof kv, ", \"local_kernel_ms\": %.6f}", (double)rank_rows[8][(size_t)p] * 1e-6);
                per_rank += kv;
            }
            per_rank += "]";
            char js[3072];
            snprintf(js, sizeof js,
                     "{\"gflops\": %.4f, \"ms_per_step\": %.6f, \"steps\": %d, \"warmup\": %d, \"runtime_s\": %.6f, \"ranks\": %d, \"loopback\": %s, "
                     "\"exchange\": \"%s\", \"n_rows\": %ld, \"nnz\": %ld, \"protocol\": \"%s\", \"ba_synch\": %d, \"graph_replay\": %s, \"graph_launches\": %ld, "
                     "\"eager_steps\": %ld, \"overlap\": %s, \"step_form\": \"%s\", \"step_form_candidates_ms\": {%s}, \"other_ba_synch_ms_per_step\": %.6f, \"y_checked\": %s, \"y_mismatches\": %ld, \"y_checksum_rank0\": %.17g, "
                     "\"rank0\": {\"n_local\": %ld, \"n_halo\": %ld, \"n_send\": %ld, \"interior\": %ld, \"boundary\": %ld, \"tiles\": %s, \"n_elements\": %ld, "
                     "\"n_chunks\": %ld, \"n_rows_padded\": %ld, \"algorithmic_bytes\": %.0f, \"local_kernel_ms\": %.6f}, "
                     "\"rccl_nranks\": %d, \"versions\": {\"hip_build\": %d, \"hip_runtime\": %d, \"rccl_build\": %d, \"rccl_runtime\": %d}",
                     perf, runtime / n_iter * 1e3, n_iter, warm, runtime, P, meta[8] ? "true" : "false", host_exchange ? "host" : "rccl", (long)n_rows_g, (long)nnz_g,
                     protocol, c.ba_synch && c.comm_halos ? 1 : 0, meta[9] ? "true" : "false", (long)meta[10], (long)meta[11], form == "plain" ? "false" : "true", form.c_str(), form_report.c_str(), runtime_other / n_iter * 1e3,
                     mism_total < 0 ? "null" : mism_total == 0 ? "true" : "false", (long)mism_total, checksum, (long)n_local, (long)n_halo, (long)n_send,
                     (long)meta[4], (long)meta[5], meta[6] ? "true" : "false", (long)n_el, (long)n_chunks, (long)n_pad, bytes, kernel_ms,
                     rccl_nranks, ver[0], ver[1], ver[2], ver[3]);
            const std::string full = std::string(js) + ", \"per_rank\": " + per_rank + "}";
            if (c.json == "-") printf("%s\n", full.c_str());
            else { std::ofstream jf(c.json); jf << full << std::endl; }
        }
    }
    stage("report written");
    CK(uspmv_dist_barrier(D, st));
    (void)hipFree(d_x); (void)hipFree(d_y);
    uspmv_dist_free(D);
    (void)hipStreamDestroy(st);
    uspmv_hostcomm_t *hc = g.hc;
    g.hc = nullptr;
    uspmv_hostcomm_free(hc);
    return mism_total > 0 ? 3 : 0;
}
