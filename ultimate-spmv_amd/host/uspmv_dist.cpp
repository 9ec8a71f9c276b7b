// Multi-rank path of the `uspmv` harness: one process per GPU, halo exchange on RCCL (xGMI).
//
// Replaces, for the SINGLEVEC colwise one-precision path, the reference's MPI flow:
//   init_local_structs            code/main.cpp:1075-1334   (partition, convert, halo discovery)
//   collect_comm_info             code/mpi_funcs.hpp:1061-1124 (who sends what to whom)
//   init/finalize_halo_exchange   code/classes_structs.hpp:857-995 (per-iteration exchange)
//   bench loop with barriers      code/main.cpp:458-474
//
// Launch: any launcher that exports RANK / WORLD_SIZE / LOCAL_RANK (torchrun, srun with a wrapper,
// `for r in ...; do RANK=$r WORLD_SIZE=$N LOCAL_RANK=$r uspmv ... & done`).  The RCCL unique id travels
// through a file under $USPMV_ID_DIR (default /tmp) keyed by $MASTER_PORT / $USPMV_JOB_ID.
//
// Per SpMV: pack kernel over the concatenated send list -> grouped ncclSend/ncclRecv
// with every receive landing directly in x[n_local + recv_cumsum[p]] (the reference's halo numbering)
// on a side stream, overlapped with the interior tiles; boundary tiles after the exchange.
//
// STATUS: exercised on hardware with WORLD_SIZE = 1 only (the development box has one GPU and RCCL
// refuses several ranks per device); the partition / discovery / packing functions it calls are the
// ones tested bit-exactly against the reference, and the same orchestration is tested with 2-8 ranks in
// the Python driver (tests/test_distributed_cpu.py, tests/test_distributed_gpu.py).
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <algorithm>
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

[[noreturn]] void die(int rank, const std::string &msg) {
    fprintf(stderr, "[rank %d] ERROR: %s\n", rank, msg.c_str());
    exit(1);
}
#define CK(call) do { int rc_ = (call); if (rc_ != USPMV_OK) die(D.rank, std::string(#call) + ": " + uspmv_last_error()); } while (0)
#define HK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) die(D.rank, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
#define NK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) die(D.rank, std::string(#call) + ": " + ncclGetErrorString(r_)); } while (0)

int env_int(const char *a, const char *b, int dflt) {
    const char *v = getenv(a);
    if (!v && b) v = getenv(b);
    return v ? atoi(v) : dflt;
}

struct Dist {
    int rank = 0, P = 1, local_rank = 0;
    ncclComm_t comm = nullptr;
    hipStream_t main_stream = nullptr, comm_stream = nullptr;
    hipEvent_t ev_main = nullptr, ev_comm = nullptr;
    int *d_scratch = nullptr;
};

void barrier(Dist &D) {  // MPI_Barrier twin: a tiny all-reduce, then a host wait
    NK(ncclAllReduce(D.d_scratch, D.d_scratch, 1, ncclInt32, ncclSum, D.comm, D.main_stream));
    HK(hipStreamSynchronize(D.main_stream));
}

}  // namespace

bool uspmv_dist_requested() { return env_int("WORLD_SIZE", "USPMV_WORLD_SIZE", 1) > 1 || getenv("USPMV_FORCE_DIST"); }

int uspmv_run_distributed(const DistConfig &c, uspmv_coo_t *total) {
    Dist D;
    D.rank = env_int("RANK", "USPMV_RANK", 0);
    D.P = env_int("WORLD_SIZE", "USPMV_WORLD_SIZE", 1);
    D.local_rank = env_int("LOCAL_RANK", "USPMV_LOCAL_RANK", D.rank);
    int ndev = 0;
    CK(uspmv_device_count(&ndev));
    if (ndev < 1) die(D.rank, "no HIP device visible");
    CK(uspmv_set_device(D.local_rank % ndev));  // device = my_rank % num_devices (code/main.cpp:1838-1842)

    // ---- RCCL bootstrap through a file
    ncclUniqueId id;
    const char *dir = getenv("USPMV_ID_DIR");
    const char *job = getenv("USPMV_JOB_ID") ? getenv("USPMV_JOB_ID") : getenv("MASTER_PORT") ? getenv("MASTER_PORT") : "0";
    const std::string path = std::string(dir ? dir : "/tmp") + "/uspmv_rccl_" + job + ".id";
    if (D.rank == 0) {
        NK(ncclGetUniqueId(&id));
        const std::string tmp = path + ".tmp";
        { std::ofstream f(tmp, std::ios::binary); f.write((const char *)&id, sizeof id); }
        if (rename(tmp.c_str(), path.c_str()) != 0) die(D.rank, "cannot publish the RCCL id at " + path);
    } else {
        for (int tries = 0;; ++tries) {
            std::ifstream f(path, std::ios::binary);
            if (f && f.read((char *)&id, sizeof id)) break;
            if (tries > 6000) die(D.rank, "timed out waiting for " + path);
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
    }
    NK(ncclCommInitRank(&D.comm, D.P, id, D.rank));
    HK(hipStreamCreate(&D.main_stream));
    HK(hipStreamCreate(&D.comm_stream));
    HK(hipEventCreateWithFlags(&D.ev_main, hipEventDisableTiming));
    HK(hipEventCreateWithFlags(&D.ev_comm, hipEventDisableTiming));
    HK(hipMalloc((void **)&D.d_scratch, 64));
    HK(hipMemset(D.d_scratch, 0, 64));
    barrier(D);
    if (D.rank == 0) unlink(path.c_str());

    // ---- partition + local structs (every rank holds the global COO; the reference scatters from rank 0)
    int64_t n_rows_g, n_cols_g, nnz_g;
    CK(uspmv_coo_dims(total, &n_rows_g, &n_cols_g, &nnz_g));
    std::vector<int32_t> wsa((size_t)D.P + 1);
    CK(uspmv_seg_work_sharing_arr(total, c.seg_nnz ? USPMV_SEG_NNZ : USPMV_SEG_ROWS, D.P, wsa.data()));
    uspmv_coo_t *local = nullptr;
    CK(uspmv_seg_local_coo(total, wsa.data(), D.rank, &local));
    uspmv_scs_t *scs = nullptr;
    CK(uspmv_convert_to_scs(local, c.C, c.sigma, USPMV_F64, nullptr, &scs));
    uspmv_halo_t *halo = nullptr;
    CK(uspmv_halo_discover(scs, wsa.data(), D.rank, D.P, &halo));
    const int32_t *o2n, *n2o;
    CK(uspmv_scs_arrays(scs, nullptr, nullptr, nullptr, nullptr, &o2n, &n2o));
    CK(uspmv_permute_scs_cols(scs, o2n));
    int64_t meta[8];
    CK(uspmv_scs_meta(scs, meta));
    const int64_t n_local = wsa[(size_t)D.rank + 1] - wsa[(size_t)D.rank], n_pad = meta[4], n_chunks = meta[5], n_el = meta[6];
    int64_t n_halo;
    const int32_t *recv_cumsum, *recv_idxs, *recv_counts;
    CK(uspmv_halo_meta(halo, &n_halo, &recv_cumsum, &recv_idxs, &recv_counts));
    const int64_t vec_len = n_local + std::max(n_pad - n_local, n_halo);  // padded_vec_size (code/main.cpp:1406-1412)

    // ---- who sends what to whom: all-gather of the recv counts, then grouped index send/recv
    int32_t *d_counts_all = nullptr, *d_counts = nullptr;
    HK(hipMalloc((void **)&d_counts_all, sizeof(int32_t) * (size_t)D.P * D.P));
    HK(hipMalloc((void **)&d_counts, sizeof(int32_t) * (size_t)D.P));
    HK(hipMemcpy(d_counts, recv_counts, sizeof(int32_t) * (size_t)D.P, hipMemcpyHostToDevice));
    NK(ncclAllGather(d_counts, d_counts_all, (size_t)D.P, ncclInt32, D.comm, D.main_stream));
    HK(hipStreamSynchronize(D.main_stream));
    std::vector<int32_t> counts_all((size_t)D.P * D.P);
    HK(hipMemcpy(counts_all.data(), d_counts_all, sizeof(int32_t) * counts_all.size(), hipMemcpyDeviceToHost));
    std::vector<int64_t> send_off((size_t)D.P + 1, 0), recv_off((size_t)D.P + 1, 0);
    for (int p = 0; p < D.P; ++p) {
        send_off[(size_t)p + 1] = send_off[(size_t)p] + counts_all[(size_t)p * D.P + D.rank];  // what p needs from me
        recv_off[(size_t)p + 1] = recv_off[(size_t)p] + recv_counts[p];
    }
    const int64_t n_send = send_off[(size_t)D.P];
    int32_t *d_recv_idxs = nullptr, *d_send_idxs = nullptr;
    HK(hipMalloc((void **)&d_recv_idxs, sizeof(int32_t) * (size_t)std::max<int64_t>(n_halo, 1)));
    HK(hipMalloc((void **)&d_send_idxs, sizeof(int32_t) * (size_t)std::max<int64_t>(n_send, 1)));
    HK(hipMemcpy(d_recv_idxs, recv_idxs, sizeof(int32_t) * (size_t)n_halo, hipMemcpyHostToDevice));
    NK(ncclGroupStart());
    for (int p = 0; p < D.P; ++p) {
        const int64_t ns = send_off[(size_t)p + 1] - send_off[(size_t)p], nr = recv_counts[p];
        if (nr) NK(ncclSend(d_recv_idxs + recv_off[(size_t)p], (size_t)nr, ncclInt32, p, D.comm, D.main_stream));
        if (ns) NK(ncclRecv(d_send_idxs + send_off[(size_t)p], (size_t)ns, ncclInt32, p, D.comm, D.main_stream));
    }
    NK(ncclGroupEnd());
    HK(hipStreamSynchronize(D.main_stream));

    // ---- device state
    uspmv_dmat_t *A = nullptr;
    CK(uspmv_dmat_upload(scs, &A));
    int64_t n_tiles = 0, n_staged = 0;
    if (c.tlc) CK(uspmv_dmat_optimize(A, scs, 0, &n_tiles, &n_staged));
    int tile_rows = 0;
    CK(uspmv_dmat_tile_rows(A, &tile_rows));
    int32_t *interior = nullptr, *boundary = nullptr;
    int64_t n_int = 0, n_bnd = 0;
    CK(uspmv_scs_split_chunks(scs, n_local, &interior, &n_int, &boundary, &n_bnd));
    const bool use_tiles = tile_rows > 0 && n_staged > 0;
    std::vector<int32_t> ids_int, ids_bnd;
    if (use_tiles) {  // interior / boundary at tile granularity
        const int64_t cpt = tile_rows / c.C;
        std::vector<char> is_b((size_t)n_tiles, 0);
        for (int64_t k = 0; k < n_bnd; ++k) is_b[(size_t)(boundary[k] / cpt)] = 1;
        for (int64_t t = 0; t < n_tiles; ++t) (is_b[(size_t)t] ? ids_bnd : ids_int).push_back((int32_t)t);
    } else {
        ids_int.assign(interior, interior + n_int);
        ids_bnd.assign(boundary, boundary + n_bnd);
    }
    uspmv_free(interior); uspmv_free(boundary);
    int32_t *d_int = nullptr, *d_bnd = nullptr, *d_perm = nullptr;
    HK(hipMalloc((void **)&d_int, sizeof(int32_t) * std::max<size_t>(ids_int.size(), 1)));
    HK(hipMalloc((void **)&d_bnd, sizeof(int32_t) * std::max<size_t>(ids_bnd.size(), 1)));
    HK(hipMalloc((void **)&d_perm, sizeof(int32_t) * (size_t)std::max<int64_t>(n_local, 1)));
    HK(hipMemcpy(d_int, ids_int.data(), sizeof(int32_t) * ids_int.size(), hipMemcpyHostToDevice));
    HK(hipMemcpy(d_bnd, ids_bnd.data(), sizeof(int32_t) * ids_bnd.size(), hipMemcpyHostToDevice));
    HK(hipMemcpy(d_perm, o2n, sizeof(int32_t) * (size_t)n_local, hipMemcpyHostToDevice));
    double *d_x = nullptr, *d_y = nullptr, *d_send = nullptr;
    HK(hipMalloc((void **)&d_x, sizeof(double) * (size_t)vec_len));
    HK(hipMalloc((void **)&d_y, sizeof(double) * (size_t)vec_len));
    HK(hipMalloc((void **)&d_send, sizeof(double) * (size_t)std::max<int64_t>(n_send, 1)));
    HK(hipMemset(d_y, 0, sizeof(double) * (size_t)vec_len));
    {   // x = DefaultValues::x = 5.0 on the local rows (permutation of a constant is the constant), 0 elsewhere
        std::vector<double> hx((size_t)vec_len, 0.0);
        std::fill(hx.begin(), hx.begin() + n_local, 5.0);
        HK(hipMemcpy(d_x, hx.data(), sizeof(double) * hx.size(), hipMemcpyHostToDevice));
    }

    auto step = [&]() {
        if (c.comm_halos && D.P > 1) {
            HK(hipEventRecord(D.ev_main, D.main_stream));
            HK(hipStreamWaitEvent(D.comm_stream, D.ev_main, 0));
            CK(uspmv_pack_send_buf(d_x, d_perm, d_send_idxs, n_send, 0, d_send, USPMV_F64, D.comm_stream));
            NK(ncclGroupStart());
            for (int p = 0; p < D.P; ++p) {
                const int64_t ns = send_off[(size_t)p + 1] - send_off[(size_t)p], nr = recv_counts[p];
                if (nr) NK(ncclRecv(d_x + n_local + recv_cumsum[p], (size_t)nr, ncclDouble, p, D.comm, D.comm_stream));
                if (ns) NK(ncclSend(d_send + send_off[(size_t)p], (size_t)ns, ncclDouble, p, D.comm, D.comm_stream));
            }
            NK(ncclGroupEnd());
            HK(hipEventRecord(D.ev_comm, D.comm_stream));
            if (use_tiles) CK(uspmv_spmv_tiles(A, d_int, (int64_t)ids_int.size(), d_x, d_y, D.main_stream));
            else CK(uspmv_spmv_chunks(A, d_int, (int64_t)ids_int.size(), d_x, d_y, D.main_stream));
            HK(hipStreamWaitEvent(D.main_stream, D.ev_comm, 0));
            if (use_tiles) CK(uspmv_spmv_tiles(A, d_bnd, (int64_t)ids_bnd.size(), d_x, d_y, D.main_stream));
            else CK(uspmv_spmv_chunks(A, d_bnd, (int64_t)ids_bnd.size(), d_x, d_y, D.main_stream));
        } else {
            CK(uspmv_spmv(A, d_x, d_y, D.main_stream));
        }
        if (c.ba_synch && D.P > 1) NK(ncclAllReduce(D.d_scratch, D.d_scratch, 1, ncclInt32, ncclSum, D.comm, D.main_stream));
    };

    // ---- bench loop (code/main.cpp:408-474): 100 warm-ups, doubling batches, barriers around each batch
    for (int k = 0; k < 100; ++k) step();
    barrier(D);
    int n_iter = 2;
    double runtime = 0;
    do {
        barrier(D);
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < n_iter; ++k) step();
        barrier(D);
        runtime = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        n_iter *= 2;
        // every rank must take the same decision: agree on the slowest rank's clock
        double *d_t = (double *)(D.d_scratch + 8);
        HK(hipMemcpy(d_t, &runtime, sizeof(double), hipMemcpyHostToDevice));
        NK(ncclAllReduce(d_t, d_t, 1, ncclDouble, ncclMax, D.comm, D.main_stream));
        HK(hipStreamSynchronize(D.main_stream));
        HK(hipMemcpy(&runtime, d_t, sizeof(double), hipMemcpyDeviceToHost));
    } while (runtime < c.bench_time);
    n_iter /= 2;
    const double perf = (double)nnz_g * 2.0 / (runtime / n_iter) / 1e9;

    // ---- report
    std::vector<int32_t> halos((size_t)D.P, 0);
    {
        int32_t h = (int32_t)n_halo;
        HK(hipMemcpy(d_counts, &h, sizeof(int32_t), hipMemcpyHostToDevice));
        NK(ncclAllGather(d_counts, d_counts_all, 1, ncclInt32, D.comm, D.main_stream));
        HK(hipStreamSynchronize(D.main_stream));
        HK(hipMemcpy(halos.data(), d_counts_all, sizeof(int32_t) * (size_t)D.P, hipMemcpyDeviceToHost));
    }
    if (D.rank == 0) {
        const double bytes = n_el * 12.0 + 8.0 * n_chunks + 8.0 * (n_local + n_halo) + 8.0 * n_pad;  // rank 0's share
        std::ofstream f("spmv_bench.txt", std::ios::app);
        f << c.matrix_name << " with " << D.P << " RCCL ranks (one per GPU), halo exchange " << (c.comm_halos ? "on" : "off") << std::endl;
        f << "kernel: scs, block_vec_size: 1, C: " << c.C << " sigma: " << c.sigma << ", data_type: double, revisions: " << n_iter
          << ", seg_method: " << (c.seg_nnz ? "seg-nnz" : "seg-rows") << ", MPI_mode: singlevec" << std::endl << std::endl;
        char buf[256];
        snprintf(buf, sizeof buf, "%-32s%-32s\n%-32s%-32s\n%-32.16g%-32.16g\n\n", "Total Gflops:", "Total Walltime:", "-------------",
                 "-------------", perf, runtime);
        f << buf;
        if (c.verbose) {
            f << "Rank Idx:                       Per rank Elems Recvd:\n---------                       -------------\n";
            for (int p = 0; p < D.P; ++p) f << p << "                               " << halos[(size_t)p] << "\n";
            f << std::endl;
        }
        printf("%d ranks, n = %ld, nnz = %ld: Total Gflops: %.4f (%d iterations in %.4f s, %.6f ms per SpMV); rank 0: %.1f GB/s algorithmic, "
               "%ld halo elements, %zu interior + %zu boundary %s\n", D.P, (long)n_rows_g, (long)nnz_g, perf, n_iter, runtime,
               runtime / n_iter * 1e3, bytes / (runtime / n_iter) / 1e9, (long)n_halo, ids_int.size(), ids_bnd.size(), use_tiles ? "tiles" : "chunks");
    }
    barrier(D);
    uspmv_dmat_free(A); uspmv_halo_free(halo); uspmv_scs_free(scs); uspmv_coo_free(local);
    (void)hipFree(d_x); (void)hipFree(d_y); (void)hipFree(d_send); (void)hipFree(d_int); (void)hipFree(d_bnd); (void)hipFree(d_perm);
    (void)hipFree(d_recv_idxs); (void)hipFree(d_send_idxs); (void)hipFree(d_counts); (void)hipFree(d_counts_all); (void)hipFree(D.d_scratch);
    ncclCommDestroy(D.comm);
    return 0;
}
