// Distributed SpMV step on RCCL (xGMI), in C++ behind the C ABI (include/uspmv.h, sections L4a / L4b).
//
// Replaces the per-iteration part of the reference's MPI flow for the single-vector one-precision path:
//   init_local_structs            code/main.cpp:1075-1334      (partition block -> SELL-C-sigma, halo discovery)
//   collect_comm_info             code/mpi_funcs.hpp:1061-1124 (who sends what to whom)
//   init/finalize_halo_exchange   code/classes_structs.hpp:857-995
//   the iteration of bench_spmv   code/main.cpp:458-474        (exchange, then kernel, then the -ba_synch barrier)
// One process per GPU.  Per SpMV: ONE pack kernel over the concatenated send list, ONE grouped ncclSend/ncclRecv with every
// receive landing directly in x[n_local + recv_cumsum[p]] (the reference's halo numbering) on a side stream, the tiles /
// chunks that touch no halo column meanwhile on the caller's stream, the boundary ones after the exchange.  The whole step
// (two streams, two events, three kernels, the RCCL group, the optional barrier) can be captured once into a hipGraph and
// replayed: per step the host then pays one hipGraphLaunch instead of ~12 runtime calls (the strong-scaling regime of
// BASELINE config 5, where a rank's kernel takes < 0.2 ms).
//
// Set-up ("who sends what to whom") runs over a uspmv_transport (host/comm_plan.cpp): RCCL with device staging by default,
// the host communicator when the caller passes one (real processes on a CPU box test exactly this code), identity in loopback.
//
// Loopback (comm_size == 1 and P > 1): this process plays logical rank `rank` of a P-way partition and every neighbour is
// itself -- sends and receives become RCCL self send/recv pairs of the ids it asked for.  With an x that repeats with the
// block height this reproduces the true multi-rank result for the rank's rows; it is how a single-GPU box exercises the
// RCCL calls end to end (tests/test_dist_native_gpu.py).  It needs equal block heights (seg-rows on a divisible size): the
// ids a rank asks block p for index p's rows, and the set-up refuses ids outside this block.
// USPMV_EXCHANGE_HOST: the per-step exchange staged through host memory over the transport -- P real processes may then share
// ONE GPU, which is how the step runs with unequal seg-nnz blocks and asymmetric lists on a single-GPU box.
//
// Round 3 additions in this file: optional arrangements of the step around the exchange ("pad_split": tiles that touch the halo only
// through the reference's padding run before the exchange under a sign / finiteness guard; "fused_step": one launch whose boundary
// tiles look once at an exchange counter and defer themselves otherwise), uspmv_dist_autotune (times the arrangements collectively and
// keeps the fastest, self-checked), the block-vector step in two parts over marked chunk-length arrays (uspmv_dist_spmmv, also with
// the phased block plan built on the rank's matrix), the block exchange staged through the host for real ranks on one GPU.
#include <rccl/rccl.h>

#include <chrono>
#include <string>

#include "uspmv_device.hpp"

struct uspmv_dist {
    int rank = 0, P = 1, comm_rank = 0, comm_size = 1;
    bool loopback = false, overlap = true, tiles = false, owns_setup = false, no_pack = false, ba_synch = false, host_exchange = false;
    bool diag_skip_exchange = false;
    bool autotune_all = false;        // uspmv_dist_autotune also times the speculative arrangements (pad, fused); off: overlap | plain only
    int diag_spmmv_part = 0;          // diagnosis: the two-part block-vector step runs only its interior (1) or boundary (2) part
    int capture_mode = hipStreamCaptureModeRelaxed;
    ncclComm_t comm = nullptr;
    uspmv_transport_t tr{};           // set-up transport (and the per-step one of USPMV_EXCHANGE_HOST)
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_main = nullptr, ev_comm = nullptr;
    uspmv_dmat_t *A = nullptr;
    uspmv_scs_t *scs = nullptr;       // owned when built by uspmv_dist_create_from_coo
    uspmv_halo_t *halo = nullptr;
    uspmv_comm_plan_t *plan = nullptr;
    int dtype = USPMV_F64;
    int64_t n_local = 0, n_halo = 0, n_send = 0, n_int = 0, n_bnd = 0, vec_len = 0, n_rows_padded = 0;
    std::vector<int64_t> send_off, recv_off;
    std::vector<int32_t> recv_counts;
    int32_t *d_send_idxs = nullptr, *d_perm = nullptr, *d_int = nullptr, *d_bnd = nullptr;
    int32_t *d_src = nullptr;         // pack: send[i] = x[d_src[i]], d_src = perm[send_idxs] composed once
    // Padding tiles (tile lists only).  The reference pads chunks with (value +0, column 0); on ranks > 0 column 0 is a halo column
    // (code/mpi_funcs.hpp:279-306), so more than half of a rank's tiles "touch the halo" only through fma(+0, x[pad_col], acc).  Those
    // run with the interior tiles, before the exchange has delivered x[pad_col]: for a finite operand of the same sign the product is
    // the same signed zero and every row's chain comes out bit for bit as with the delivered value.  The step keeps the value the
    // slot held before the exchange, compares it with the delivered one (pad_guard_kernel), and only when either is not finite or the
    // signs differ runs the padding tiles AGAIN after the boundary tiles (a conditional tile list: ids or -1).
    // d_early = interior and padding tiles in ascending order (ONE launch before the exchange completes); d_late = the boundary tiles
    // with real halo references followed by n_pad conditional entries (ONE launch after it): the guard writes ids or -1 there
    int32_t *d_early = nullptr, *d_late = nullptr, *d_pad = nullptr;
    // ONE-launch step ("fused_step" 1, the default with tile lists): d_step = [early | real boundary | padding again (conditional)] in one
    // launch on the side stream; late entries that find the exchange unfinished defer themselves to a second, small launch behind it
    // (uspmv_dev::StepSync -- nothing spins).  Saves the second launch's ramp-up and the join in the common case.
    int32_t *d_step = nullptr, *d_defer = nullptr;
    uspmv_dev::StepSync *d_ss = nullptr;
    uspmv_dev::StepArgs sa;
    bool fused = false, sync_dirty = false, capturing = false;   // (off by default: 0.222 against 0.209 ms, same file)
    int64_t n_bnd_real = 0, n_pad = 0;
    int32_t pad_col = -1;
    bool pad_split = false;           // (off by default: on one GPU it measures 0.221 against 0.209 ms per step, profiles/r03/dist_step_ab.txt)
    void *d_stale = nullptr;
    void *d_send = nullptr;
    void *h_send = nullptr, *h_recv = nullptr;   // pinned staging of USPMV_EXCHANGE_HOST
    int *d_scratch = nullptr;
    // captured step
    hipGraphExec_t gexec = nullptr;
    void *g_x = nullptr, *g_y = nullptr;
    hipStream_t g_stream = nullptr;
    bool graph_failed = false;
    int64_t graph_launches = 0, eager_steps = 0;
    // block-vector exchange plans, one per (b, layout, mode) used so far
    struct BlockPlan {
        int b = 0, layout = 0, mode = 0;
        int32_t *d_src = nullptr;                    // wire order of the send buffer: send[i] = X[d_src[i]]
        std::vector<int32_t *> d_unpack;             // staged receive (bulk, column-wise): per vector, halo slot -> position in d_recv
        void *d_send = nullptr, *d_recv = nullptr;
        void *h_send = nullptr, *h_recv = nullptr;   // pinned staging of USPMV_EXCHANGE_HOST
    };
    std::vector<BlockPlan> block_plans;
    std::vector<int32_t> h_send_idxs, h_perm;        // host copies for building those plans
    // two-part SpMMV (interior chunks while the block-vector exchange runs): see uspmv_dmat::part_len
    bool parts = false;                              // the handle carries the caller's-order arrays
    int64_t plan_b = 0, plan_bnd_tiles = 0;          // block width of the phased plan built through "block_plan", its boundary tiles
    int64_t spmmv_two_part = 0, spmmv_one_part = 0;  // steps taken in either form
};

namespace {

#define NCCL_TRY(call)                                                                                           \
    do {                                                                                                         \
        ncclResult_t r_ = (call);                                                                                \
        if (r_ != ncclSuccess) return uspmv::fail(USPMV_ERR_HIP, "%s failed: %s (%s:%d)", #call, ncclGetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

inline int peer(const uspmv_dist *D, int p) { return D->loopback ? 0 : p; }
inline ncclDataType_t nccl_vt(const uspmv_dist *D) { return D->dtype == USPMV_F64 ? ncclDouble : ncclFloat; }
inline size_t vsize(const uspmv_dist *D) { return D->dtype == USPMV_F64 ? 8 : 4; }

// ---- set-up transports that live in this file: RCCL with device staging, identity for loopback
struct DevBuf {   // scoped device allocation: freed on every return path
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 16)); }
};

int rccl_alltoallv(void *ctx, const void *send, const int64_t *so, void *recv, const int64_t *ro) {
    auto *D = (uspmv_dist *)ctx;
    const int P = D->comm_size;
    const int64_t sb = so[P] - so[0], rb = ro[P] - ro[0];
    DevBuf ds, dr;
    HIP_TRY(ds.alloc((size_t)sb));
    HIP_TRY(dr.alloc((size_t)rb));
    if (sb) HIP_TRY(hipMemcpy(ds.p, (const char *)send + so[0], (size_t)sb, hipMemcpyHostToDevice));
    hipStream_t st = D->side_stream;
    NCCL_TRY(ncclGroupStart());
    for (int q = 0; q < P; ++q) {
        const int64_t ns = so[q + 1] - so[q], nr = ro[q + 1] - ro[q];
        if (nr) NCCL_TRY(ncclRecv((char *)dr.p + (ro[q] - ro[0]), (size_t)nr, ncclInt8, q, D->comm, st));
        if (ns) NCCL_TRY(ncclSend((const char *)ds.p + (so[q] - so[0]), (size_t)ns, ncclInt8, q, D->comm, st));
    }
    NCCL_TRY(ncclGroupEnd());
    HIP_TRY(hipStreamSynchronize(st));
    if (rb) HIP_TRY(hipMemcpy((char *)recv + ro[0], dr.p, (size_t)rb, hipMemcpyDeviceToHost));
    return USPMV_OK;
}
int rccl_allgather(void *ctx, const void *send, void *recv, int64_t bytes) {
    auto *D = (uspmv_dist *)ctx;
    if (bytes == 0) return USPMV_OK;
    DevBuf ds, dr;
    HIP_TRY(ds.alloc((size_t)bytes));
    HIP_TRY(dr.alloc((size_t)bytes * (size_t)D->comm_size));
    HIP_TRY(hipMemcpy(ds.p, send, (size_t)bytes, hipMemcpyHostToDevice));
    NCCL_TRY(ncclAllGather(ds.p, dr.p, (size_t)bytes, ncclInt8, D->comm, D->side_stream));
    HIP_TRY(hipStreamSynchronize(D->side_stream));
    HIP_TRY(hipMemcpy(recv, dr.p, (size_t)bytes * (size_t)D->comm_size, hipMemcpyDeviceToHost));
    return USPMV_OK;
}
int rccl_barrier(void *ctx) {
    auto *D = (uspmv_dist *)ctx;
    NCCL_TRY(ncclAllReduce(D->d_scratch, D->d_scratch, 1, ncclInt32, ncclSum, D->comm, D->side_stream));
    HIP_TRY(hipStreamSynchronize(D->side_stream));
    return USPMV_OK;
}
// loopback: "rank p needs from me" := what I asked p for -- every exchange is the identity on this process
int self_alltoallv(void *ctx, const void *send, const int64_t *so, void *recv, const int64_t *ro) {
    const int P = ((uspmv_dist *)ctx)->P;
    for (int q = 0; q < P; ++q) {
        if (so[q + 1] - so[q] != ro[q + 1] - ro[q]) return uspmv::fail(USPMV_ERR_INVALID, "loopback all-to-all-v: segment lengths differ");
        if (so[q + 1] > so[q]) memmove((char *)recv + ro[q], (const char *)send + so[q], (size_t)(so[q + 1] - so[q]));
    }
    return USPMV_OK;
}
int self_allgather(void *ctx, const void *send, void *recv, int64_t bytes) {
    const int P = ((uspmv_dist *)ctx)->P;
    for (int q = 0; q < P; ++q) memcpy((char *)recv + (size_t)q * (size_t)bytes, send, (size_t)bytes);
    return USPMV_OK;
}
int self_barrier(void *) { return USPMV_OK; }

// ---- the per-step exchange
// pack_send_buf (code/mpi_funcs.hpp:25-33) with the two index levels composed at set-up; thread 0 also keeps the value the padding
// column's slot holds BEFORE the exchange overwrites it (see uspmv_dist::pad_col)
template <typename VT>
__global__ void pack_kernel(VT *__restrict__ out, const VT *__restrict__ x, const int *__restrict__ src, const long n, const int pad_col, VT *__restrict__ stale) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = x[src[i]];
    if (i == 0 && pad_col >= 0) *stale = x[pad_col];
}

// after the exchange: must the padding tiles run again?  (not when the slot held, and now holds, finite values of one sign:
// fma(+0, s, acc) == fma(+0, c, acc) bit for bit then)
template <typename VT>
__global__ void pad_guard_kernel(const VT *__restrict__ x, const int pad_col, const VT *__restrict__ stale, const int *__restrict__ pad_ids,
                                 int *__restrict__ rerun, const long n_pad, int *__restrict__ counter) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const VT s = *stale, c = x[pad_col];
    const bool need = !(isfinite(s) && isfinite(c) && (signbit(s) == signbit(c)));
    if (k < n_pad) rerun[k] = need ? pad_ids[k] : -1;
    if (k == 0 && need) atomicAdd(counter, 1);
}

__global__ void exchange_done_kernel(uspmv_dev::StepSync *ss) { __hip_atomic_store(&ss->flag, ss->flag + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

int sync_reset(uspmv_dist *D) {
    if (!D->d_ss) return USPMV_OK;
    HIP_TRY(hipDeviceSynchronize());
    uspmv_dev::StepSync z{};
    z.expect = 1;
    HIP_TRY(hipMemcpy(D->d_ss, &z, sizeof z, hipMemcpyHostToDevice));
    D->sync_dirty = false;
    return USPMV_OK;
}

inline bool fused_on(const uspmv_dist *D) { return D->fused && D->overlap && D->tiles && D->d_step && D->sa.n_real + D->sa.n_cond > 0; }

inline bool pads_on(const uspmv_dist *D) { return D->pad_split && D->overlap && D->tiles && D->n_pad > 0 && D->pad_col >= 0; }

int pack(uspmv_dist *D, void *d_x, hipStream_t st) {
    const long n = D->no_pack ? 0 : (long)D->n_send;           // (-no_pack 1: a stale buffer travels, timing only)
    const int pc = pads_on(D) ? D->pad_col : -1;
    if (n == 0 && pc < 0) return USPMV_OK;
    const unsigned grid = (unsigned)std::max<long>((n + 255) / 256, 1);
    if (D->dtype == USPMV_F64)
        hipLaunchKernelGGL(pack_kernel<double>, dim3(grid), dim3(256), 0, st, (double *)D->d_send, (const double *)d_x, D->d_src, n, pc, (double *)D->d_stale);
    else
        hipLaunchKernelGGL(pack_kernel<float>, dim3(grid), dim3(256), 0, st, (float *)D->d_send, (const float *)d_x, D->d_src, n, pc, (float *)D->d_stale);
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

int pad_guard(uspmv_dist *D, const void *d_x, hipStream_t st) {
    const unsigned grid = (unsigned)((D->n_pad + 255) / 256);
    if (D->dtype == USPMV_F64)
        hipLaunchKernelGGL(pad_guard_kernel<double>, dim3(grid), dim3(256), 0, st, (const double *)d_x, D->pad_col, (const double *)D->d_stale, D->d_pad, D->d_late + D->n_bnd_real,
                           (long)D->n_pad, D->d_scratch + 2);
    else
        hipLaunchKernelGGL(pad_guard_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)d_x, D->pad_col, (const float *)D->d_stale, D->d_pad, D->d_late + D->n_bnd_real,
                           (long)D->n_pad, D->d_scratch + 2);
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

int exchange_rccl(uspmv_dist *D, void *d_x, hipStream_t st) {
    const size_t vsz = vsize(D);
    if (int rc = pack(D, d_x, st)) return rc;
    if (D->diag_skip_exchange) return USPMV_OK;
    NCCL_TRY(ncclGroupStart());
    for (int p = 0; p < D->P; ++p) {
        const int64_t ns = D->send_off[(size_t)p + 1] - D->send_off[(size_t)p], nr = D->recv_counts[(size_t)p];
        if (nr) NCCL_TRY(ncclRecv((char *)d_x + (size_t)(D->n_local + D->recv_off[(size_t)p]) * vsz, (size_t)nr, nccl_vt(D), peer(D, p), D->comm, st));
        if (ns) NCCL_TRY(ncclSend((const char *)D->d_send + (size_t)D->send_off[(size_t)p] * vsz, (size_t)ns, nccl_vt(D), peer(D, p), D->comm, st));
    }
    NCCL_TRY(ncclGroupEnd());
    return USPMV_OK;
}

// USPMV_EXCHANGE_HOST: pack -> pinned host buffer -> all-to-all-v over the transport -> tail of x.  The host waits for the
// pack (and returns with the upload queued on `st`), so the caller may queue independent device work on another stream first.
int exchange_host(uspmv_dist *D, void *d_x, hipStream_t st) {
    const size_t vsz = vsize(D);
    if (int rc = pack(D, d_x, st)) return rc;
    if (D->n_send) HIP_TRY(hipMemcpyAsync(D->h_send, D->d_send, (size_t)D->n_send * vsz, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    std::vector<int64_t> so((size_t)D->P + 1), ro((size_t)D->P + 1);
    for (int p = 0; p <= D->P; ++p) { so[(size_t)p] = D->send_off[(size_t)p] * (int64_t)vsz; ro[(size_t)p] = D->recv_off[(size_t)p] * (int64_t)vsz; }
    if (int rc = D->tr.alltoallv(D->tr.ctx, D->h_send, so.data(), D->h_recv, ro.data())) return rc;
    if (D->n_halo) HIP_TRY(hipMemcpyAsync((char *)d_x + (size_t)D->n_local * vsz, D->h_recv, (size_t)D->n_halo * vsz, hipMemcpyHostToDevice, st));
    return USPMV_OK;
}

inline int exchange(uspmv_dist *D, void *d_x, hipStream_t st) { return D->host_exchange ? exchange_host(D, d_x, st) : exchange_rccl(D, d_x, st); }

int part(uspmv_dist *D, const int32_t *ids, int64_t n, const void *x, void *y, hipStream_t st) {
    if (n == 0) return USPMV_OK;
    return D->tiles ? uspmv_spmv_tiles(D->A, ids, n, x, y, st) : uspmv_spmv_chunks(D->A, ids, n, x, y, st);
}

// -ba_synch 1: the barrier the reference issues after every iteration (code/main.cpp:467, :417; default on,
// code/classes_structs.hpp:90) as a stream-ordered one-element all-reduce -- no rank's next step starts before every rank has
// finished this one, and the host is not involved (so it is part of the captured graph)
int step_barrier(uspmv_dist *D, hipStream_t main) {
    if (!D->ba_synch || D->P == 1) return USPMV_OK;
    if (D->host_exchange) {
        HIP_TRY(hipStreamSynchronize(main));
        return D->tr.barrier ? D->tr.barrier(D->tr.ctx) : USPMV_OK;
    }
    if (D->diag_skip_exchange) return USPMV_OK;
    NCCL_TRY(ncclAllReduce(D->d_scratch, D->d_scratch + 1, 1, ncclInt32, ncclSum, D->comm, main));
    return USPMV_OK;
}

// The one-launch form of the step below (uspmv_dist::d_step)
int step_fused(uspmv_dist *D, void *d_x, void *d_y, hipStream_t main) {
    if (D->sync_dirty) if (int rc = sync_reset(D)) return rc;
    D->sync_dirty = true;                                            // (cleared when every launch of the step has been issued)
    HIP_TRY(hipEventRecord(D->ev_main, main));
    HIP_TRY(hipStreamWaitEvent(D->side_stream, D->ev_main, 0));
    int rc = D->dtype == USPMV_F64 ? uspmv_dev::launch_spmv_tlc_step<double>(D->A, D->d_step, D->sa, 1, (const double *)d_x, (double *)d_y, D->side_stream)
                                   : uspmv_dev::launch_spmv_tlc_step<float>(D->A, D->d_step, D->sa, 1, (const float *)d_x, (float *)d_y, D->side_stream);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(D->ev_comm, D->side_stream));
    if ((rc = exchange(D, d_x, main))) return rc;                    // (a failure here leaves the counters out of step: sync_dirty stays set)
    hipLaunchKernelGGL(exchange_done_kernel, dim3(1), dim3(1), 0, main, D->d_ss);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamWaitEvent(main, D->ev_comm, 0));                // join: the deferred list is complete
    rc = D->dtype == USPMV_F64 ? uspmv_dev::launch_spmv_tlc_step<double>(D->A, D->d_step, D->sa, 2, (const double *)d_x, (double *)d_y, main)
                               : uspmv_dev::launch_spmv_tlc_step<float>(D->A, D->d_step, D->sa, 2, (const float *)d_x, (float *)d_y, main);
    if (rc) return rc;
    D->sync_dirty = false;
    return step_barrier(D, main);
}

// One SpMV: the interior tiles on the side stream, the exchange meanwhile on the caller's stream, the boundary tiles after both.
// The RCCL group stays on the CALLER's stream on purpose: when the step is captured, that is the capture's origin stream.  Under
// the HIP 7.0 / RCCL 2.26 pair that torch bundles, hipStreamEndCapture crashes if an RCCL p2p group was captured on a stream that
// JOINED the capture through an event, while the same group on the origin stream, and kernels on joined streams, capture fine
// (profiles/r03/graph_capture_diag_before_fix.txt / _after_fix.txt; the system's HIP 7.2 / RCCL 2.27 takes either form).  Same DAG, same overlap.
int step(uspmv_dist *D, void *d_x, void *d_y, hipStream_t main, bool comm_halos) {
    if (D->P == 1 || !comm_halos) return uspmv_spmv(D->A, d_x, d_y, main);
    if (!D->overlap) {
        if (int rc = exchange(D, d_x, main)) return rc;
        if (int rc = uspmv_spmv(D->A, d_x, d_y, main)) return rc;
        return step_barrier(D, main);
    }
    // (not under capture: replayed from a hipGraph the one-launch form measured 0.54 ms against 0.21 ms, profiles/r03/dist_step_ab.txt)
    if (fused_on(D) && !D->capturing && (pads_on(D) || D->n_pad == 0)) return step_fused(D, d_x, d_y, main);
    const bool pads = pads_on(D);                                    // padding tiles run with the interior ones (see uspmv_dist::pad_col)
    HIP_TRY(hipEventRecord(D->ev_main, main));                       // fork: everything queued on `main` so far precedes the interior tiles
    HIP_TRY(hipStreamWaitEvent(D->side_stream, D->ev_main, 0));
    if (int rc = pads ? part(D, D->d_early, D->n_int + D->n_pad, d_x, d_y, D->side_stream) : part(D, D->d_int, D->n_int, d_x, d_y, D->side_stream)) return rc;
    HIP_TRY(hipEventRecord(D->ev_comm, D->side_stream));
    if (int rc = exchange(D, d_x, main)) return rc;                  // pack kernel + grouped send / recv into the tail of x (host mode: blocks the host)
    if (pads) if (int rc = pad_guard(D, d_x, main)) return rc;
    HIP_TRY(hipStreamWaitEvent(main, D->ev_comm, 0));                // join
    // (pads: the conditional entries are switched off by the guard unless x[pad_col] changed sign or is not finite)
    if (int rc = pads ? part(D, D->d_late, D->n_bnd_real + D->n_pad, d_x, d_y, main) : part(D, D->d_bnd, D->n_bnd, d_x, d_y, main)) return rc;
    return step_barrier(D, main);
}

void drop_graph(uspmv_dist *D) {
    if (D->gexec) (void)hipGraphExecDestroy(D->gexec);
    D->gexec = nullptr; D->g_x = D->g_y = nullptr; D->g_stream = nullptr;
}

#define DBG(msg) do { if (getenv("USPMV_VERBOSE")) { fprintf(stderr, "[uspmv] dist capture: %s\n", msg); fflush(stderr); } } while (0)

int capture(uspmv_dist *D, void *d_x, void *d_y, hipStream_t main) {
    drop_graph(D);
    hipGraph_t g = nullptr;
    DBG("begin");
    hipError_t e = hipStreamBeginCapture(main, (hipStreamCaptureMode)D->capture_mode);
    if (e != hipSuccess) { (void)hipGetLastError(); return uspmv::fail(USPMV_ERR_HIP, "uspmv_dist_run: hipStreamBeginCapture: %s", hipGetErrorString(e)); }
    D->capturing = true;
    const int rc = step(D, d_x, d_y, main, true);
    D->capturing = false;
    DBG("step issued");
    e = hipStreamEndCapture(main, &g);
    DBG("end capture");
    if (rc != USPMV_OK || e != hipSuccess || !g) {
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        return rc != USPMV_OK ? rc : uspmv::fail(USPMV_ERR_HIP, "uspmv_dist_run: hipStreamEndCapture: %s", hipGetErrorString(e));
    }
    e = hipGraphInstantiate(&D->gexec, g, nullptr, nullptr, 0);
    DBG("instantiated");
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) { D->gexec = nullptr; (void)hipGetLastError(); return uspmv::fail(USPMV_ERR_HIP, "uspmv_dist_run: hipGraphInstantiate: %s", hipGetErrorString(e)); }
    D->g_x = d_x; D->g_y = d_y; D->g_stream = main;
    return USPMV_OK;
}

}  // namespace

extern "C" {

int uspmv_comm_unique_id(void *id128) {
    if (!id128) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_comm_unique_id: NULL argument");
    static_assert(sizeof(ncclUniqueId) == USPMV_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return USPMV_OK;
}

int uspmv_runtime_versions(int v[4]) {
    if (!v) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_runtime_versions: NULL argument");
    v[0] = HIP_VERSION; v[1] = 0; v[2] = NCCL_VERSION_CODE; v[3] = 0;
    (void)hipRuntimeGetVersion(&v[1]);
    (void)ncclGetVersion(&v[3]);
    return USPMV_OK;
}

void uspmv_dist_free(uspmv_dist_t *D) {
    if (!D) return;
    drop_graph(D);
    for (auto &bp : D->block_plans) {
        (void)hipFree(bp.d_src); (void)hipFree(bp.d_send); (void)hipFree(bp.d_recv);
        if (bp.h_send) (void)hipHostFree(bp.h_send);
        if (bp.h_recv) (void)hipHostFree(bp.h_recv);
        for (int32_t *u : bp.d_unpack) (void)hipFree(u);
    }
    (void)hipFree(D->d_send_idxs); (void)hipFree(D->d_perm); (void)hipFree(D->d_int); (void)hipFree(D->d_bnd); (void)hipFree(D->d_send); (void)hipFree(D->d_scratch);
    (void)hipFree(D->d_src); (void)hipFree(D->d_early); (void)hipFree(D->d_late); (void)hipFree(D->d_pad); (void)hipFree(D->d_stale);
    (void)hipFree(D->d_step); (void)hipFree(D->d_defer); (void)hipFree(D->d_ss);
    if (D->h_send) (void)hipHostFree(D->h_send);
    if (D->h_recv) (void)hipHostFree(D->h_recv);
    if (D->ev_main) (void)hipEventDestroy(D->ev_main);
    if (D->ev_comm) (void)hipEventDestroy(D->ev_comm);
    if (D->side_stream) (void)hipStreamDestroy(D->side_stream);
    if (D->comm) (void)ncclCommDestroy(D->comm);
    uspmv_comm_plan_free(D->plan);
    if (D->owns_setup) { uspmv_dmat_free(D->A); uspmv_halo_free(D->halo); uspmv_scs_free(D->scs); }
    delete D;
}

int uspmv_dist_create_ex(const void *comm_id, int comm_rank, int comm_size, int rank, int P, uspmv_dmat_t *A, const uspmv_halo_t *halo,
                         const int32_t *old_to_new_idx, const int32_t *interior_ids, int64_t n_interior, const int32_t *boundary_ids,
                         int64_t n_boundary, int ids_are_tiles, const uspmv_dist_options_t *opt, uspmv_dist_t **out) {
    const bool host_ex = opt && opt->exchange == USPMV_EXCHANGE_HOST;
    if (opt && opt->exchange != USPMV_EXCHANGE_RCCL && opt->exchange != USPMV_EXCHANGE_HOST)
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_create: unknown exchange %d", opt->exchange);
    if (!A || !halo || !out || P < 1 || rank < 0 || rank >= P || n_interior < 0 || n_boundary < 0 || (n_interior > 0 && !interior_ids) ||
        (n_boundary > 0 && !boundary_ids))
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_create: bad argument");
    if (host_ex) {
        if (!opt->transport || !opt->transport->alltoallv) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_create: USPMV_EXCHANGE_HOST needs a transport");
        comm_rank = rank; comm_size = P;
    } else {
        if (!comm_id || comm_size < 1 || comm_rank < 0 || comm_rank >= comm_size) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_create: bad communicator argument");
        if (!(comm_size == P && comm_rank == rank) && !(comm_size == 1 && comm_rank == 0))
            return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_create: the communicator must have one rank per partition block (comm_size == P) or a single rank (loopback)");
    }
    if (opt && opt->transport && (opt->transport->size != P || opt->transport->rank != rank))
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_create: the transport is rank %d of %d, the block is %d of %d", opt->transport->rank, opt->transport->size, rank, P);
    if (halo->P != P || halo->rank != rank) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_create: halo description belongs to another partition");
    if (halo->n_local > 0 && !old_to_new_idx) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_create: NULL permutation");
    if (int rc = uspmv_dev::check_dmat(A, "uspmv_dist_create")) return rc;
    if (int rc = uspmv_dev::require_device()) return rc;
    auto *D = new uspmv_dist;
    D->rank = rank; D->P = P; D->comm_rank = comm_rank; D->comm_size = comm_size; D->loopback = !host_ex && comm_size == 1 && P > 1;
    D->host_exchange = host_ex;
    D->A = A; D->dtype = A->dtype; D->tiles = ids_are_tiles != 0;
    D->n_local = halo->n_local; D->n_halo = halo->n_halo; D->n_int = n_interior; D->n_bnd = n_boundary;
    D->recv_counts = halo->recv_counts;
    auto bail = [&](int code) { uspmv_dist_free(D); return code; };
#define D_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return bail(uspmv::fail(USPMV_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__)); } while (0)
#define D_NCCL(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return bail(uspmv::fail(USPMV_ERR_HIP, "%s failed: %s (%s:%d)", #call, ncclGetErrorString(r_), __FILE__, __LINE__)); } while (0)
    if (!host_ex) {
        ncclUniqueId id;
        memcpy(&id, comm_id, sizeof id);
        D_NCCL(ncclCommInitRank(&D->comm, comm_size, id, comm_rank));
    }
    D_HIP(hipStreamCreateWithFlags(&D->side_stream, hipStreamNonBlocking));
    D_HIP(hipEventCreateWithFlags(&D->ev_main, hipEventDisableTiming));
    D_HIP(hipEventCreateWithFlags(&D->ev_comm, hipEventDisableTiming));
    D_HIP(hipMalloc((void **)&D->d_scratch, 256));
    D_HIP(hipMemset(D->d_scratch, 0, 256));
    // ---- who sends what to whom (collect_comm_info, code/mpi_funcs.hpp:1061-1124) over the set-up transport
    if (opt && opt->transport) D->tr = *opt->transport;
    else if (D->loopback) { D->tr.ctx = D; D->tr.rank = rank; D->tr.size = P; D->tr.alltoallv = self_alltoallv; D->tr.allgather = self_allgather; D->tr.barrier = self_barrier; }
    else { D->tr.ctx = D; D->tr.rank = comm_rank; D->tr.size = comm_size; D->tr.alltoallv = rccl_alltoallv; D->tr.allgather = rccl_allgather; D->tr.barrier = rccl_barrier; }
    if (int rc = uspmv_comm_plan_create(&D->tr, halo, &D->plan)) {
        if (D->loopback && rc == USPMV_ERR_INVALID) {
            const std::string why = uspmv_last_error();
            rc = uspmv::fail(USPMV_ERR_INVALID, "%s -- loopback needs equal block heights (seg-rows on a size divisible by P); unequal blocks run with real "
                             "ranks (USPMV_EXCHANGE_HOST on one GPU, RCCL on several)", why.c_str());
        }
        return bail(rc);
    }
    D->send_off = D->plan->send_off; D->recv_off = D->plan->recv_off; D->n_send = D->plan->n_send;
    D->h_send_idxs = D->plan->send_idxs;
    D->h_perm.assign(old_to_new_idx, old_to_new_idx + D->n_local);
    // ---- device state of the step
    const size_t vsz = vsize(D);
    D_HIP(hipMalloc((void **)&D->d_send_idxs, 4 * (size_t)std::max<int64_t>(D->n_send, 1)));
    if (D->n_send) D_HIP(hipMemcpy(D->d_send_idxs, D->h_send_idxs.data(), 4 * (size_t)D->n_send, hipMemcpyHostToDevice));
    D_HIP(hipMalloc((void **)&D->d_perm, 4 * (size_t)std::max<int64_t>(D->n_local, 1)));
    if (D->n_local) D_HIP(hipMemcpy(D->d_perm, old_to_new_idx, 4 * (size_t)D->n_local, hipMemcpyHostToDevice));
    {
        std::vector<int32_t> src((size_t)D->n_send);
        for (int64_t i = 0; i < D->n_send; ++i) src[(size_t)i] = old_to_new_idx[D->h_send_idxs[(size_t)i]];     // (ids validated in [0, n_local) by the plan)
        D_HIP(hipMalloc((void **)&D->d_src, 4 * (size_t)std::max<int64_t>(D->n_send, 1)));
        if (D->n_send) D_HIP(hipMemcpy(D->d_src, src.data(), 4 * (size_t)D->n_send, hipMemcpyHostToDevice));
        D_HIP(hipMalloc(&D->d_stale, 16));
        D_HIP(hipMemset(D->d_stale, 0, 16));
    }
    D_HIP(hipMalloc((void **)&D->d_int, 4 * (size_t)std::max<int64_t>(n_interior, 1)));
    D_HIP(hipMalloc((void **)&D->d_bnd, 4 * (size_t)std::max<int64_t>(n_boundary, 1)));
    if (n_interior) D_HIP(hipMemcpy(D->d_int, interior_ids, 4 * (size_t)n_interior, hipMemcpyHostToDevice));
    if (n_boundary) D_HIP(hipMemcpy(D->d_bnd, boundary_ids, 4 * (size_t)n_boundary, hipMemcpyHostToDevice));
    D_HIP(hipMalloc(&D->d_send, vsz * (size_t)std::max<int64_t>(D->n_send, 1)));
    if (host_ex) {
        D_HIP(hipHostMalloc(&D->h_send, vsz * (size_t)std::max<int64_t>(D->n_send, 1), hipHostMallocDefault));
        D_HIP(hipHostMalloc(&D->h_recv, vsz * (size_t)std::max<int64_t>(D->n_halo, 1), hipHostMallocDefault));
    }
    D->n_rows_padded = A->n_chunks * A->C;
    D->vec_len = D->n_local + std::max(D->n_rows_padded - D->n_local, D->n_halo);       // padded_vec_size (code/main.cpp:1406-1412)
    if (P > 1 && !A->alt && A->n_chunks > 0) {
        // the same split for block vectors: per chunk, does it (or its tile) touch a halo column
        std::vector<unsigned char> flag((size_t)A->n_chunks, 0);
        const int64_t cpt = ids_are_tiles ? std::max<int64_t>(A->tlc_tile_rows / A->C, 1) : 1;
        bool ok = true;
        for (int64_t k = 0; k < n_boundary && ok; ++k) {
            const int64_t c0 = (int64_t)boundary_ids[k] * cpt;
            if (boundary_ids[k] < 0 || c0 >= A->n_chunks) { ok = false; break; }
            for (int64_t c = c0; c < std::min(c0 + cpt, A->n_chunks); ++c) flag[(size_t)c] = 1;
        }
        if (!ok) return bail(uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_create: boundary id outside the matrix"));
        if (int rc = uspmv_dev::dmat_part_set_chunks(A, flag.data())) return bail(rc);
        D->parts = true;
    }
#undef D_HIP
#undef D_NCCL
    *out = D;
    return USPMV_OK;
}

int uspmv_dist_create(const void *comm_id, int comm_rank, int comm_size, int rank, int P, uspmv_dmat_t *A, const uspmv_halo_t *halo,
                      const int32_t *old_to_new_idx, const int32_t *interior_ids, int64_t n_interior, const int32_t *boundary_ids,
                      int64_t n_boundary, int ids_are_tiles, uspmv_dist_t **out) {
    return uspmv_dist_create_ex(comm_id, comm_rank, comm_size, rank, P, A, halo, old_to_new_idx, interior_ids, n_interior, boundary_ids, n_boundary,
                                ids_are_tiles, nullptr, out);
}

int uspmv_dist_create_from_coo_ex(const void *comm_id, int comm_rank, int comm_size, int rank, int P, const uspmv_coo_t *local,
                                  const int32_t *wsa, int64_t C, int64_t sigma, int dtype, int tlc, const uspmv_dist_options_t *opt,
                                  uspmv_dist_t **out) {
    if (!local || !wsa || !out) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_create_from_coo: NULL argument");
    if (P < 1 || rank < 0 || rank >= P) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_create_from_coo: bad rank / P");
    // every rank sees the whole work_sharing_arr: a block without rows is refused HERE, by all ranks alike, before the first collective
    // (the conversion of that block would fail on its rank alone and leave the others waiting in the set-up exchange)
    for (int p = 0; p < P; ++p)
        if (wsa[p + 1] <= wsa[p])
            return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_create_from_coo: block %d of %d owns no rows (work_sharing_arr %d .. %d) -- fewer ranks or another "
                               "partition (the reference's seg methods only repair an empty LAST block, code/mpi_funcs.hpp:602-606)", p, P, (int)wsa[p], (int)wsa[p + 1]);
    if (int rc = uspmv_dev::require_device()) return rc;
    uspmv_scs_t *scs = nullptr;
    uspmv_halo_t *halo = nullptr;
    uspmv_dmat_t *A = nullptr;
    int32_t *interior = nullptr, *boundary = nullptr;
    auto cleanup = [&]() { uspmv_dmat_free(A); uspmv_halo_free(halo); uspmv_scs_free(scs); uspmv_free(interior); uspmv_free(boundary); };
    // order of the reference: convert -> halo discovery (rewrites columns) -> column permutation (code/main.cpp:1128, :1271-1308)
    int rc = uspmv_convert_to_scs(local, C, sigma, dtype, nullptr, &scs);
    if (!rc) rc = uspmv_halo_discover(scs, wsa, rank, P, &halo);
    const int32_t *o2n = nullptr;
    if (!rc) rc = uspmv_scs_arrays(scs, nullptr, nullptr, nullptr, nullptr, &o2n, nullptr);
    if (!rc) rc = uspmv_permute_scs_cols(scs, o2n);
    if (!rc) rc = uspmv_dmat_upload(scs, &A);
    int64_t n_tiles = 0, n_staged = 0;
    if (!rc && tlc) {
        // (a rank's block keeps the 256-row tiles unless its lines ask for more: larger tiles measured level on a block of the 304^3 stencil
        //  -- 0.1785 / 0.1791 / 0.1822 ms -- and a 512-row tile = a whole sigma window always holds a padded chunk, i.e. no interior tile is left)
        uspmv_dev::MeasureOff no_measure;                     // (this thread's planner calls only: no global tuning state is flipped)
        rc = uspmv_dmat_optimize(A, scs, 0, &n_tiles, &n_staged);
    }
    int tile_rows = 0;
    if (!rc) rc = uspmv_dmat_tile_rows(A, &tile_rows);
    const int64_t n_local = wsa[rank + 1] - wsa[rank];
    int64_t n_int = 0, n_bnd = 0;
    if (!rc) rc = uspmv_scs_split_chunks(scs, n_local, &interior, &n_int, &boundary, &n_bnd);
    if (rc) { cleanup(); return rc; }
    std::vector<int32_t> ids_int, ids_bnd, ids_pad, ids_real;
    int32_t pad_col = -1;
    const bool use_tiles = tile_rows > 0 && n_staged > 0 && !A->alt;
    if (use_tiles) {   // interior / boundary at tile granularity (a tile = tile_rows/C chunks)
        // three classes per tile: no halo column at all / halo only through the +0.0 padding on one column / real halo references
        std::vector<uint8_t> cls;
        rc = uspmv_scs_classify_chunks(scs, n_local, &cls, &pad_col);
        if (rc) { cleanup(); return rc; }
        const int64_t cpt = tile_rows / C;
        std::vector<uint8_t> tc((size_t)n_tiles, 0);
        for (int64_t c = 0; c < (int64_t)cls.size(); ++c) tc[(size_t)(c / cpt)] = std::max(tc[(size_t)(c / cpt)], cls[(size_t)c]);
        for (int64_t t = 0; t < n_tiles; ++t) {
            if (tc[(size_t)t] == 0) ids_int.push_back((int32_t)t);
            else { ids_bnd.push_back((int32_t)t); (tc[(size_t)t] == 1 ? ids_pad : ids_real).push_back((int32_t)t); }
        }
    } else {
        ids_int.assign(interior, interior + n_int);
        ids_bnd.assign(boundary, boundary + n_bnd);
    }
    uspmv_free(interior); uspmv_free(boundary); interior = boundary = nullptr;
    uspmv_dist_t *D = nullptr;
    rc = uspmv_dist_create_ex(comm_id, comm_rank, comm_size, rank, P, A, halo, o2n, ids_int.data(), (int64_t)ids_int.size(), ids_bnd.data(),
                              (int64_t)ids_bnd.size(), use_tiles ? 1 : 0, opt, &D);
    if (rc) { cleanup(); return rc; }
    if (use_tiles && P > 1) {
        const bool pads = pad_col >= 0 && !ids_pad.empty();
        if (!pads) { ids_real = ids_bnd; ids_pad.clear(); }
        std::vector<int32_t> early(ids_int.size() + ids_pad.size()), late(ids_real.size() + ids_pad.size(), -1), stepl;
        std::merge(ids_int.begin(), ids_int.end(), ids_pad.begin(), ids_pad.end(), early.begin());
        std::copy(ids_real.begin(), ids_real.end(), late.begin());
        // the late entries three quarters into the grid: late enough for the exchange to have completed in the common case, early enough
        // for their slower (system-scope) loads not to be the tail of the launch
        const size_t late0 = early.size() - early.size() / 4;
        stepl.assign(early.begin(), early.begin() + (long)late0);
        stepl.insert(stepl.end(), ids_real.begin(), ids_real.end());
        stepl.insert(stepl.end(), ids_pad.begin(), ids_pad.end());
        stepl.insert(stepl.end(), early.begin() + (long)late0, early.end());
        const size_t n_late = late.size();
        auto up = [](int32_t **d, const std::vector<int32_t> &h) -> hipError_t {
            hipError_t e = hipMalloc((void **)d, 4 * std::max<size_t>(h.size(), 1));
            if (e == hipSuccess && !h.empty()) e = hipMemcpy(*d, h.data(), 4 * h.size(), hipMemcpyHostToDevice);
            return e;
        };
        hipError_t e = up(&D->d_step, stepl);
        if (e == hipSuccess && pads) e = up(&D->d_pad, ids_pad);
        if (e == hipSuccess && pads) e = up(&D->d_early, early);
        if (e == hipSuccess && pads) e = up(&D->d_late, late);
        if (e == hipSuccess) e = hipMalloc((void **)&D->d_defer, 4 * 2 * std::max<size_t>(n_late, 1));
        if (e == hipSuccess) e = hipMalloc((void **)&D->d_ss, sizeof(uspmv_dev::StepSync));
        if (e != hipSuccess) {
            uspmv_dist_free(D); cleanup();
            return uspmv::fail(USPMV_ERR_ALLOC, "uspmv_dist_create_from_coo: %s", hipGetErrorString(e));
        }
        if (pads) { D->n_pad = (int64_t)ids_pad.size(); D->n_bnd_real = (int64_t)ids_real.size(); D->pad_col = pad_col; }
        D->sa.n_early = (long)early.size(); D->sa.n_real = (long)ids_real.size(); D->sa.n_cond = (long)ids_pad.size(); D->sa.defer_cap = (long)std::max<size_t>(n_late, 1);
        D->sa.late0 = (long)late0;
        D->sa.ss = D->d_ss; D->sa.defer = D->d_defer; D->sa.stale = D->d_stale; D->sa.pad_col = pads ? pad_col : -1;
        if ((rc = sync_reset(D))) { uspmv_dist_free(D); cleanup(); return rc; }
    }
    D->owns_setup = true; D->scs = scs; D->halo = halo;
    *out = D;
    return USPMV_OK;
}

int uspmv_dist_create_from_coo(const void *comm_id, int comm_rank, int comm_size, int rank, int P, const uspmv_coo_t *local,
                               const int32_t *wsa, int64_t C, int64_t sigma, int dtype, int tlc, uspmv_dist_t **out) {
    return uspmv_dist_create_from_coo_ex(comm_id, comm_rank, comm_size, rank, P, local, wsa, C, sigma, dtype, tlc, nullptr, out);
}

int uspmv_dist_comm_plan(const uspmv_dist_t *D, int64_t *n_send, const int64_t **send_off, const int32_t **send_idxs, const int64_t **recv_off) {
    if (!D || !D->plan) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_comm_plan: NULL argument");
    return uspmv_comm_plan_meta(D->plan, n_send, send_off, send_idxs, recv_off);
}

int uspmv_dist_set_option(uspmv_dist_t *D, const char *key, int value) {
    if (!D || !key) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_set_option: NULL argument");
    const std::string k(key);
    if (k == "overlap") { if ((value != 0) != D->overlap) drop_graph(D); D->overlap = value != 0; }
    else if (k == "no_pack") { if ((value != 0) != D->no_pack) drop_graph(D); D->no_pack = value != 0; }
    else if (k == "ba_synch") { if ((value != 0) != D->ba_synch) drop_graph(D); D->ba_synch = value != 0; }
    else if (k == "fused_step") { if ((value != 0) != D->fused) drop_graph(D); D->fused = value != 0; }
    else if (k == "pad_split") { if ((value != 0) != D->pad_split) drop_graph(D); D->pad_split = value != 0; }
    else if (k == "autotune_all") D->autotune_all = value != 0;
    else if (k == "diag_spmmv_part") {
        if (value < 0 || value > 2) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_set_option: diag_spmmv_part is 0 (both), 1 (interior only) or 2 (boundary only)");
        D->diag_spmmv_part = value;
    }
    else if (k == "block_plan") {
        // the phased block plan for block vectors of `value` columns on the rank's matrix (64-byte X rows: dp 8, sp 16; other widths
        // and shapes the planner turns down keep the gather kernels) + its interior / boundary tiles.  0: drop the plan.
        if (value < 0) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_set_option: block_plan takes a block width >= 0");
        D->plan_b = 0; D->plan_bnd_tiles = 0;
        int64_t nt = 0, ns = 0;
        if (value == 0) { uspmv_dev::dmat_block_plan_release(D->A); return USPMV_OK; }
        int rc = D->scs ? uspmv_dmat_optimize_block(D->A, D->scs, value, &nt, &ns) : uspmv_dmat_optimize_block_device(D->A, value, &nt, &ns);
        if (rc) return rc;
        if (D->A->pb) {
            D->plan_b = value;
            if (D->parts) rc = uspmv_dev::dmat_part_set_plan(D->A, (long)D->n_local, &D->plan_bnd_tiles);
        }
        return rc;
    }
    else if (k == "capture_mode") {
        if (value < 0 || value > 2) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_set_option: capture_mode is 0 (global), 1 (thread-local) or 2 (relaxed)");
        D->capture_mode = value == 0 ? hipStreamCaptureModeGlobal : value == 1 ? hipStreamCaptureModeThreadLocal : hipStreamCaptureModeRelaxed;
        drop_graph(D); D->graph_failed = false;
    }
    else if (k == "diag_skip_exchange") { drop_graph(D); D->diag_skip_exchange = value != 0; }
    else return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_set_option: unknown key '%s'", key);
    return USPMV_OK;
}

int uspmv_dist_comm_count(const uspmv_dist_t *D, int *n_ranks) {
    if (!D || !n_ranks) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_comm_count: NULL argument");
    *n_ranks = 0;                                                   // (host-staged exchange: no RCCL communicator exists)
    if (D->comm) NCCL_TRY(ncclCommCount(D->comm, n_ranks));
    return USPMV_OK;
}

int uspmv_dist_pad_info(const uspmv_dist_t *D, int64_t meta[4]) {
    if (!D || !meta) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_pad_info: NULL argument");
    int reruns = 0, r2 = 0;
    if (D->d_scratch) HIP_TRY(hipMemcpy(&reruns, D->d_scratch + 2, 4, hipMemcpyDeviceToHost));
    if (D->d_ss) { HIP_TRY(hipMemcpy(&r2, &D->d_ss->reruns, 4, hipMemcpyDeviceToHost)); reruns += r2; }
    meta[0] = D->n_pad; meta[1] = D->n_pad ? D->n_bnd_real : D->n_bnd; meta[2] = D->pad_col; meta[3] = reruns;
    return USPMV_OK;
}

int uspmv_dist_spmmv_info(const uspmv_dist_t *D, int64_t meta[6]) {
    if (!D || !meta) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_spmmv_info: NULL argument");
    meta[0] = D->spmmv_two_part; meta[1] = D->spmmv_one_part; meta[2] = D->A->pb ? D->plan_b : 0; meta[3] = D->A->pb ? D->A->pb_n_tiles : 0;
    meta[4] = D->A->pb ? D->plan_bnd_tiles : 0; meta[5] = D->parts && !D->A->alt;
    return USPMV_OK;
}

int uspmv_dist_info(const uspmv_dist_t *D, int64_t meta[12]) {
    if (!D || !meta) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_info: NULL argument");
    meta[0] = D->n_local; meta[1] = D->n_halo; meta[2] = D->vec_len; meta[3] = D->n_send; meta[4] = D->n_int; meta[5] = D->n_bnd;
    meta[6] = D->tiles; meta[7] = D->n_rows_padded; meta[8] = D->loopback; meta[9] = D->gexec != nullptr; meta[10] = D->graph_launches;
    meta[11] = D->eager_steps;
    return USPMV_OK;
}

int uspmv_dist_parts(const uspmv_dist_t *D, const uspmv_scs_t **scs, const uspmv_dmat_t **A, const uspmv_halo_t **halo) {
    if (!D) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_parts: NULL argument");
    if (scs) *scs = D->scs;
    if (A) *A = D->A;
    if (halo) *halo = D->halo;
    return USPMV_OK;
}

int uspmv_dist_set_overlap(uspmv_dist_t *D, int overlap) {
    if (!D) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_set_overlap: NULL argument");
    if ((overlap != 0) != D->overlap) drop_graph(D);
    D->overlap = overlap != 0;
    return USPMV_OK;
}

int uspmv_dist_set_no_pack(uspmv_dist_t *D, int no_pack) {
    if (!D) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_set_no_pack: NULL argument");
    if ((no_pack != 0) != D->no_pack) drop_graph(D);
    D->no_pack = no_pack != 0;
    return USPMV_OK;
}

int uspmv_dist_spmv(uspmv_dist_t *D, void *d_x, void *d_y, int comm_halos, void *stream) {
    if (!D || !d_x || !d_y) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_spmv: NULL argument");
    ++D->eager_steps;
    return step(D, d_x, d_y, (hipStream_t)stream, comm_halos != 0);
}

int uspmv_dist_run(uspmv_dist_t *D, void *d_x, void *d_y, int n_steps, int use_graph, void *stream) {
    if (!D || !d_x || !d_y || n_steps < 0) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_run: bad argument");
    hipStream_t main = (hipStream_t)stream;
    if (use_graph && D->P > 1 && !D->graph_failed && !D->host_exchange) {   // (the host-staged exchange blocks the host: nothing to capture)
        if (!main) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_run: graph replay needs an explicit (non-default) stream");
        if (!D->gexec || D->g_x != d_x || D->g_y != d_y || D->g_stream != main) {
            // RCCL sets up its p2p channels at the first use of a pair: one eager step before the capture
            if (int rc = step(D, d_x, d_y, main, true)) return rc;
            HIP_TRY(hipStreamSynchronize(main));
            if (capture(D, d_x, d_y, main) != USPMV_OK) {
                D->graph_failed = true;
                if (getenv("USPMV_VERBOSE")) fprintf(stderr, "[uspmv] dist: graph capture unavailable (%s); eager steps\n", uspmv_last_error());
            }
        }
        if (D->gexec) {
            DBG("launching");
            for (int k = 0; k < n_steps; ++k) HIP_TRY(hipGraphLaunch(D->gexec, main));
            DBG("launched");
            D->graph_launches += n_steps;
            return USPMV_OK;
        }
    }
    for (int k = 0; k < n_steps; ++k)
        if (int rc = step(D, d_x, d_y, main, true)) return rc;
    D->eager_steps += n_steps;
    return USPMV_OK;
}

// How the step should be arranged around the exchange depends on what the exchange costs next to the kernel on the machine at hand
// (one GPU cannot tell for real links): time the arrangements and keep the fastest.  Collective: every rank must call it; the ranks
// agree on the slowest rank's clock through one all-reduce per measurement, so all take the same decision.
int uspmv_dist_autotune(uspmv_dist_t *D, void *d_x, void *d_y, int use_graph, const uspmv_coo_t *local, const int32_t *wsa, void *stream, int *form,
                        double ms[4]) {
    if (!D || !d_x || !d_y || !form || !ms) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_autotune: NULL argument");
    for (int k = 0; k < 4; ++k) ms[k] = 0;
    *form = USPMV_STEP_OVERLAP;
    if (D->P < 2) return USPMV_OK;
    hipStream_t st = (hipStream_t)stream;
    auto apply = [&](int f) -> int {
        if (int rc = uspmv_dist_set_option(D, "overlap", f == USPMV_STEP_PLAIN ? 0 : 1)) return rc;
        if (int rc = uspmv_dist_set_option(D, "pad_split", f == USPMV_STEP_PAD || f == USPMV_STEP_FUSED ? 1 : 0)) return rc;
        return uspmv_dist_set_option(D, "fused_step", f == USPMV_STEP_FUSED ? 1 : 0);
    };
    // Candidates: the two arrangements every run has executed (overlap | plain).  The padding-tile and one-launch forms have only ever
    // run next to an exchange on the SAME GPU (loopback / host-staged), where they measured slower; they are timed only on request
    // ("autotune_all" 1, -step_form auto_all) until a run with real peers has shown what they are worth.
    std::vector<int> cand = {USPMV_STEP_OVERLAP, USPMV_STEP_PLAIN};
    if (D->autotune_all) {
        cand.push_back(USPMV_STEP_PAD);
        if (!use_graph || D->host_exchange) cand.push_back(USPMV_STEP_FUSED);      // (a captured step never takes the one-launch form)
    }
    // every collective the measurement uses once before anything is timed (RCCL sets its channels up lazily, 40 ms the first time), and two
    // rounds over the candidates of which the faster counts: the first candidate must not pay for a cold start
    { double warm = 0; if (int rc = uspmv_dist_barrier(D, st)) return rc; if (int rc = uspmv_dist_allreduce_max(D, &warm, st)) return rc; }
    for (int round = 0; round < 2; ++round)
        for (int f : cand) {
            if (int rc = apply(f)) return rc;
            if (int rc = uspmv_dist_run(D, d_x, d_y, 10, use_graph, st)) return rc;
            HIP_TRY(hipStreamSynchronize(st));
            if (int rc = uspmv_dist_barrier(D, st)) return rc;
            const auto t0 = std::chrono::steady_clock::now();
            if (int rc = uspmv_dist_run(D, d_x, d_y, 40, use_graph, st)) return rc;
            HIP_TRY(hipStreamSynchronize(st));
            if (int rc = uspmv_dist_barrier(D, st)) return rc;
            double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 40 * 1e3;
            if (int rc = uspmv_dist_allreduce_max(D, &t, st)) return rc;
            ms[f] = round == 0 ? t : std::min(ms[f], t);
        }
    int best = cand[0];
    for (int f : cand) if (ms[f] < ms[best]) best = f;
    if (int rc = apply(best)) return rc;
    // an arrangement beyond the two plain ones is only kept if one step of it passes the bitwise self-check on every rank (collective: all
    // ranks reach the same verdict); otherwise the faster of overlap / plain takes over and ms[] of the rejected form is negated
    if ((best == USPMV_STEP_PAD || best == USPMV_STEP_FUSED) && local && wsa) {
        int64_t mm = 0;
        DevBuf keep;                                            // (the check runs on its own x: put the caller's local part back afterwards)
        const size_t xb = vsize(D) * (size_t)std::max<int64_t>(D->n_local, 1);
        HIP_TRY(keep.alloc(xb));
        HIP_TRY(hipMemcpyAsync(keep.p, d_x, xb, hipMemcpyDeviceToDevice, st));
        const int rc_check = uspmv_dist_check(D, local, wsa, d_x, d_y, use_graph, stream, &mm, nullptr);
        HIP_TRY(hipMemcpyAsync(d_x, keep.p, xb, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (rc_check) return rc_check;
        std::vector<int64_t> all((size_t)std::max(D->P, D->comm_size), 0);
        if (int rc = uspmv_dist_allgather_i64(D, mm, all.data(), stream)) return rc;
        int64_t tot = 0;
        for (int p = 0; p < (D->loopback ? 1 : D->comm_size); ++p) tot += all[(size_t)p];
        if (tot) {
            ms[best] = -ms[best];
            best = ms[USPMV_STEP_PLAIN] < ms[USPMV_STEP_OVERLAP] ? USPMV_STEP_PLAIN : USPMV_STEP_OVERLAP;
            if (int rc = apply(best)) return rc;
        }
    }
    *form = best;
    return USPMV_OK;
}

int uspmv_dist_barrier(uspmv_dist_t *D, void *stream) {
    if (!D) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_barrier: NULL argument");
    if (D->host_exchange) {
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        return D->tr.barrier ? D->tr.barrier(D->tr.ctx) : USPMV_OK;
    }
    NCCL_TRY(ncclAllReduce(D->d_scratch, D->d_scratch + 1, 1, ncclInt32, ncclSum, D->comm, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return USPMV_OK;
}

int uspmv_dist_allreduce_max(uspmv_dist_t *D, double *value, void *stream) {
    if (!D || !value) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_allreduce_max: NULL argument");
    if (D->host_exchange) {
        if (!D->tr.allgather) return uspmv::fail(USPMV_ERR_UNSUPPORTED, "uspmv_dist_allreduce_max: the transport has no all-gather");
        std::vector<double> all((size_t)D->P, *value);
        if (int rc = D->tr.allgather(D->tr.ctx, value, all.data(), 8)) return rc;
        for (double v : all) *value = std::max(*value, v);
        return USPMV_OK;
    }
    double *d_t = (double *)(D->d_scratch + 16);
    HIP_TRY(hipMemcpy(d_t, value, 8, hipMemcpyHostToDevice));
    NCCL_TRY(ncclAllReduce(d_t, d_t, 1, ncclDouble, ncclMax, D->comm, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipMemcpy(value, d_t, 8, hipMemcpyDeviceToHost));
    return USPMV_OK;
}

int uspmv_dist_allgather_i64(uspmv_dist_t *D, int64_t value, int64_t *all, void *stream) {
    if (!D || !all) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_allgather_i64: NULL argument");
    if (D->loopback) { for (int p = 0; p < D->P; ++p) all[p] = value; return USPMV_OK; }
    if (D->host_exchange) {
        if (!D->tr.allgather) return uspmv::fail(USPMV_ERR_UNSUPPORTED, "uspmv_dist_allgather_i64: the transport has no all-gather");
        return D->tr.allgather(D->tr.ctx, &value, all, 8);
    }
    DevBuf d_v, d_all;
    HIP_TRY(d_v.alloc(8));
    HIP_TRY(d_all.alloc(8 * (size_t)D->comm_size));
    HIP_TRY(hipMemcpy(d_v.p, &value, 8, hipMemcpyHostToDevice));
    NCCL_TRY(ncclAllGather(d_v.p, d_all.p, 1, ncclInt64, D->comm, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipMemcpy(all, d_all.p, 8 * (size_t)D->comm_size, hipMemcpyDeviceToHost));
    return USPMV_OK;
}

// Self-check of the distributed path on the object's own matrix (include/uspmv.h): x_global[j] = 1 + 1e-3 * (j mod 1000), one
// step, y of the local rows compared bitwise with the entry-ordered FMA chains of the block's COO (host/dist_check.cpp).
int uspmv_dist_check(uspmv_dist_t *D, const uspmv_coo_t *local, const int32_t *wsa, void *d_x, void *d_y, int use_graph, void *stream,
                     int64_t *mismatches, double *checksum) {
    if (!D || !local || !wsa || !d_x || !d_y) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_check: NULL argument");
    if (!D->scs) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_check: needs an object made by uspmv_dist_create_from_coo (it owns the row permutation)");
    if (local->n_rows != D->n_local || wsa[D->rank + 1] - wsa[D->rank] != D->n_local)
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_check: the COO block has %ld rows, the object %ld", (long)local->n_rows, (long)D->n_local);
    const size_t vsz = vsize(D);
    const int64_t nl = D->n_local;
    hipStream_t st = (hipStream_t)stream;
    // x in the kernel's (permuted) order: x_perm[k] = x_global[wsa[rank] + new_to_old[k]]; halo tail and padding zero
    std::vector<char> hx((size_t)D->vec_len * vsz, 0), hy((size_t)D->vec_len * vsz);
    const int32_t *n2o = D->scs->new_to_old_idx.data(), *o2n = D->scs->old_to_new_idx.data();
    for (int64_t k = 0; k < nl; ++k) {
        const double v = uspmv::check_x((int64_t)wsa[D->rank] + n2o[k]);
        if (D->dtype == USPMV_F64) ((double *)hx.data())[k] = v; else ((float *)hx.data())[k] = (float)v;
    }
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(d_x, hx.data(), hx.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(d_y, 0xff, (size_t)D->n_rows_padded * vsz));            // NaN pattern: a row nobody writes cannot pass
    if (int rc = uspmv_dist_run(D, d_x, d_y, 1, use_graph, stream)) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(hy.data(), d_y, (size_t)D->n_rows_padded * vsz, hipMemcpyDeviceToHost));
    std::vector<char> ref((size_t)std::max<int64_t>(nl, 1) * vsz);
    if (int rc = uspmv::dist_reference_rows(local, wsa, D->P, D->rank, D->loopback, D->dtype, ref.data())) return rc;
    int64_t bad = 0;
    double sum = 0.0;
    for (int64_t i = 0; i < nl; ++i) {   // copy_back_result: y_orig[i] = y[old_to_new[i]] (code/utilities.hpp:3862)
        const char *got = hy.data() + (size_t)o2n[i] * vsz, *want = ref.data() + (size_t)i * vsz;
        if (memcmp(got, want, vsz) != 0) ++bad;
        sum += D->dtype == USPMV_F64 ? *(const double *)got : (double)*(const float *)got;
    }
    if (mismatches) *mismatches = bad;
    if (checksum) *checksum = sum;
    return USPMV_OK;
}

}  // extern "C"

// ---- block vectors (SpMMV): the halo exchange of b vectors in the reference's three message patterns
//   USPMV_BULKVEC   one message per neighbour carrying all b vectors           (code/classes_structs.hpp:909-924, :971-981)
//   USPMV_MULTIVEC  one message per neighbour and vector, all posted together  (:893-907, :953-960)
//   USPMV_SINGLEVEC one exchange per vector, one after the other               (:875-891, :943-951; the loop of code/mpi_funcs.hpp:35-60)
// Row-wise X ([element][v]): a neighbour's bulk block is exactly the row-wise halo region, so it lands in the tail of X directly.
// Column-wise X: per-vector messages land in the halo region of their column directly; the bulk block ([v][element] per neighbour)
// lands in a staging buffer and b small gathers move it.  Row-wise X with per-vector messages would need strided receives: refused.
namespace {

uspmv_dist::BlockPlan *block_plan(uspmv_dist *D, int b, int layout, int mode) {
    for (auto &bp : D->block_plans)
        if (bp.b == b && bp.layout == layout && bp.mode == mode) return &bp;
    uspmv_dist::BlockPlan bp;
    bp.b = b; bp.layout = layout; bp.mode = mode;
    const int64_t ld = D->vec_len, ns = D->n_send, nh = D->n_halo;
    const size_t vsz = D->dtype == USPMV_F64 ? 8 : 4;
    if ((int64_t)b * std::max(ld, (int64_t)1) > INT32_MAX) { uspmv::fail(USPMV_ERR_OVERFLOW, "uspmv_dist_spmmv: b * padded_vec_size exceeds int32"); return nullptr; }
    std::vector<int32_t> src((size_t)(ns * b));
    if (layout == USPMV_ROWWISE) {                         // wire = [element][v] (bulk only)
        for (int64_t i = 0; i < ns; ++i)
            for (int v = 0; v < b; ++v) src[(size_t)(i * b + v)] = (int32_t)((int64_t)D->h_perm[(size_t)D->h_send_idxs[(size_t)i]] * b + v);
    } else if (mode == USPMV_BULKVEC) {                    // wire = per neighbour [v][element]
        size_t o = 0;
        for (int p = 0; p < D->P; ++p)
            for (int v = 0; v < b; ++v)
                for (int64_t i = D->send_off[(size_t)p]; i < D->send_off[(size_t)p + 1]; ++i)
                    src[o++] = (int32_t)(D->h_perm[(size_t)D->h_send_idxs[(size_t)i]] + (int64_t)v * ld);
    } else {                                               // wire = [v][element over all neighbours]
        for (int v = 0; v < b; ++v)
            for (int64_t i = 0; i < ns; ++i) src[(size_t)(v * ns + i)] = (int32_t)(D->h_perm[(size_t)D->h_send_idxs[(size_t)i]] + (int64_t)v * ld);
    }
    hipError_t e = hipMalloc((void **)&bp.d_src, 4 * std::max<size_t>(src.size(), 1));
    if (e == hipSuccess && !src.empty()) e = hipMemcpy(bp.d_src, src.data(), 4 * src.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&bp.d_send, vsz * std::max<size_t>((size_t)(ns * b), 1));
    if (e == hipSuccess && layout == USPMV_COLWISE && mode == USPMV_BULKVEC) {
        e = hipMalloc(&bp.d_recv, vsz * std::max<size_t>((size_t)(nh * b), 1));
        std::vector<int32_t> un((size_t)nh);
        for (int v = 0; v < b && e == hipSuccess; ++v) {
            for (int p = 0; p < D->P; ++p) {
                const int64_t nr = D->recv_counts[(size_t)p], ro = D->recv_off[(size_t)p];
                for (int64_t k = 0; k < nr; ++k) un[(size_t)(ro + k)] = (int32_t)((int64_t)b * ro + (int64_t)v * nr + k);
            }
            int32_t *d_u = nullptr;
            e = hipMalloc((void **)&d_u, 4 * std::max<size_t>(un.size(), 1));
            if (e == hipSuccess && nh) e = hipMemcpy(d_u, un.data(), 4 * un.size(), hipMemcpyHostToDevice);
            bp.d_unpack.push_back(d_u);
        }
    }
    if (e != hipSuccess) {   // nothing of a half-built plan stays behind (every failed call would leak it again)
        (void)hipFree(bp.d_src); (void)hipFree(bp.d_send); (void)hipFree(bp.d_recv);
        for (int32_t *u : bp.d_unpack) (void)hipFree(u);
        (void)hipGetLastError();
        uspmv::fail(USPMV_ERR_ALLOC, "uspmv_dist_spmmv: %s", hipGetErrorString(e));
        return nullptr;
    }
    D->block_plans.push_back(bp);
    return &D->block_plans.back();
}

int exchange_block(uspmv_dist *D, void *d_X, int b, int layout, int mode, hipStream_t st) {
    uspmv_dist::BlockPlan *bp = block_plan(D, b, layout, mode);
    if (!bp) return USPMV_ERR_ALLOC;
    const size_t vsz = D->dtype == USPMV_F64 ? 8 : 4;
    const int64_t ld = D->vec_len, ns = D->n_send;
    if (ns * b > 0)
        if (int rc = uspmv_apply_permutation_dev(bp->d_send, d_X, bp->d_src, ns * b, D->dtype, st)) return rc;
    char *X = (char *)d_X;
    const char *S = (const char *)bp->d_send;
    if (D->host_exchange) {
        // USPMV_EXCHANGE_HOST: the same wire formats staged through pinned host memory and the transport's all-to-all-v, so that real
        // ranks sharing one GPU exercise the block plans on unequal blocks and asymmetric lists (tests); the host waits for the pack
        const int64_t nh = D->n_halo;
        if (!bp->h_send) {
            HIP_TRY(hipHostMalloc(&bp->h_send, vsz * (size_t)std::max<int64_t>(ns * b, 1), hipHostMallocDefault));
            HIP_TRY(hipHostMalloc(&bp->h_recv, vsz * (size_t)std::max<int64_t>(nh * b, 1), hipHostMallocDefault));
        }
        if (ns * b > 0) HIP_TRY(hipMemcpyAsync(bp->h_send, bp->d_send, (size_t)(ns * b) * vsz, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        std::vector<int64_t> so((size_t)D->P + 1), ro((size_t)D->P + 1);
        if (mode == USPMV_BULKVEC) {                         // one message per neighbour: b * count elements
            for (int p = 0; p <= D->P; ++p) { so[(size_t)p] = D->send_off[(size_t)p] * b * (int64_t)vsz; ro[(size_t)p] = D->recv_off[(size_t)p] * b * (int64_t)vsz; }
            if (int rc = D->tr.alltoallv(D->tr.ctx, bp->h_send, so.data(), bp->h_recv, ro.data())) return rc;
            char *R = layout == USPMV_ROWWISE ? X + (size_t)D->n_local * b * vsz : (char *)bp->d_recv;
            if (nh * b > 0) HIP_TRY(hipMemcpyAsync(R, bp->h_recv, (size_t)(nh * b) * vsz, hipMemcpyHostToDevice, st));
            if (layout == USPMV_COLWISE && nh)
                for (int v = 0; v < b; ++v)
                    if (int rc = uspmv_apply_permutation_dev(X + (size_t)(v * ld + D->n_local) * vsz, bp->d_recv, bp->d_unpack[(size_t)v], nh, D->dtype, st)) return rc;
            return USPMV_OK;
        }
        for (int p = 0; p <= D->P; ++p) { so[(size_t)p] = D->send_off[(size_t)p] * (int64_t)vsz; ro[(size_t)p] = D->recv_off[(size_t)p] * (int64_t)vsz; }
        for (int v = 0; v < b; ++v) {                        // per-vector messages (multivec and singlevec differ in posting order only)
            if (int rc = D->tr.alltoallv(D->tr.ctx, (const char *)bp->h_send + (size_t)(v * ns) * vsz, so.data(), (char *)bp->h_recv + (size_t)(v * nh) * vsz, ro.data())) return rc;
            if (nh) HIP_TRY(hipMemcpyAsync(X + (size_t)(v * ld + D->n_local) * vsz, (const char *)bp->h_recv + (size_t)(v * nh) * vsz, (size_t)nh * vsz, hipMemcpyHostToDevice, st));
        }
        return USPMV_OK;
    }
    auto group = [&](int v0, int v1) -> int {                // per-vector messages of vectors [v0, v1)
        NCCL_TRY(ncclGroupStart());
        for (int v = v0; v < v1; ++v)
            for (int p = 0; p < D->P; ++p) {
                const int64_t nsp = D->send_off[(size_t)p + 1] - D->send_off[(size_t)p], nr = D->recv_counts[(size_t)p];
                if (nr) NCCL_TRY(ncclRecv(X + (size_t)(v * ld + D->n_local + D->recv_off[(size_t)p]) * vsz, (size_t)nr, nccl_vt(D), peer(D, p), D->comm, st));
                if (nsp) NCCL_TRY(ncclSend(S + (size_t)(v * ns + D->send_off[(size_t)p]) * vsz, (size_t)nsp, nccl_vt(D), peer(D, p), D->comm, st));
            }
        NCCL_TRY(ncclGroupEnd());
        return USPMV_OK;
    };
    if (mode == USPMV_SINGLEVEC) {
        for (int v = 0; v < b; ++v) if (int rc = group(v, v + 1)) return rc;
        return USPMV_OK;
    }
    if (mode == USPMV_MULTIVEC) return group(0, b);
    // bulk: one message per neighbour
    char *R = layout == USPMV_ROWWISE ? X + (size_t)D->n_local * b * vsz : (char *)bp->d_recv;
    NCCL_TRY(ncclGroupStart());
    for (int p = 0; p < D->P; ++p) {
        const int64_t nsp = D->send_off[(size_t)p + 1] - D->send_off[(size_t)p], nr = D->recv_counts[(size_t)p];
        if (nr) NCCL_TRY(ncclRecv(R + (size_t)(D->recv_off[(size_t)p] * b) * vsz, (size_t)(nr * b), nccl_vt(D), peer(D, p), D->comm, st));
        if (nsp) NCCL_TRY(ncclSend(S + (size_t)(D->send_off[(size_t)p] * b) * vsz, (size_t)(nsp * b), nccl_vt(D), peer(D, p), D->comm, st));
    }
    NCCL_TRY(ncclGroupEnd());
    if (layout == USPMV_COLWISE && D->n_halo)
        for (int v = 0; v < b; ++v)
            if (int rc = uspmv_apply_permutation_dev(X + (size_t)(v * ld + D->n_local) * vsz, bp->d_recv, bp->d_unpack[(size_t)v], D->n_halo, D->dtype, st)) return rc;
    return USPMV_OK;
}

}  // namespace

extern "C" int uspmv_dist_spmmv(uspmv_dist_t *D, void *d_X, void *d_Y, int b, int layout, int mode, int comm_halos, void *stream) {
    if (!D || !d_X || !d_Y || b < 1) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_spmmv: bad argument");
    if (layout != USPMV_COLWISE && layout != USPMV_ROWWISE) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_spmmv: unknown layout %d", layout);
    if (mode != USPMV_BULKVEC && mode != USPMV_MULTIVEC && mode != USPMV_SINGLEVEC) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dist_spmmv: unknown mode %d", mode);
    if (layout == USPMV_ROWWISE && mode != USPMV_BULKVEC)
        return uspmv::fail(USPMV_ERR_UNSUPPORTED, "uspmv_dist_spmmv: row-wise block vectors travel in one message per neighbour (bulkvec) only");
    hipStream_t main = (hipStream_t)stream;
    if (!(D->P > 1 && comm_halos)) return uspmv_spmmv(D->A, d_X, d_Y, b, D->vec_len, layout, stream);
    // (a handle whose only block plan is the one-list-per-tile one runs that plan's kernels on the whole matrix: they are ahead of the
    //  gather kernels by more than the exchange costs)
    if (!(D->overlap && D->parts && !D->A->alt) || (D->A->bt && !(D->A->pb && D->A->part_len[1][0]))) {
        // exchange first, then the whole matrix (the reference's order, code/mpi_funcs.hpp:25-60)
        if (int rc = exchange_block(D, d_X, b, layout, mode, main)) return rc;
        if (int rc = uspmv_spmmv(D->A, d_X, d_Y, b, D->vec_len, layout, stream)) return rc;
        ++D->spmmv_one_part;
        return step_barrier(D, main);
    }
    // Two parts, as the single-vector step: the chunks (plan tiles) that touch no halo row run on the side stream while the
    // exchange is under way on the caller's stream, the others after it.  A part is the same kernel over chunk lengths in which the
    // other part's chunks are marked (uspmv_dmat::part_len), so every row's FMA chain is what the one-part step computes.
    // Column-wise X: the interior part re-lays out the local rows into the handle's row-major workspace, the boundary part the
    // halo rows once they have arrived.
    uspmv_dmat_t *A = D->A;
    struct PartGuard { uspmv_dmat_t *A; ~PartGuard() { A->part = 0; } } guard{A};
    A->part_split = (long)D->n_local;
    HIP_TRY(hipEventRecord(D->ev_main, main));
    HIP_TRY(hipStreamWaitEvent(D->side_stream, D->ev_main, 0));
    A->part = 1;
    if (D->diag_spmmv_part != 2)
        if (int rc = uspmv_spmmv(A, d_X, d_Y, b, D->vec_len, layout, D->side_stream)) return rc;
    HIP_TRY(hipEventRecord(D->ev_comm, D->side_stream));
    if (int rc = exchange_block(D, d_X, b, layout, mode, main)) return rc;
    HIP_TRY(hipStreamWaitEvent(main, D->ev_comm, 0));
    A->part = 2;
    if (D->diag_spmmv_part != 1)
        if (int rc = uspmv_spmmv(A, d_X, d_Y, b, D->vec_len, layout, main)) return rc;
    ++D->spmmv_two_part;
    return step_barrier(D, main);
}
