// Host communicator of one node: the ranks of a job (one process per GPU) meet in ONE memory-mapped segment and run the few
// collectives the SET-UP of the distributed SpMV needs -- barrier, broadcast, all-gather, all-to-all-v.  It stands where the
// reference has MPI_Bcast / MPI_Allgather / the index all-to-all of collect_comm_info (code/mpi_funcs.hpp:143-171, :190-196,
// :732-736) and, unlike RCCL, needs no GPU: the same C++ set-up code is driven by real processes on a CPU-only box
// (tests/test_dist_setup_mp.py), by several ranks sharing ONE GPU (USPMV_EXCHANGE_HOST, tests/test_dist_native_gpu.py) and by
// the `uspmv` harness for its rank hand-off (RCCL id, partition, per-rank blocks).  The per-step halo exchange of a
// production run stays on RCCL over xGMI (csrc/uspmv_dist_api.hip).
//
// Rendezvous: rank 0 creates <dir>/uspmv_hc_<job>.tmp.<pid>, initialises it and renames it into place (atomic: a leftover of
// a crashed job with the same key is replaced, never read half-written); the other ranks open it, check magic / size / that
// the creating process is alive, and take a seat.  Once every seat is taken rank 0 unlinks the name: nothing stays behind,
// whatever happens later.  Every wait has a deadline and a shared `failed` flag, so a dead peer yields an error, not a hang.
#include <fcntl.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstring>
#include <new>
#include <string>

#include "uspmv_internal.hpp"

namespace {

constexpr uint64_t HC_MAGIC = 0x5553504d56484331ull;  // "USPMVHC1"
constexpr uint32_t HC_VERSION = 2;

struct Shared {
    uint64_t magic;
    uint32_t version;
    int32_t size;
    int64_t creator_pid;
    uint64_t nonce;
    int64_t slot_bytes, total_bytes;
    std::atomic<int32_t> seats;        // ranks > 0 that have attached
    std::atomic<int32_t> bar_count, bar_gen;
    std::atomic<int32_t> failed;       // a rank that gave up: everybody else returns an error at once
    char pad[64];
};
static_assert(std::atomic<int32_t>::is_always_lock_free, "shared atomics must be lock-free");

double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
void nap(long ns) {
    timespec ts{0, ns};
    nanosleep(&ts, nullptr);
}

}  // namespace

struct uspmv_hostcomm {
    int rank = 0, size = 1;
    double timeout_s = 300.0;
    std::string path;
    Shared *sh = nullptr;
    size_t map_bytes = 0;
    bool unlinked = false;
    // layout behind the header
    int64_t *table(int r) const { return (int64_t *)((char *)sh + sizeof(Shared)) + (size_t)r * ((size_t)size + 1); }
    int64_t *scalar(int r) const { return (int64_t *)((char *)sh + sizeof(Shared)) + (size_t)size * ((size_t)size + 1) + (size_t)r; }
    char *slot(int r) const { return (char *)sh + data_off() + (size_t)r * (size_t)sh->slot_bytes; }
    size_t data_off() const {
        size_t o = sizeof(Shared) + 8 * ((size_t)size * ((size_t)size + 1) + (size_t)size);
        return (o + 255) & ~(size_t)255;
    }
    static size_t data_off_for(int size) {
        size_t o = sizeof(Shared) + 8 * ((size_t)size * ((size_t)size + 1) + (size_t)size);
        return (o + 255) & ~(size_t)255;
    }
};

namespace {

int hc_fail(uspmv_hostcomm *h, const char *what) {
    if (h && h->sh) h->sh->failed.store(1);
    return uspmv::fail(USPMV_ERR_COMM, "host communicator (rank %d of %d): %s", h ? h->rank : -1, h ? h->size : 0, what);
}

int hc_barrier(uspmv_hostcomm *h) {
    if (h->size == 1) return USPMV_OK;
    Shared *s = h->sh;
    if (s->failed.load()) return hc_fail(h, "a peer reported a failure");
    const int gen = s->bar_gen.load();
    if (s->bar_count.fetch_add(1) + 1 == h->size) {
        s->bar_count.store(0);
        s->bar_gen.fetch_add(1);
        return USPMV_OK;
    }
    const double deadline = now_s() + h->timeout_s;
    for (long spin = 0;; ++spin) {
        if (s->bar_gen.load() != gen) return USPMV_OK;
        if (s->failed.load()) return hc_fail(h, "a peer reported a failure");
        if (spin < 2000) sched_yield();
        else {
            nap(spin < 20000 ? 20000 : 500000);
            if (now_s() > deadline) return hc_fail(h, "timed out in a barrier (a peer died or never arrived)");
        }
    }
}

bool pid_alive(int64_t pid) { return pid > 0 && (kill((pid_t)pid, 0) == 0 || errno == EPERM); }

}  // namespace

extern "C" {

int uspmv_hostcomm_create(const char *job, int rank, int size, double timeout_s, uspmv_hostcomm_t **out) {
    if (!job || !out || size < 1 || rank < 0 || rank >= size) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_hostcomm_create: bad argument");
    auto *h = new uspmv_hostcomm;
    h->rank = rank; h->size = size;
    if (timeout_s > 0) h->timeout_s = timeout_s;
    const char *dir = getenv("USPMV_HC_DIR");
    struct stat sb;
    if (!dir) dir = (stat("/dev/shm", &sb) == 0 && S_ISDIR(sb.st_mode) && access("/dev/shm", W_OK) == 0) ? "/dev/shm" : "/tmp";
    std::string key(job);
    for (char &c : key) if (!(isalnum((unsigned char)c) || c == '_' || c == '-' || c == '.')) c = '_';
    h->path = std::string(dir) + "/uspmv_hc_" + key;
    int64_t slot = 4 << 20;
    if (const char *e = getenv("USPMV_HC_SLOT_BYTES")) slot = std::max<int64_t>(64, atoll(e));
    slot = (slot + 63) & ~(int64_t)63;
    const size_t total = uspmv_hostcomm::data_off_for(size) + (size_t)size * (size_t)slot;
    auto bail = [&](int fd, const char *what) {
        if (fd >= 0) close(fd);
        int rc = uspmv::fail(USPMV_ERR_COMM, "uspmv_hostcomm_create (rank %d of %d, %s): %s: %s", rank, size, h->path.c_str(), what, strerror(errno));
        delete h;
        return rc;
    };
    if (rank == 0) {
        const std::string tmp = h->path + ".tmp." + std::to_string((long)getpid());
        unlink(tmp.c_str());
        int fd = open(tmp.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) return bail(-1, "cannot create the segment");
        if (ftruncate(fd, (off_t)total) != 0) { unlink(tmp.c_str()); return bail(fd, "ftruncate"); }
        void *m = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) { unlink(tmp.c_str()); return bail(fd, "mmap"); }
        close(fd);
        memset(m, 0, uspmv_hostcomm::data_off_for(size));
        Shared *s = new (m) Shared;
        s->version = HC_VERSION; s->size = size; s->creator_pid = (int64_t)getpid();
        s->slot_bytes = slot; s->total_bytes = (int64_t)total;
        uint64_t nonce = 0;
        if (int rfd = open("/dev/urandom", O_RDONLY); rfd >= 0) { if (read(rfd, &nonce, 8) != 8) nonce = 0; close(rfd); }
        if (!nonce) nonce = ((uint64_t)getpid() << 32) ^ (uint64_t)(now_s() * 1e6);
        s->nonce = nonce;
        s->seats.store(0); s->bar_count.store(0); s->bar_gen.store(0); s->failed.store(0);
        std::atomic_thread_fence(std::memory_order_seq_cst);
        s->magic = HC_MAGIC;
        h->sh = s; h->map_bytes = total;
        if (rename(tmp.c_str(), h->path.c_str()) != 0) {   // atomically replaces a leftover of a dead job with the same key
            unlink(tmp.c_str()); munmap(m, total); h->sh = nullptr;
            return bail(-1, "cannot publish the segment");
        }
    } else {
        const double deadline = now_s() + h->timeout_s;
        for (;;) {
            int fd = open(h->path.c_str(), O_RDWR);
            if (fd >= 0) {
                struct stat st;
                if (fstat(fd, &st) == 0 && (size_t)st.st_size == total) {
                    void *m = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                    if (m != MAP_FAILED) {
                        Shared *s = (Shared *)m;
                        // a segment left behind by a crashed job fails one of these and is ignored until rank 0 replaces it
                        if (s->magic == HC_MAGIC && s->version == HC_VERSION && s->size == size && s->slot_bytes == slot && pid_alive(s->creator_pid) &&
                            !s->failed.load() && s->seats.fetch_add(1) < size - 1) {
                            close(fd);
                            h->sh = s; h->map_bytes = total;
                            break;
                        }
                        munmap(m, total);
                    }
                }
                close(fd);
            }
            if (now_s() > deadline) { errno = ETIMEDOUT; return bail(-1, "rank 0's segment never appeared"); }
            nap(2000000);
        }
    }
    // everyone seated -> rank 0 removes the name; the mapping lives on
    if (int rc = hc_barrier(h)) { uspmv_hostcomm_free(h); return rc; }
    if (rank == 0) { unlink(h->path.c_str()); h->unlinked = true; }
    *out = h;
    return USPMV_OK;
}

void uspmv_hostcomm_free(uspmv_hostcomm_t *h) {
    if (!h) return;
    if (h->rank == 0 && !h->unlinked && h->sh) unlink(h->path.c_str());
    if (h->sh) munmap((void *)h->sh, h->map_bytes);
    delete h;
}

int uspmv_hostcomm_info(const uspmv_hostcomm_t *h, int *rank, int *size, uint64_t *nonce) {
    if (!h) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_hostcomm_info: NULL argument");
    if (rank) *rank = h->rank;
    if (size) *size = h->size;
    if (nonce) *nonce = h->sh->nonce;
    return USPMV_OK;
}

int uspmv_hostcomm_barrier(uspmv_hostcomm_t *h) {
    if (!h) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_hostcomm_barrier: NULL argument");
    return hc_barrier(h);
}

/* a rank that cannot go on (error outside the communicator) tells the others, so that they fail instead of waiting */
int uspmv_hostcomm_abort(uspmv_hostcomm_t *h) {
    if (!h) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_hostcomm_abort: NULL argument");
    if (h->sh) h->sh->failed.store(1);
    return USPMV_OK;
}

int uspmv_hostcomm_bcast(uspmv_hostcomm_t *h, void *buf, int64_t bytes, int root) {
    if (!h || bytes < 0 || (bytes > 0 && !buf) || root < 0 || root >= h->size) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_hostcomm_bcast: bad argument");
    if (h->size == 1 || bytes == 0) return USPMV_OK;
    const int64_t slot = h->sh->slot_bytes;
    for (int64_t o = 0; o < bytes; o += slot) {
        const int64_t n = std::min(slot, bytes - o);
        if (h->rank == root) memcpy(h->slot(root), (const char *)buf + o, (size_t)n);
        if (int rc = hc_barrier(h)) return rc;
        if (h->rank != root) memcpy((char *)buf + o, h->slot(root), (size_t)n);
        if (int rc = hc_barrier(h)) return rc;
    }
    return USPMV_OK;
}

int uspmv_hostcomm_allgather(uspmv_hostcomm_t *h, const void *send, void *recv, int64_t bytes) {
    if (!h || bytes < 0 || (bytes > 0 && (!send || !recv))) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_hostcomm_allgather: bad argument");
    if (h->size == 1) { if (bytes) memmove(recv, send, (size_t)bytes); return USPMV_OK; }
    const int64_t slot = h->sh->slot_bytes;
    for (int64_t o = 0; o < bytes; o += slot) {
        const int64_t n = std::min(slot, bytes - o);
        memcpy(h->slot(h->rank), (const char *)send + o, (size_t)n);
        if (int rc = hc_barrier(h)) return rc;
        for (int q = 0; q < h->size; ++q) memcpy((char *)recv + (size_t)q * (size_t)bytes + o, h->slot(q), (size_t)n);
        if (int rc = hc_barrier(h)) return rc;
    }
    return USPMV_OK;
}

/* recv[recv_off[q] .. recv_off[q+1]) <- rank q's send[send_off_q[me] .. send_off_q[me+1]); offsets in bytes, size+1 entries.
 * A rank whose segment from q does not have the length q sends fails the whole call on every rank. */
int uspmv_hostcomm_alltoallv(uspmv_hostcomm_t *h, const void *send, const int64_t *send_off, void *recv, const int64_t *recv_off) {
    if (!h || !send_off || !recv_off) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_hostcomm_alltoallv: NULL argument");
    const int P = h->size, me = h->rank;
    for (int p = 0; p < P; ++p)
        if (send_off[p + 1] < send_off[p] || recv_off[p + 1] < recv_off[p]) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_hostcomm_alltoallv: offsets must ascend");
    if ((send_off[P] > send_off[0] && !send) || (recv_off[P] > recv_off[0] && !recv)) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_hostcomm_alltoallv: NULL buffer");
    if (P == 1) {
        if (send_off[1] - send_off[0] != recv_off[1] - recv_off[0]) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_hostcomm_alltoallv: self segment lengths differ");
        if (send_off[1] > send_off[0]) memmove((char *)recv + recv_off[0], (const char *)send + send_off[0], (size_t)(send_off[1] - send_off[0]));
        return USPMV_OK;
    }
    const int64_t slot = h->sh->slot_bytes;
    memcpy(h->table(me), send_off, 8 * ((size_t)P + 1));
    if (int rc = hc_barrier(h)) return rc;
    int64_t hi = 0;
    bool ok = true;
    for (int q = 0; q < P; ++q) {
        const int64_t *t = h->table(q);
        hi = std::max(hi, t[P]);
        if (t[me + 1] - t[me] != recv_off[q + 1] - recv_off[q]) ok = false;
    }
    if (!ok) return hc_fail(h, "all-to-all-v: a peer sends a segment of another length than this rank expects");
    // the send buffers pass through the slots window by window: bytes [w0, w1) of every rank's buffer per round
    for (int64_t w0 = 0; w0 < hi; w0 += slot) {
        const int64_t w1 = w0 + slot;
        const int64_t a = std::max(w0, send_off[0]), b = std::min(w1, send_off[P]);
        if (b > a) memcpy(h->slot(me) + (a - w0), (const char *)send + a, (size_t)(b - a));
        if (int rc = hc_barrier(h)) return rc;
        for (int q = 0; q < P; ++q) {
            const int64_t *t = h->table(q);
            const int64_t sa = std::max(w0, t[me]), sb = std::min(w1, t[me + 1]);
            if (sb > sa) memcpy((char *)recv + recv_off[q] + (sa - t[me]), h->slot(q) + (sa - w0), (size_t)(sb - sa));
        }
        if (int rc = hc_barrier(h)) return rc;
    }
    // every send buffer empty: no window round, so nothing has separated the peers' reads of this rank's offset table from the next
    // call's write to it -- close the table phase here (a rank running ahead would otherwise change table(me) under a slower reader)
    if (hi <= 0) if (int rc = hc_barrier(h)) return rc;
    return USPMV_OK;
}

int uspmv_hostcomm_allreduce_max_f64(uspmv_hostcomm_t *h, double *value) {
    if (!h || !value) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_hostcomm_allreduce_max_f64: NULL argument");
    if (h->size == 1) return USPMV_OK;
    memcpy(h->scalar(h->rank), value, 8);
    if (int rc = hc_barrier(h)) return rc;
    double m = *value;
    for (int q = 0; q < h->size; ++q) { double v; memcpy(&v, h->scalar(q), 8); m = std::max(m, v); }
    if (int rc = hc_barrier(h)) return rc;
    *value = m;
    return USPMV_OK;
}

// ---- the transport view of a host communicator
static int tr_alltoallv(void *ctx, const void *send, const int64_t *so, void *recv, const int64_t *ro) { return uspmv_hostcomm_alltoallv((uspmv_hostcomm_t *)ctx, send, so, recv, ro); }
static int tr_allgather(void *ctx, const void *send, void *recv, int64_t bytes) { return uspmv_hostcomm_allgather((uspmv_hostcomm_t *)ctx, send, recv, bytes); }
static int tr_barrier(void *ctx) { return uspmv_hostcomm_barrier((uspmv_hostcomm_t *)ctx); }

int uspmv_hostcomm_transport(uspmv_hostcomm_t *h, uspmv_transport_t *t) {
    if (!h || !t) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_hostcomm_transport: NULL argument");
    t->ctx = h; t->rank = h->rank; t->size = h->size;
    t->alltoallv = tr_alltoallv; t->allgather = tr_allgather; t->barrier = tr_barrier;
    return USPMV_OK;
}

}  // extern "C"
