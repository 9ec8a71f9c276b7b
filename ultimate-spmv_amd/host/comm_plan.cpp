// Exchange plan: from what this rank NEEDS from whom (uspmv_halo_discover) to what it must SEND to whom.
//
// The reference does this in collect_comm_info (code/mpi_funcs.hpp:1061-1124): organize_cumsums all-gathers every rank's
// recv_counts_cumsum and reads its own column of that matrix (:179-232), collect_comm_idxs ships the requested row ids to
// their owners with MPI_Isend / MPI_Irecv of MPI_INT (:117-172).  Here both steps are all-to-all-v calls over a
// uspmv_transport (include/uspmv.h, L4a) -- one int per pair for the counts, then the ids -- so the same code runs over RCCL,
// over the host communicator (real processes without a GPU) and in loopback.  No HIP in this file.
#include <algorithm>
#include <cstring>

#include "uspmv_internal.hpp"

extern "C" {

int uspmv_comm_plan_create(const uspmv_transport_t *t, const uspmv_halo_t *halo, uspmv_comm_plan_t **out) {
    if (!t || !halo || !out || !t->alltoallv) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_comm_plan_create: NULL argument");
    const int P = halo->P;
    if (t->size != P || t->rank != halo->rank)
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_comm_plan_create: transport is rank %d of %d, the halo description belongs to block %d of %d",
                           t->rank, t->size, halo->rank, P);
    if ((int)halo->recv_counts.size() != P || (int64_t)halo->recv_idxs.size() != halo->n_halo)
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_comm_plan_create: inconsistent halo description");
    auto *p = new uspmv_comm_plan;
    p->P = P; p->rank = halo->rank; p->n_local = halo->n_local;
    p->recv_off.assign((size_t)P + 1, 0);
    for (int q = 0; q < P; ++q) p->recv_off[(size_t)q + 1] = p->recv_off[(size_t)q] + halo->recv_counts[(size_t)q];
    // ---- how many: rank q learns recv_counts[q] of every rank (its send counts)
    std::vector<int32_t> send_counts((size_t)P, 0);
    std::vector<int64_t> o4((size_t)P + 1);
    for (int q = 0; q <= P; ++q) o4[(size_t)q] = 4 * (int64_t)q;
    int rc = t->alltoallv(t->ctx, halo->recv_counts.data(), o4.data(), send_counts.data(), o4.data());
    if (rc) { delete p; return rc; }
    p->send_off.assign((size_t)P + 1, 0);
    for (int q = 0; q < P; ++q) {
        if (send_counts[(size_t)q] < 0) { delete p; return uspmv::fail(USPMV_ERR_COMM, "uspmv_comm_plan_create: rank %d announced a negative count", q); }
        p->send_off[(size_t)q + 1] = p->send_off[(size_t)q] + send_counts[(size_t)q];
    }
    p->n_send = p->send_off[(size_t)P];
    // ---- which: the ids this rank asked owner q for travel to q
    p->send_idxs.assign((size_t)std::max<int64_t>(p->n_send, 1), 0);
    std::vector<int64_t> so((size_t)P + 1), ro((size_t)P + 1);
    for (int q = 0; q <= P; ++q) { so[(size_t)q] = 4 * p->recv_off[(size_t)q]; ro[(size_t)q] = 4 * p->send_off[(size_t)q]; }
    static const int32_t none = 0;
    rc = t->alltoallv(t->ctx, halo->n_halo ? (const void *)halo->recv_idxs.data() : (const void *)&none, so.data(), p->send_idxs.data(), ro.data());
    if (rc) { delete p; return rc; }
    p->send_idxs.resize((size_t)p->n_send);
    // ---- every id must be one of MY rows (a wrong partition on a peer, or loopback with unequal blocks, shows up here and
    //      not as an out-of-bounds gather in the pack kernel); the ranks then agree on the outcome, so that a refusal on one
    //      rank is an error on all of them and nobody walks on into the next collective alone
    int32_t bad_from = -1, bad_id = 0;
    for (int q = 0; q < P && bad_from < 0; ++q)
        for (int64_t k = p->send_off[(size_t)q]; k < p->send_off[(size_t)q + 1]; ++k) {
            const int32_t id = p->send_idxs[(size_t)k];
            if (id < 0 || id >= p->n_local) { bad_from = q; bad_id = id; break; }
        }
    std::vector<int32_t> verdicts((size_t)P, 0);
    const int32_t mine = bad_from >= 0 ? 1 : 0;
    verdicts[(size_t)p->rank] = mine;
    if (t->allgather && P > 1) {
        rc = t->allgather(t->ctx, &mine, verdicts.data(), 4);
        if (rc) { delete p; return rc; }
    }
    if (bad_from >= 0) {
        rc = uspmv::fail(USPMV_ERR_INVALID, "uspmv_comm_plan_create: rank %d asks block %d for its row %d, but the block has %ld rows",
                         bad_from, p->rank, bad_id, (long)p->n_local);
        delete p;
        return rc;
    }
    for (int q = 0; q < P; ++q)
        if (verdicts[(size_t)q]) {
            rc = uspmv::fail(USPMV_ERR_COMM, "uspmv_comm_plan_create: rank %d refused the ids it was asked for (partition mismatch)", q);
            delete p;
            return rc;
        }
    *out = p;
    return USPMV_OK;
}

int uspmv_comm_plan_meta(const uspmv_comm_plan_t *p, int64_t *n_send, const int64_t **send_off, const int32_t **send_idxs, const int64_t **recv_off) {
    if (!p) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_comm_plan_meta: NULL plan");
    if (n_send) *n_send = p->n_send;
    if (send_off) *send_off = p->send_off.data();
    if (send_idxs) *send_idxs = p->send_idxs.data();
    if (recv_off) *recv_off = p->recv_off.data();
    return USPMV_OK;
}

void uspmv_comm_plan_free(uspmv_comm_plan_t *p) { delete p; }

}  // extern "C"
