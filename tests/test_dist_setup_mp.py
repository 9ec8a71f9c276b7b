"""The C++ set-up of the distributed step, driven by REAL processes (world_size 2, 4, 8) on the CPU.

What runs in every rank is the product's own code -- uspmv_hostcomm_* (host/hostcomm.cpp), uspmv_seg_*, uspmv_convert_to_scs,
uspmv_halo_discover, uspmv_comm_plan_create (host/comm_plan.cpp: the code uspmv_dist_create runs on the GPU box, there over RCCL)
-- with no torch.distributed and no Python twin in between.  Checked against the reference's own numbers (tests/golden/halo.npz,
made by the genuine collect_local_needed_heri / seg_work_sharing_arr, code/mpi_funcs.hpp:242-415, :424-622): what a rank must
SEND to p is what p's reference recv list asks it for (comm_send_idxs, code/mpi_funcs.hpp:117-172), and x_local after one
exchange over that plan equals the reference's x_local, halo tail included."""
import multiprocessing as mp
import os
import sys
import time

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, mtx_path


def _job(tag):
    return f"t{os.getpid()}_{tag}_{time.monotonic_ns()}"


def _run(target, world, args, timeout=120):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, q) + tuple(args)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in procs:
            res.append(q.get(timeout=timeout))
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    return sorted(res)


def _pkg():
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    import __graft_entry__ as ge
    return ge.load_package()


def _setup_worker(rank, world, q, job, case):
    try:
        pkg = _pkg()
        from ultimate_spmv_amd import binding as B
        name, Cc, sg, method = case
        hc = pkg.HostComm(job, rank, world, timeout_s=60)
        key = f"{name}_C{Cc}_s{sg}_{method}_P{world}"
        h = np.load(os.path.join(GOLDEN, "halo.npz"))
        tot = pkg.read_mtx(mtx_path(name))
        wsa = pkg.seg_work_sharing_arr(tot, method, world)
        assert np.array_equal(wsa, h[key + "_wsa"])
        loc = B.seg_local_coo(tot, wsa, rank)
        s = pkg.convert_to_scs(loc, Cc, sg, B.F64)
        halo = pkg.HaloPlan(s, wsa, rank, world)
        a = s.arrays()
        pkg.permute_scs_cols(s, a["old_to_new_idx"])
        a = s.arrays()
        assert np.array_equal(a["col_idxs"], h[f"{key}_r{rank}_col_idxs"])
        assert np.array_equal(halo.recv_idxs, h[f"{key}_r{rank}_recv_idxs"])
        plan = pkg.CommPlan(hc.transport, halo)
        # what I must send to p == what p's reference recv list asks me (owner = rank) for
        n_local = int(wsa[rank + 1] - wsa[rank])
        for p in range(world):
            cum = h[f"{key}_r{p}_recv_cumsum"].astype(np.int64)
            cnt = np.diff(cum)
            off = np.concatenate([[0], np.cumsum(cnt)])
            want = h[f"{key}_r{p}_recv_idxs"][off[rank]:off[rank + 1]]
            got = plan.send_idxs[plan.send_off[p]:plan.send_off[p + 1]]
            assert np.array_equal(got, want), (p, got[:8], want[:8])
        assert plan.n_send == plan.send_off[-1] and plan.recv_off[-1] == halo.n_halo
        assert np.all(plan.send_idxs >= 0) and np.all(plan.send_idxs < n_local)
        # one halo exchange over that plan (pack_send_buf: send[i] = x[perm[send_idxs[i]]], code/classes_structs.hpp:813-831)
        xg = 1.0 + 1e-3 * (np.arange(tot.n_rows) % 1000)
        ld = n_local + max(s.n_rows_padded - n_local, halo.n_halo)
        x = np.zeros(ld)
        x[:n_local] = pkg.apply_permutation(xg[wsa[rank]:wsa[rank + 1]], a["new_to_old_idx"])
        send = x[a["old_to_new_idx"][plan.send_idxs]] if plan.n_send else np.zeros(0)
        x[n_local:n_local + halo.n_halo] = hc.alltoallv(send, np.diff(plan.send_off), np.diff(plan.recv_off))
        gx = h[f"{key}_r{rank}_x_local"]
        assert np.array_equal(x[:len(gx)], gx), "x_local (halo tail included) differs from the reference's"
        # the ranks agree on the totals
        tot_send = hc.allgather(np.array([plan.n_send, halo.n_halo], np.int64))
        assert tot_send[:, 0].sum() == tot_send[:, 1].sum()
        hc.barrier()
        hc.close()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))


@pytest.mark.parametrize("case,world", [(("bcsstk13", 32, 512, "seg-nnz"), 4), (("bcsstk13", 32, 512, "seg-nnz"), 8),
                                        (("bcsstk13", 32, 512, "seg-rows"), 2), (("FDM-2d-16", 16, 512, "seg-nnz"), 3),
                                        (("impcol_e", 8, 16, "seg-nnz"), 2), (("FDM-2d-16", 4, 8, "seg-rows"), 4)])
def test_cpp_setup_with_real_processes_matches_reference(case, world, pkg):
    key = f"{case[0]}_C{case[1]}_s{case[2]}_{case[3]}_P{world}_wsa"
    assert key in np.load(os.path.join(GOLDEN, "halo.npz"))
    for rank, msg in _run(_setup_worker, world, (_job("setup"), case)):
        assert msg == "ok", f"rank {rank}: {msg}"


def _coll_worker(rank, world, q, job, slot):
    try:
        if slot:
            os.environ["USPMV_HC_SLOT_BYTES"] = str(slot)
        pkg = _pkg()
        hc = pkg.HostComm(job, rank, world, timeout_s=60)
        nonces = hc.allgather(np.array([hc.nonce], np.uint64))
        assert len(set(nonces.ravel().tolist())) == 1 and hc.nonce != 0
        # broadcast of something larger than a slot
        buf = np.arange(1000, dtype=np.int64) * 7 if rank == 1 % world else np.zeros(1000, np.int64)
        hc.bcast(buf, root=1 % world)
        assert np.array_equal(buf, np.arange(1000, dtype=np.int64) * 7)
        # all-gather
        g = hc.allgather(np.full(37, rank, np.int32))
        assert g.shape == (world, 37) and all((g[r] == r).all() for r in range(world))
        # ragged all-to-all-v: rank r sends (r + 2 q) % 5 elements r*1000 + q*10 + k to q, empty segments included
        sc = [(rank + 2 * qq) % 5 for qq in range(world)]
        rc = [(qq + 2 * rank) % 5 for qq in range(world)]
        send = np.concatenate([np.array([rank * 1000 + qq * 10 + k for k in range(sc[qq])], np.float64) for qq in range(world)] or [np.zeros(0)])
        got = hc.alltoallv(send, sc, rc)
        want = np.concatenate([np.array([qq * 1000 + rank * 10 + k for k in range(rc[qq])], np.float64) for qq in range(world)] or [np.zeros(0)])
        assert np.array_equal(got, want)
        # an all-EMPTY all-to-all-v straight before non-empty ones, many times, with one rank running late: a rank that left the empty call
        # early must not rewrite its offset table under a peer that still reads it (the table phase is closed by a barrier)
        import time
        for it in range(40):
            if rank == it % world:
                time.sleep(0.002)
            e = hc.alltoallv(np.zeros(0, np.float64), [0] * world, [0] * world)
            assert e.size == 0
            sc2 = [(rank + qq + it) % 3 for qq in range(world)]
            rc2 = [(qq + rank + it) % 3 for qq in range(world)]
            send2 = np.concatenate([np.full(sc2[qq], rank * 100 + qq + it, np.float64) for qq in range(world)] or [np.zeros(0)])
            got2 = hc.alltoallv(send2, sc2, rc2)
            want2 = np.concatenate([np.full(rc2[qq], qq * 100 + rank + it, np.float64) for qq in range(world)] or [np.zeros(0)])
            assert np.array_equal(got2, want2), (it, got2, want2)
        assert hc.allreduce_max(float(rank)) == float(world - 1)
        hc.barrier()
        hc.close()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))


@pytest.mark.parametrize("world,slot", [(1, 0), (2, 0), (3, 64), (4, 128)])
def test_hostcomm_collectives(world, slot, pkg):
    """slot 64 / 128 bytes: every collective needs several rounds through the shared slots"""
    for rank, msg in _run(_coll_worker, world, (_job("coll"), slot)):
        assert msg == "ok", f"rank {rank}: {msg}"


def _late_worker(rank, world, q, job, stale_kind):
    try:
        pkg = _pkg()
        import ctypes as C
        d = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
        path = os.path.join(d, "uspmv_hc_" + job)
        if rank == 1:
            # a leftover of a "crashed" job is already in place when this rank looks: garbage, or a well-formed segment of a DEAD creator
            if stale_kind == "garbage":
                open(path, "wb").write(b"\x00" * 4096)
            time.sleep(0.3)
        else:
            time.sleep(1.0)     # rank 0 arrives late and replaces the leftover atomically
        hc = pkg.HostComm(job, rank, world, timeout_s=30)
        g = hc.allgather(np.array([rank], np.int32))
        assert g.ravel().tolist() == list(range(world))
        assert not os.path.exists(path), "the segment's name must be gone once everybody has attached"
        hc.close()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))


def test_hostcomm_ignores_leftover_of_a_crashed_job(pkg):
    for rank, msg in _run(_late_worker, 2, (_job("stale"), "garbage")):
        assert msg == "ok", f"rank {rank}: {msg}"


def _dead_creator_segment(pkg, job):
    """make a WELL-FORMED segment whose creator is dead: a child process creates a 2-rank communicator as rank 0 and is killed
    while it waits for rank 1"""
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_orphan, args=(job,))
    p.start()
    d = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
    path = os.path.join(d, "uspmv_hc_" + job)
    for _ in range(200):
        if os.path.exists(path):
            break
        time.sleep(0.05)
    assert os.path.exists(path)
    p.kill(); p.join()
    return path


def _orphan(job):
    pkg = _pkg()
    pkg.HostComm(job, 0, 2, timeout_s=60)   # blocks in the seating barrier until killed


def _after_crash_worker(rank, world, q, job):
    try:
        pkg = _pkg()
        if rank == 0:
            time.sleep(0.7)    # rank 1 meets the dead job's segment first and must not sit down in it
        hc = pkg.HostComm(job, rank, world, timeout_s=30)
        assert hc.allgather(np.array([rank], np.int32)).ravel().tolist() == [0, 1]
        hc.close()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))


def test_hostcomm_does_not_join_a_dead_jobs_segment(pkg):
    job = _job("dead")
    path = _dead_creator_segment(pkg, job)
    try:
        for rank, msg in _run(_after_crash_worker, 2, (job,)):
            assert msg == "ok", f"rank {rank}: {msg}"
    finally:
        if os.path.exists(path):
            os.unlink(path)


def _missing_peer_worker(rank, world, q, job):
    try:
        pkg = _pkg()
        t0 = time.time()
        try:
            pkg.HostComm(job, rank, world, timeout_s=1.5)
            q.put((rank, "FAIL: created a communicator although a rank never arrived"))
        except pkg.UspmvError as e:
            assert e.status == 8 and time.time() - t0 < 20, e   # USPMV_ERR_COMM, not a hang
            q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))


def test_hostcomm_times_out_instead_of_hanging(pkg):
    """world 3, but only ranks 0 and 1 start"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    job = _job("missing")
    procs = [ctx.Process(target=_missing_peer_worker, args=(r, 3, q, job)) for r in (0, 1)]
    for p in procs:
        p.start()
    res = [q.get(timeout=60) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def _mismatch_worker(rank, world, q, job):
    """rank 1 uses ANOTHER partition than rank 0: the ids it asks for do not exist there -> an error on every rank, no out-of-bounds gather"""
    try:
        pkg = _pkg()
        from ultimate_spmv_amd import binding as B
        hc = pkg.HostComm(job, rank, world, timeout_s=30)
        tot = pkg.read_mtx(mtx_path("bcsstk13"))
        wsa = pkg.seg_work_sharing_arr(tot, "seg-rows", world)
        if rank == 1:
            wsa = wsa.copy(); wsa[1] = 1500     # believes block 0 has 1500 rows (it has 1001): asks it for rows it does not own
        loc = B.seg_local_coo(tot, wsa, rank)
        s = pkg.convert_to_scs(loc, 32, 512, B.F64)
        halo = pkg.HaloPlan(s, wsa, rank, world)
        try:
            pkg.CommPlan(hc.transport, halo)
            q.put((rank, "FAIL: plan accepted"))
        except pkg.UspmvError as e:
            assert e.status in (1, 8), e
            q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))


def test_partition_mismatch_is_an_error_on_every_rank(pkg):
    for rank, msg in _run(_mismatch_worker, 2, (_job("mismatch"),)):
        assert msg == "ok", f"rank {rank}: {msg}"
