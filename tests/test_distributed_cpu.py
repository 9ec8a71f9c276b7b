"""N > 1 path on CPU: world_size-2 gloo processes run the product's DistSpmv (partitioning, halo
discovery, index all-to-all, the per-step all-to-all-v straight into the tail of x, the
interior / boundary split).  The product has no CPU kernels, so the tests inject the oracle as
the pack / kernel callables; what is under test is the distributed data movement."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT, golden, mtx_path


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, case, overlap, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
        import __graft_entry__ as ge
        pkg = ge.load_package()
        from oracle import oracle as orc
        from ultimate_spmv_amd import binding as B
        from ultimate_spmv_amd.distributed import DistSpmv, seg_from_row_counts
        dist.init_process_group("gloo", rank=rank, world_size=world)
        name, Cc, sg, method = case
        key = f"{name}_C{Cc}_s{sg}_{method}_P{world}"
        h = np.load(os.path.join(GOLDEN, "halo.npz"))
        tot = pkg.read_mtx(mtx_path(name))
        wsa = pkg.seg_work_sharing_arr(tot, method, world)
        I, _, _ = tot.arrays()
        assert np.array_equal(wsa, seg_from_row_counts(np.bincount(I, minlength=tot.n_rows), method, world))
        loc = B.seg_local_coo(tot, wsa, rank)

        def pack(x, perm, idx, out):      # oracle twin of pack_send_buf
            out[:len(idx)] = torch.from_numpy(orc.pack_send_buf(x.numpy(), perm.numpy(), idx.numpy()))

        def spmv_all(d, x, y):
            a = d.scs.arrays()
            y[:d.scs.n_rows_padded] = torch.from_numpy(orc.spmv_scs(d.scs.C, d.scs.n_chunks, a["chunk_ptrs"],
                                                                    a["chunk_lengths"], a["col_idxs"], a["values"], x.numpy()))
            return y

        def spmv_ids(d, ids, x, y):
            a = d.scs.arrays()
            full = orc.spmv_scs(d.scs.C, d.scs.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], x.numpy())
            for c in ids.numpy():
                y[c * d.scs.C:(c + 1) * d.scs.C] = torch.from_numpy(full[c * d.scs.C:(c + 1) * d.scs.C])
            return y

        d = DistSpmv(loc, wsa, Cc, sg, device="cpu", overlap=overlap, pack_fn=pack, spmv_fn=spmv_all,
                     spmv_chunks_fn=spmv_ids)
        assert d.n_halo == len(h[f"{key}_r{rank}_recv_idxs"])
        assert np.array_equal(d.scs.arrays()["col_idxs"], h[f"{key}_r{rank}_col_idxs"])
        xg = 1.0 + 1e-3 * (np.arange(tot.n_rows) % 1000)
        x = d.new_x(xg[wsa[rank]:wsa[rank + 1]])
        y = d.new_y()
        for _ in range(2):               # twice: buffers are reused across steps
            d.spmv(x, y)
        assert np.array_equal(x.numpy(), h[f"{key}_r{rank}_x_local"]), "halo region differs from the reference's"
        yo = d.y_to_original_order(y)
        assert np.array_equal(yo, h[key + "_y_global"][wsa[rank]:wsa[rank + 1]])
        # every rank's send list == what the peers asked for
        sent = torch.tensor([d.n_send], dtype=torch.int64); got = torch.tensor([d.n_halo], dtype=torch.int64)
        dist.all_reduce(sent); dist.all_reduce(got)
        assert int(sent) == int(got)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))


CASES = [(("bcsstk13", 32, 512, "seg-rows"), 2), (("matrix1", 10, 3, "seg-rows"), 2), (("impcol_e", 8, 16, "seg-nnz"), 2),
         (("FDM-2d-16", 16, 512, "seg-nnz"), 3), (("bcsstk13", 32, 512, "seg-nnz"), 4), (("bcsstk13", 32, 512, "seg-nnz"), 8)]


@pytest.mark.parametrize("case,world", CASES)
@pytest.mark.parametrize("overlap", [False, True])
def test_gloo_halo_exchange(case, world, overlap, pkg, orc):
    if world == 8 and not overlap:
        pytest.skip("8 ranks: overlap variant only (keeps the CPU suite short)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, overlap, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def _worker_block(rank, world, port, case, b, q):
    """Distributed SpMMV (block-vector halo exchange in one message per neighbour), both layouts."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
        import __graft_entry__ as ge
        pkg = ge.load_package()
        from oracle import oracle as orc
        from ultimate_spmv_amd import binding as B
        from ultimate_spmv_amd.distributed import DistSpmv
        dist.init_process_group("gloo", rank=rank, world_size=world)
        name, Cc, sg, method = case
        tot = pkg.read_mtx(mtx_path(name))
        wsa = pkg.seg_work_sharing_arr(tot, method, world)
        loc = B.seg_local_coo(tot, wsa, rank)

        def gather(out, vec, idx):
            out.copy_(vec[idx.long()])

        def spmmv(d, X, Y, bb, ld, layout):
            a = d.scs.arrays()
            Y.copy_(torch.from_numpy(orc.spmmv_scs(d.scs.C, d.scs.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"],
                                                   a["values"], X.numpy(), bb, ld, layout == B.ROWWISE)))
            return Y

        d = DistSpmv(loc, wsa, Cc, sg, device="cpu", gather_fn=gather, spmmv_fn=spmmv)
        # single-rank truth, column by column, on the whole matrix (same per-row summation order)
        st = pkg.convert_to_scs(tot, Cc, sg)
        ta = st.arrays()
        pkg.permute_scs_cols(st, ta["old_to_new_idx"]); ta = st.arrays()
        cols = [(1.0 + 1e-3 * (np.arange(tot.n_rows) % 1000)) * (1.0 + v / 8.0) for v in range(b)]
        truth = []
        for xg in cols:
            xp = np.zeros(max(st.n_rows_padded, tot.n_rows)); xp[:tot.n_rows] = pkg.apply_permutation(xg, ta["new_to_old_idx"])
            yp = orc.spmv_scs(Cc, st.n_chunks, ta["chunk_ptrs"], ta["chunk_lengths"], ta["col_idxs"], ta["values"], xp)
            truth.append(pkg.apply_permutation(yp, ta["old_to_new_idx"]))
        ld = d.padded_vec_size
        for layout in (B.COLWISE, B.ROWWISE):
            X = d.new_X([xg[wsa[rank]:wsa[rank + 1]] for xg in cols], b, layout)
            Y = torch.zeros(b * ld, dtype=torch.float64)
            for _ in range(2):
                d.spmmv(X, Y, b, layout)
            for v in range(b):
                yv = Y[v:d.scs.n_rows_padded * b:b] if layout == B.ROWWISE else Y[v * ld:v * ld + d.scs.n_rows_padded]
                yo = pkg.apply_permutation(np.ascontiguousarray(yv.numpy()), d.old_to_new)
                assert np.array_equal(yo, truth[v][wsa[rank]:wsa[rank + 1]]), (layout, v)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))


@pytest.mark.parametrize("case,world,b", [(("bcsstk13", 32, 512, "seg-nnz"), 2, 4), (("impcol_e", 8, 16, "seg-rows"), 3, 3),
                                          (("bcsstk13", 32, 512, "seg-rows"), 4, 8)])
def test_gloo_block_vector_halo_exchange(case, world, b, pkg, orc):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_block, args=(r, world, port, case, b, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"
