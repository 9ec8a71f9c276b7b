"""N > 1 path with device vectors: several ranks share the one GPU of the test box (RCCL refuses duplicate
devices, so the collective itself goes through gloo + a host bounce; everything else -- pack kernel,
side stream, interior / boundary tiles of the tile-local-column kernel, ordering -- is the production
path).  Checked against the reference's own per-rank x_local and global y."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT, mtx_path

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, case, overlap, tlc, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        sys.path.insert(0, ROOT)
        import torch
        import torch.distributed as dist
        import __graft_entry__ as ge
        pkg = ge.load_package()
        from ultimate_spmv_amd import binding as B
        from ultimate_spmv_amd.distributed import DistSpmv
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        name, Cc, sg, method = case
        key = f"{name}_C{Cc}_s{sg}_{method}_P{world}"
        h = np.load(os.path.join(GOLDEN, "halo.npz"))
        tot = pkg.read_mtx(mtx_path(name))
        wsa = pkg.seg_work_sharing_arr(tot, method, world)
        loc = B.seg_local_coo(tot, wsa, rank)
        d = DistSpmv(loc, wsa, Cc, sg, device="cuda:0", overlap=overlap, tlc=tlc)
        assert d.use_tiles == (tlc and Cc >= 32 and 256 % Cc == 0)   # narrower chunks: internal re-chunk, chunk-id split
        xg = 1.0 + 1e-3 * (np.arange(tot.n_rows) % 1000)
        x = d.new_x(xg[wsa[rank]:wsa[rank + 1]])
        y = d.new_y()
        for _ in range(3):
            d.spmv(x, y)
        torch.cuda.synchronize()
        assert np.array_equal(x.cpu().numpy(), h[f"{key}_r{rank}_x_local"])
        assert np.array_equal(d.y_to_original_order(y), h[key + "_y_global"][wsa[rank]:wsa[rank + 1]])
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))


CASES = [(("bcsstk13", 32, 512, "seg-rows"), 2), (("bcsstk13", 32, 512, "seg-nnz"), 4), (("FDM-2d-16", 16, 512, "seg-nnz"), 3),
         (("matrix1", 10, 3, "seg-rows"), 2)]


@pytest.mark.parametrize("case,world", CASES)
@pytest.mark.parametrize("overlap,tlc", [(False, False), (True, False), (True, True)])
def test_ranks_sharing_one_gpu(case, world, overlap, tlc, pkg):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, overlap, tlc, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def _worker_block(rank, world, port, case, b, q):
    """Distributed SpMMV with device block vectors (production gather + SpMMV kernels, gloo bounce for the collective):
    every column equals the single-GPU SpMV of that column on the whole matrix, bit for bit, both layouts."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        sys.path.insert(0, ROOT)
        import torch
        import torch.distributed as dist
        import __graft_entry__ as ge
        pkg = ge.load_package()
        from ultimate_spmv_amd import binding as B
        from ultimate_spmv_amd.distributed import DistSpmv
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        name, Cc, sg, method = case
        tot = pkg.read_mtx(mtx_path(name))
        wsa = pkg.seg_work_sharing_arr(tot, method, world)
        d = DistSpmv(B.seg_local_coo(tot, wsa, rank), wsa, Cc, sg, device="cuda:0")
        st = pkg.convert_to_scs(tot, Cc, sg)
        ta = st.arrays()
        pkg.permute_scs_cols(st, ta["old_to_new_idx"]); ta = st.arrays()
        At = pkg.DeviceMatrix(st, "cuda:0")
        cols = [(1.0 + 1e-3 * (np.arange(tot.n_rows) % 1000)) * (1.0 + v / 8.0) for v in range(b)]
        truth = []
        for xg in cols:
            xp = np.zeros(max(st.n_rows_padded, tot.n_rows)); xp[:tot.n_rows] = pkg.apply_permutation(xg, ta["new_to_old_idx"])
            yt = torch.zeros(st.n_rows_padded, dtype=torch.float64, device="cuda:0")
            pkg.spmv(At, torch.from_numpy(xp).cuda(), yt)
            truth.append(pkg.apply_permutation(yt.cpu().numpy(), ta["old_to_new_idx"]))
        ld = d.padded_vec_size
        for layout in (B.COLWISE, B.ROWWISE):
            X = d.new_X([xg[wsa[rank]:wsa[rank + 1]] for xg in cols], b, layout)
            Y = torch.zeros(b * ld, dtype=torch.float64, device="cuda:0")
            for _ in range(2):
                d.spmmv(X, Y, b, layout)
            torch.cuda.synchronize()
            Yh = Y.cpu().numpy()
            for v in range(b):
                yv = Yh[v:d.scs.n_rows_padded * b:b] if layout == B.ROWWISE else Yh[v * ld:v * ld + d.scs.n_rows_padded]
                assert np.array_equal(pkg.apply_permutation(np.ascontiguousarray(yv), d.old_to_new), truth[v][wsa[rank]:wsa[rank + 1]]), (layout, v)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))


@pytest.mark.parametrize("case,world,b", [(("bcsstk13", 32, 512, "seg-nnz"), 2, 8), (("bcsstk13", 32, 512, "seg-rows"), 4, 4)])
def test_block_vectors_ranks_sharing_one_gpu(case, world, b, pkg):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_block, args=(r, world, port, case, b, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"
