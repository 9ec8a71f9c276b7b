"""The C++ distributed step (uspmv_dist_*, csrc/uspmv_dist_api.hip) on one GPU, in LOOPBACK: the process plays one block of
a P-way row partition and every neighbour is itself, so the pack kernel, the grouped RCCL send/recv landing in the tail of x,
the interior / boundary tile split, the side stream and the hipGraph replay all run for real (RCCL self send/recv).  With an
x that repeats with the block height the result is the true multi-rank y of the block's rows, checked bit for bit against the
oracle's single-rank SpMV of the whole matrix (the reference's flow: init_local_structs code/main.cpp:1075-1334 ->
init/finalize_halo_exchange code/classes_structs.hpp:857-995 -> kernel)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, make_x

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "ultimate-spmv_amd", "uspmv")


def _global_reference(pkg, orc, shape, P, C, sigma, scale=1.0):
    """y of the whole matrix for x_global = P copies of the ramp over one block (seg-rows, equal blocks), original order."""
    coo = pkg.gen_stencil27(*shape)
    n = coo.n_rows
    assert n % P == 0
    nl = n // P
    xg = np.tile(make_x(nl) * scale, P)
    s = pkg.convert_to_scs(coo, C, sigma)
    a = s.arrays()
    pkg.permute_scs_cols(s, a["old_to_new_idx"])
    a = s.arrays()
    xp = np.zeros(s.n_rows_padded)
    xp[:n] = pkg.apply_permutation(xg, a["new_to_old_idx"])
    y = orc.spmv_scs(C, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], xp)
    return pkg.apply_permutation(y, a["old_to_new_idx"]), nl


@pytest.mark.parametrize("P,shape,C,sigma", [(2, (24, 24, 24), 32, 512), (4, (16, 16, 40), 32, 512), (3, (20, 9, 27), 16, 64)])
def test_native_step_loopback_bitexact(pkg, orc, P, shape, C, sigma):
    import torch
    torch.cuda.set_device(0)
    y_ref, nl = _global_reference(pkg, orc, shape, P, C, sigma)
    counts = pkg.gen_stencil27_row_counts(*shape)
    wsa = pkg.seg_from_row_counts(counts, "seg-rows", P)
    assert np.array_equal(np.diff(wsa), np.full(P, nl))
    cid = pkg.comm_unique_id()
    for rank in range(P):
        loc = pkg.gen_stencil27(*shape, row_begin=int(wsa[rank]), row_end=int(wsa[rank + 1]))
        d = pkg.DistNative(loc, wsa, C, sigma, rank, P, cid if rank == 0 else pkg.comm_unique_id(), comm_rank=0, comm_size=1)
        assert d.loopback and d.n_local == nl and d.n_halo > 0 and d.n_send == d.n_halo
        assert d.n_interior > 0 and d.n_boundary > 0
        x = d.new_x(make_x(nl))
        want = y_ref[wsa[rank]:wsa[rank + 1]]
        # eager step (overlap), eager without overlap, graph replay
        y = d.new_y(); d.spmv(x, y); d.synchronize()
        assert np.array_equal(d.y_to_original_order(y)[:nl], want), (P, rank, "eager")
        x_tail = x[d.n_local:d.n_local + d.n_halo].clone()
        d.set_overlap(False)
        x[d.n_local:].zero_()
        y2 = d.new_y(); d.spmv(x, y2); d.synchronize()
        assert torch.equal(y, y2) and torch.equal(x[d.n_local:d.n_local + d.n_halo], x_tail)
        d.set_overlap(True)
        x[d.n_local:].zero_()
        # uspmv_dist_run, eager form.  (The hipGraph replay of the step is exercised by the CLI test below, on the system's RCCL:
        # inside a Python process the library binds to the RCCL / HIP runtime that torch ships, whose hipStreamEndCapture crashes
        # on a captured RCCL group -- profiles/r02/dist_graph_capture.txt -- so Python callers keep use_graph = False.)
        y3 = d.new_y(); d.run(x, y3, 5, use_graph=False); d.synchronize()
        d._refresh()
        assert torch.equal(y, y3), (P, rank, "run")
        assert not d.graph_captured and d.eager_steps >= 5
        # without the exchange the boundary rows differ (the halo really is what makes y right)
        x[d.n_local:].zero_()
        y4 = d.new_y(); d.spmv(x, y4, comm_halos=False); d.synchronize()
        assert not torch.equal(y, y4)
        d.barrier()
        assert d.allreduce_max(3.5) == 3.5
        d.close()


def test_native_block_vector_exchange_loopback_bitexact(pkg, orc):
    """uspmv_dist_spmmv: the halo exchange of b vectors in the reference's three message patterns (bulkvec / multivec / singlevec,
    code/classes_structs.hpp:875-924) + the SpMMV kernel, per column against the oracle's single-rank SpMV of the whole matrix."""
    import torch
    torch.cuda.set_device(0)
    P, shape, C, sigma, b = 3, (12, 10, 27), 32, 512, 4
    refs = [_global_reference(pkg, orc, shape, P, C, sigma, scale=1.0 + v / 8.0) for v in range(b)]
    nl = refs[0][1]
    counts = pkg.gen_stencil27_row_counts(*shape)
    wsa = pkg.seg_from_row_counts(counts, "seg-rows", P)
    for rank in range(P):
        loc = pkg.gen_stencil27(*shape, row_begin=int(wsa[rank]), row_end=int(wsa[rank + 1]))
        d = pkg.DistNative(loc, wsa, C, sigma, rank, P, pkg.comm_unique_id(), comm_rank=0, comm_size=1)
        Xo = [make_x(nl) * (1.0 + v / 8.0) for v in range(b)]
        ld = d.padded_vec_size
        for layout, mode in ((pkg.COLWISE, 0), (pkg.COLWISE, 1), (pkg.COLWISE, 2), (pkg.ROWWISE, 0)):
            X = d.new_X(Xo, b, layout)
            Y = torch.zeros(b * ld, dtype=torch.float64, device="cuda")
            d.spmmv(X, Y, b, layout, mode); d.synchronize()
            Yh = Y.cpu().numpy()
            for v in range(b):
                col = Yh[v:d.n_rows_padded * b:b] if layout == pkg.ROWWISE else Yh[v * ld:v * ld + d.n_rows_padded]
                got = pkg.apply_permutation(np.ascontiguousarray(col), d.old_to_new)[:nl]
                assert np.array_equal(got, refs[v][0][wsa[rank]:wsa[rank + 1]]), (rank, layout, mode, v)
        with pytest.raises(pkg.UspmvError):
            d.spmmv(d.new_X(Xo, b, pkg.ROWWISE), torch.zeros(b * ld, dtype=torch.float64, device="cuda"), b, pkg.ROWWISE, 1)
        d.close()


def test_cli_distributed_loopback_dumps_the_right_y(pkg, orc, tmp_path):
    """`uspmv gen:... scs -seg_rows -comm_halos 1` through host/uspmv_dist.cpp with USPMV_LOOPBACK=2: per-rank generation, the
    bench loop on hipGraph replays, the spmv_bench.txt block -- and y of the block against the oracle."""
    shape, P = (24, 24, 24), 2
    y_ref, nl = _global_reference(pkg, orc, shape, P, 32, 512)
    for rank in range(P):
        pre = str(tmp_path / f"y{rank}")
        env = dict(os.environ, USPMV_LOOPBACK=str(P), USPMV_LOOPBACK_RANK=str(rank), USPMV_DIST_X="ramp", USPMV_DUMP_Y=pre,
                   USPMV_ID_DIR=str(tmp_path), USPMV_JOB_ID=f"t{rank}")
        env.pop("WORLD_SIZE", None); env.pop("RANK", None)
        r = subprocess.run([EXE, "gen:24x24x24", "scs", "-c", "32", "-s", "512", "-seg_rows", "-comm_halos", "1", "-bench_time", "0.05",
                            "-print_comm_vol", "1"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "(loopback)" in r.stdout and "hipGraph replay" in r.stdout, r.stdout
        y = np.fromfile(pre + f".{rank}", np.float64)
        assert y.shape == (nl,) and np.array_equal(y, y_ref[rank * nl:(rank + 1) * nl])
    txt = open(tmp_path / "spmv_bench.txt").read()
    assert "with 2 RCCL ranks" in txt and "seg_method: seg-rows" in txt and "Per rank Elems Recvd" in txt
    # block vectors through the harness: -block_vec_size 2 -mpi_mode multivec, vector v = ramp * (1 + v/8)
    pre = str(tmp_path / "Y")
    env = dict(os.environ, USPMV_LOOPBACK="2", USPMV_LOOPBACK_RANK="1", USPMV_DIST_X="ramp", USPMV_DUMP_Y=pre, USPMV_ID_DIR=str(tmp_path), USPMV_JOB_ID="tb")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([EXE, "gen:24x24x24", "scs", "-c", "32", "-s", "512", "-seg_rows", "-comm_halos", "1", "-bench_time", "0.05", "-block_vec_size", "2",
                        "-mpi_mode", "multivec"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    Y = np.fromfile(pre + ".1", np.float64).reshape(2, nl)
    for v in range(2):
        yv, _ = _global_reference(pkg, orc, shape, P, 32, 512, scale=1.0 + v / 8.0)
        assert np.array_equal(Y[v], yv[nl:2 * nl]), v
    assert "block_vec_size: 2" in open(tmp_path / "spmv_bench.txt").read() and "MPI_mode: multivec" in open(tmp_path / "spmv_bench.txt").read()
