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
        # uspmv_dist_run: eager, then replayed from ONE captured hipGraph -- also inside this torch process (round 2 crashed here:
        # under torch's bundled HIP 7.0 / RCCL 2.26 an RCCL group captured on a JOINED stream kills hipStreamEndCapture; the step
        # now keeps the group on the capture's origin stream, profiles/r03/graph_capture_diag.txt)
        y3 = d.new_y(); d.run(x, y3, 5, use_graph=False); d.synchronize()
        d._refresh()
        assert torch.equal(y, y3), (P, rank, "run")
        assert not d.graph_captured and d.eager_steps >= 5
        x[d.n_local:].zero_()
        y5 = d.new_y(); d.run(x, y5, 7, use_graph=True); d.synchronize()
        d._refresh()
        assert d.graph_captured and d.graph_launches == 7, (d.graph_captured, d.graph_launches)
        assert torch.equal(y, y5) and torch.equal(x[d.n_local:d.n_local + d.n_halo], x_tail), (P, rank, "graph")
        d.set_option("ba_synch", 1)            # the per-step barrier is part of the captured step
        y6 = d.new_y(); d.run(x, y6, 3, use_graph=True); d.synchronize()
        d._refresh()
        assert d.graph_captured and torch.equal(y, y6)
        d.set_option("ba_synch", 0)
        # without the exchange the boundary rows differ (the halo really is what makes y right)
        x[d.n_local:].zero_()
        y4 = d.new_y(); d.spmv(x, y4, comm_halos=False); d.synchronize()
        assert not torch.equal(y, y4)
        d.barrier()
        assert d.allreduce_max(3.5) == 3.5
        d.close()


def _global_reference_x(pkg, orc, shape, P, C, sigma, xblock):
    """as _global_reference for an arbitrary block of x (x_global = P copies of it)"""
    coo = pkg.gen_stencil27(*shape)
    n = coo.n_rows
    s = pkg.convert_to_scs(coo, C, sigma)
    a = s.arrays()
    pkg.permute_scs_cols(s, a["old_to_new_idx"])
    a = s.arrays()
    xp = np.zeros(s.n_rows_padded)
    xp[:n] = pkg.apply_permutation(np.tile(xblock, P), a["new_to_old_idx"])
    with np.errstate(invalid="ignore"):
        y = orc.spmv_scs(C, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], xp)
    return pkg.apply_permutation(y, a["old_to_new_idx"])


def test_padding_tiles_run_with_the_interior_and_rerun_when_they_must(pkg, orc):
    """The reference pads chunks with (+0, column 0), a halo column on ranks > 0: tiles that touch the halo only that way run before
    the exchange (uspmv_dist_pad_info).  Bit-exact against the oracle's single-rank product in every case: positive x[0] (no re-run:
    the slot held +0 before), negative x[0] (sign differs from the slot's old content: one re-run, none on the next step), x[0] = Inf
    (0 * Inf = NaN in every padded row, as in the reference: re-run every step), eager and graph replay, and with "pad_split" 0."""
    import torch
    torch.cuda.set_device(0)
    P, shape, C, sigma = 4, (16, 16, 40), 32, 512
    counts = pkg.gen_stencil27_row_counts(*shape)
    wsa = pkg.seg_from_row_counts(counts, "seg-rows", P)
    nl = int(wsa[1] - wsa[0])
    cases = {}
    for tag, x0 in (("pos", 3.25), ("neg", -2.5), ("inf", np.inf)):
        xb = make_x(nl).copy(); xb[0] = x0
        cases[tag] = (xb, _global_reference_x(pkg, orc, shape, P, C, sigma, xb))
    assert np.isnan(cases["inf"][1]).any() and not np.isnan(cases["neg"][1]).any()
    for rank in (0, 2):
        loc = pkg.gen_stencil27(*shape, row_begin=int(wsa[rank]), row_end=int(wsa[rank + 1]))
        d = pkg.DistNative(loc, wsa, C, sigma, rank, P, pkg.comm_unique_id(), comm_rank=0, comm_size=1)
        info = d.pad_info()
        d.set_option("pad_split", 1)                                         # (an option: off by default)
        if rank == 0:
            assert info["pad_tiles"] == 0 and info["pad_col"] == -1          # column 0 is local there
            for fused in (0, 1):                                             # the one-launch step without padding tiles
                d.set_option("fused_step", fused)
                x = d.new_x(cases["neg"][0]); y = d.new_y()
                d.spmv(x, y); d.spmv(x, y); d.synchronize()
                assert np.array_equal(d.y_to_original_order(y), cases["neg"][1][wsa[0]:wsa[1]])
            d.close(); continue
        assert info["pad_tiles"] > 0 and info["pad_col"] >= d.n_local and info["real_boundary_tiles"] > 0, info
        assert info["pad_tiles"] + info["real_boundary_tiles"] == d.n_boundary
        want = {k: v[1][wsa[rank]:wsa[rank + 1]] for k, v in cases.items()}

        def run(tag, graph=False, steps=1):
            x = d.new_x(cases[tag][0]); y = d.new_y()
            if graph: d.run(x, y, steps, use_graph=True)
            else:
                for _ in range(steps): d.spmv(x, y)
            d.synchronize()
            got = d.y_to_original_order(y)
            assert np.array_equal(got, want[tag], equal_nan=True), (tag, graph, steps)
            return d.pad_info()["reruns"]

        for fused in (0, 1):                                  # two launches around the exchange / one launch + deferred entries
            d.set_option("pad_split", 1); d.set_option("fused_step", fused)
            r0 = d.pad_info()["reruns"]
            assert run("pos") == r0                           # a fresh x: the slot held +0, same sign, no re-run
            r1 = run("neg")                                   # fresh x again: +0 -> -2.5
            assert r1 == r0 + 1, (fused, r0, r1)
            x = d.new_x(cases["neg"][0]); y = d.new_y()
            d.spmv(x, y); d.spmv(x, y); d.synchronize()       # second step on the SAME x: the slot already holds -2.5
            assert np.array_equal(d.y_to_original_order(y), want["neg"]) and d.pad_info()["reruns"] == r1 + 1
            r2 = d.pad_info()["reruns"]
            assert run("inf", steps=2) == r2 + 2              # not finite: every step
            r3 = run("neg", graph=True, steps=3)              # the captured step (always the two-launch form) carries guard and conditional list too
            assert r3 == r2 + 2 + 1
            d.set_option("pad_split", 0)
            assert run("neg") == r3 and run("inf") == r3 and run("pos") == r3
        d.close()


def test_autotune_picks_a_form_and_keeps_the_bits(pkg, orc):
    """uspmv_dist_autotune (loopback): every arrangement gets a time, the fastest is applied to the object (its options say so), x keeps its
    local part, and a step afterwards still gives the oracle's rows -- eager steps and graph replay, with the self-check armed."""
    import torch
    torch.cuda.set_device(0)
    P, shape, C, sigma = 2, (24, 24, 48), 32, 512
    y_ref, nl = _global_reference(pkg, orc, shape, P, C, sigma)
    counts = pkg.gen_stencil27_row_counts(*shape)
    wsa = pkg.seg_from_row_counts(counts, "seg-rows", P)
    rank = 1
    loc = pkg.gen_stencil27(*shape, row_begin=int(wsa[rank]), row_end=int(wsa[rank + 1]))
    d = pkg.DistNative(loc, wsa, C, sigma, rank, P, pkg.comm_unique_id(), comm_rank=0, comm_size=1)
    assert d.comm_count() == 1                       # (loopback: a communicator of one rank)
    for graph, every, forms in ((False, True, {"overlap", "plain", "pad", "fused"}), (True, True, {"overlap", "plain", "pad"}),
                                (False, False, {"overlap", "plain"}), (True, False, {"overlap", "plain"})):
        x = d.new_x(make_x(nl)); y = d.new_y()
        x0 = x.clone()
        form, ms = d.autotune(x, y, use_graph=graph, local=loc, wsa=wsa, all_forms=every)
        d.synchronize()
        assert set(ms) == forms and form in forms and all(v > 0 for v in ms.values()), (form, ms)
        assert ms[form] == min(ms.values())
        assert torch.equal(x[:nl], x0[:nl])
        d.run(x, y, 2, use_graph=graph); d.synchronize()
        assert np.array_equal(d.y_to_original_order(y)[:nl], y_ref[wsa[rank]:wsa[rank + 1]]), (graph, form)
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


def test_native_block_vector_step_in_two_parts_loopback_bitexact(pkg, orc):
    """uspmv_dist_spmmv with "overlap" 1 (default): interior chunks on the side stream while the block exchange runs, boundary chunks
    after it -- the same kernels over chunk-length arrays that mark the other part (csrc/uspmv_dist_api.hip).  Per column bit-identical
    to the oracle's single-rank SpMV of the whole matrix, in both layouts: gather kernels (b = 3 generic, b = 4 / 8 row-major form) and
    the phased block plan built through set_option("block_plan", 8), whose tiles are classified from the plan's own row lists; and
    identical to the exchange-then-compute step ("overlap" 0).  The step counters prove which form ran."""
    import torch
    torch.cuda.set_device(0)
    P, shape, C, sigma = 2, (24, 24, 48), 32, 512
    counts = pkg.gen_stencil27_row_counts(*shape)
    wsa = pkg.seg_from_row_counts(counts, "seg-rows", P)
    refs = {}
    for b, plan in ((3, 0), (4, 0), (8, 0), (8, 8), (4, 4)):
        for v in range(b):
            if v not in refs: refs[v] = _global_reference(pkg, orc, shape, P, C, sigma, scale=1.0 + v / 8.0)
        nl = refs[0][1]
        for rank in range(P):
            loc = pkg.gen_stencil27(*shape, row_begin=int(wsa[rank]), row_end=int(wsa[rank + 1]))
            d = pkg.DistNative(loc, wsa, C, sigma, rank, P, pkg.comm_unique_id(), comm_rank=0, comm_size=1)
            assert d.n_interior > 0 and d.n_boundary > 0 and d.spmmv_info()["parts"] == 1
            if plan:
                d.set_option("block_plan", plan)
                info = d.spmmv_info()
                if plan == 8:
                    assert info["plan_b"] == 8 and 0 < info["plan_boundary_tiles"] < info["plan_tiles"], info
                else:
                    assert info["plan_b"] == 0            # 32-byte rows: the one-list-per-tile plan, whole-matrix kernels only
            Xo = [make_x(nl) * (1.0 + v / 8.0) for v in range(b)]
            ld = d.padded_vec_size
            for layout, mode in ((pkg.COLWISE, 0), (pkg.COLWISE, 1), (pkg.ROWWISE, 0)):
                got = {}
                for overlap in (1, 0):
                    d.set_option("overlap", overlap)
                    before = d.spmmv_info()
                    X = d.new_X(Xo, b, layout)
                    Y = torch.full((b * ld,), 7.0, dtype=torch.float64, device="cuda")
                    d.spmmv(X, Y, b, layout, mode); d.synchronize()
                    after = d.spmmv_info()
                    two = overlap == 1 and plan != 4
                    assert after["two_part"] - before["two_part"] == (1 if two else 0), (b, plan, layout, overlap, before, after)
                    assert after["one_part"] - before["one_part"] == (0 if two else 1)
                    Yh = Y.cpu().numpy()
                    got[overlap] = Yh
                    for v in range(b):
                        col = Yh[v:d.n_rows_padded * b:b] if layout == pkg.ROWWISE else Yh[v * ld:v * ld + d.n_rows_padded]
                        res = pkg.apply_permutation(np.ascontiguousarray(col), d.old_to_new)[:nl]
                        assert np.array_equal(res, refs[v][0][wsa[rank]:wsa[rank + 1]]), (b, plan, rank, layout, mode, overlap, v)
                n = d.n_rows_padded * b if layout == pkg.ROWWISE else None
                if n is not None: assert np.array_equal(got[0][:n], got[1][:n])
            d.set_option("overlap", 1)
            if plan in (0, 8) and b == 8:                  # each part alone writes its own rows and leaves the others' prefill untouched
                parts = {}
                for part in (1, 2):
                    d.set_option("diag_spmmv_part", part)
                    Y = torch.full((b * ld,), 7.0, dtype=torch.float64, device="cuda")
                    d.spmmv(d.new_X(Xo, b, pkg.ROWWISE), Y, b, pkg.ROWWISE, 0); d.synchronize()
                    parts[part] = Y.cpu().numpy()[:d.n_rows_padded * b].reshape(-1, b)
                d.set_option("diag_spmmv_part", 0)
                full = got[1][:d.n_rows_padded * b].reshape(-1, b)
                w1, w2 = (parts[1] != 7.0).any(axis=1), (parts[2] != 7.0).any(axis=1)
                assert not (w1 & w2).any() and w1.sum() > 0 and w2.sum() > 0
                live = (full != 7.0).any(axis=1)            # (rows whose result is all 7.0 cannot be told apart: none on this matrix)
                assert (w1 | w2)[live].all()
                assert np.array_equal(np.where(w1[:, None], parts[1], parts[2])[live], full[live])
                if plan == 8: assert 0 < w2.sum() <= 64 * d.spmmv_info()["plan_boundary_tiles"]
            if plan == 8:                                  # dropping the plan drops its classes: back to the gather kernels, still two parts
                d.set_option("block_plan", 0)
                assert d.spmmv_info()["plan_b"] == 0
                X = d.new_X(Xo, b, pkg.ROWWISE)
                Y = torch.zeros(b * ld, dtype=torch.float64, device="cuda")
                d.spmmv(X, Y, b, pkg.ROWWISE, 0); d.synchronize()
                col = Y.cpu().numpy()[0:d.n_rows_padded * b:b]
                assert np.array_equal(pkg.apply_permutation(np.ascontiguousarray(col), d.old_to_new)[:nl], refs[0][0][wsa[rank]:wsa[rank + 1]])
            d.close()


def test_cli_distributed_loopback_dumps_the_right_y(pkg, orc, tmp_path):
    """`uspmv gen:... scs -seg_rows -comm_halos 1` through host/uspmv_dist.cpp with USPMV_LOOPBACK=2: per-rank generation, the
    bench loop on hipGraph replays, the spmv_bench.txt block -- and y of the block against the oracle."""
    shape, P = (24, 24, 24), 2
    y_ref, nl = _global_reference(pkg, orc, shape, P, 32, 512)
    for rank in range(P):
        pre = str(tmp_path / f"y{rank}")
        env = dict(os.environ, USPMV_LOOPBACK=str(P), USPMV_LOOPBACK_RANK=str(rank), USPMV_DIST_X="ramp", USPMV_DUMP_Y=pre,
                   USPMV_ID_DIR=str(tmp_path), USPMV_JOB_ID=f"t{rank}", OMP_NUM_THREADS=os.environ.get("USPMV_TEST_OMP", "4"), OMP_WAIT_POLICY="passive")
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
    env = dict(os.environ, USPMV_LOOPBACK="2", USPMV_LOOPBACK_RANK="1", USPMV_DIST_X="ramp", USPMV_DUMP_Y=pre, USPMV_ID_DIR=str(tmp_path), USPMV_JOB_ID="tb", OMP_NUM_THREADS=os.environ.get("USPMV_TEST_OMP", "4"), OMP_WAIT_POLICY="passive")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([EXE, "gen:24x24x24", "scs", "-c", "32", "-s", "512", "-seg_rows", "-comm_halos", "1", "-bench_time", "0.05", "-block_vec_size", "2",
                        "-mpi_mode", "multivec"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    Y = np.fromfile(pre + ".1", np.float64).reshape(2, nl)
    for v in range(2):
        yv, _ = _global_reference(pkg, orc, shape, P, 32, 512, scale=1.0 + v / 8.0)
        assert np.array_equal(Y[v], yv[nl:2 * nl]), v
    assert "block_vec_size: 2" in open(tmp_path / "spmv_bench.txt").read() and "MPI_mode: multivec" in open(tmp_path / "spmv_bench.txt").read()
    assert "steps in two parts" in r.stdout and "phased block plan: no" in r.stdout, r.stdout
    # ... and 8 columns (64-byte X rows): the harness builds the phased block plan and runs the step in two parts on it, row-wise X
    r = subprocess.run([EXE, "gen:24x24x24", "scs", "-c", "32", "-s", "512", "-seg_rows", "-comm_halos", "1", "-bench_time", "0.05", "-block_vec_size", "8",
                        "-block_vec_layout", "rowwise"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "phased block plan: yes" in r.stdout and " 0 exchange-then-compute" in r.stdout, r.stdout
    Y = np.fromfile(pre + ".1", np.float64).reshape(8, nl)
    for v in range(8):
        yv, _ = _global_reference(pkg, orc, shape, P, 32, 512, scale=1.0 + v / 8.0)
        assert np.array_equal(Y[v], yv[nl:2 * nl]), v


# ------------------------------------------------------------------------------------------------------------------------
# Round 3: unequal seg-nnz blocks with REAL ranks.  RCCL refuses several ranks on one GPU, so on a single-GPU box the ranks are
# real processes sharing the card and the per-step exchange is staged through the host communicator (USPMV_EXCHANGE_HOST): the
# same C++ object, set-up, pack kernel, interior / boundary split and kernels as a production run -- only ncclSend / ncclRecv are
# replaced by D2H + all-to-all-v + H2D.  The RCCL calls themselves are what the loopback tests above execute.
import multiprocessing as mp
import sys
import time

from conftest import GOLDEN, mtx_path


def _hx_worker(rank, world, q, job, case):
    try:
        sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ.setdefault("OMP_NUM_THREADS", "4"); os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # (several rank processes share the box's cores)
        import torch
        import __graft_entry__ as ge
        pkg = ge.load_package()
        from ultimate_spmv_amd import binding as B
        torch.cuda.set_device(0)
        name, Cc, sg, method = case
        key = f"{name}_C{Cc}_s{sg}_{method}_P{world}"
        h = np.load(os.path.join(GOLDEN, "halo.npz"))
        hc = pkg.HostComm(job, rank, world, timeout_s=120)
        tot = pkg.read_mtx(mtx_path(name))
        wsa = pkg.seg_work_sharing_arr(tot, method, world)
        assert np.array_equal(wsa, h[key + "_wsa"])
        loc = B.seg_local_coo(tot, wsa, rank)
        d = pkg.DistNative(loc, wsa, Cc, sg, rank, world, hostcomm=hc, host_exchange=True)
        nl = int(wsa[rank + 1] - wsa[rank])
        assert d.n_local == nl and not d.loopback
        # exchange plan == the reference's (what p's recv list asks this rank for)
        n_send, send_off, send_idxs, recv_off = d.comm_plan()
        for p in range(world):
            cnt = np.diff(h[f"{key}_r{p}_recv_cumsum"].astype(np.int64))
            off = np.concatenate([[0], np.cumsum(cnt)])
            assert np.array_equal(send_idxs[send_off[p]:send_off[p + 1]], h[f"{key}_r{p}_recv_idxs"][off[rank]:off[rank + 1]]), p
        assert recv_off[-1] == d.n_halo == len(h[f"{key}_r{rank}_recv_idxs"])
        xg = 1.0 + 1e-3 * (np.arange(tot.n_rows) % 1000)
        want = h[key + "_y_global"][wsa[rank]:wsa[rank + 1]]
        gx = h[f"{key}_r{rank}_x_local"]
        for overlap in (True, False):
            d.set_option("overlap", int(overlap))
            for ba in (0, 1):
                d.set_option("ba_synch", ba)
                x = d.new_x(xg[wsa[rank]:wsa[rank + 1]])
                y = d.new_y()
                d.run(x, y, 2, use_graph=True)          # (host-staged exchange: runs eagerly, the flag must not break it)
                d.synchronize()
                assert np.array_equal(x.cpu().numpy()[:len(gx)], gx), "x_local (halo tail) differs from the reference's"
                assert np.array_equal(d.y_to_original_order(y)[:nl], want), (overlap, ba)
        # the optional arrangements of the step (padding tiles in front of the exchange; one launch with deferred boundary tiles) on
        # unequal blocks and asymmetric lists, eager steps (where the handle has tile lists; otherwise the options change nothing)
        d.set_option("overlap", 1); d.set_option("ba_synch", 0)
        for pad, fused in ((1, 0), (1, 1), (0, 1)):
            d.set_option("pad_split", pad); d.set_option("fused_step", fused)
            x = d.new_x(xg[wsa[rank]:wsa[rank + 1]]); y = d.new_y()
            d.spmv(x, y); d.spmv(x, y); d.synchronize()
            assert np.array_equal(x.cpu().numpy()[:len(gx)], gx), (pad, fused)
            assert np.array_equal(d.y_to_original_order(y)[:nl], want), (pad, fused)
        d.set_option("pad_split", 0); d.set_option("fused_step", 0)
        # the product's own self-check agrees (same x as the golden: x_global[j] = 1 + 1e-3 (j mod 1000))
        x, y = d.new_x(np.zeros(nl)), d.new_y()
        bad, cs = d.check(loc, x, y)
        assert bad == 0, bad
        assert np.array_equal(d.y_to_original_order(y)[:nl], want)
        d.barrier()
        assert d.allreduce_max(float(rank)) == float(world - 1)
        d.close(); hc.close()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))


@pytest.mark.parametrize("case,world", [(("impcol_e", 8, 16, "seg-nnz"), 2), (("FDM-2d-16", 16, 512, "seg-nnz"), 3), (("bcsstk13", 32, 512, "seg-nnz"), 4)])
def test_native_step_real_ranks_unequal_seg_nnz_blocks(pkg, case, world):
    """-seg_nnz partitions with unequal blocks (bcsstk13 at P = 4: heights 724 / 459 / 417 / 403, asymmetric send / recv counts): the C++
    step object in `world` real processes on one GPU, against the reference's x_local and y (tests/golden/halo.npz)."""
    assert f"{case[0]}_C{case[1]}_s{case[2]}_{case[3]}_P{world}_wsa" in np.load(os.path.join(GOLDEN, "halo.npz"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    job = f"hx{os.getpid()}_{time.monotonic_ns()}"
    procs = [ctx.Process(target=_hx_worker, args=(r, world, q, job, case)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in procs:
            res.append(q.get(timeout=300))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    for rank, msg in sorted(res):
        assert msg == "ok", f"rank {rank}: {msg}"


def _fuzz_worker(rank, world, q, job, n_cases, seed0):
    try:
        sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ.setdefault("OMP_NUM_THREADS", "4"); os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # (several rank processes share the box's cores)
        import torch
        import __graft_entry__ as ge
        pkg = ge.load_package()
        from ultimate_spmv_amd import binding as B
        from oracle import oracle as orc
        torch.cuda.set_device(0)
        hc = pkg.HostComm(job, rank, world, timeout_s=180)
        done = []
        for case in range(n_cases):
            rng = np.random.default_rng(seed0 + case)            # the same matrix on every rank
            n = int(rng.choice([7, 40, 333, 1200, 5000]))
            kind = rng.choice(["banded", "scattered", "ties"])
            lens = rng.integers(1, 12, n) if kind != "ties" else rng.choice([1, 2, 2, 3, 3, 3, 7], n)
            lens = np.minimum(lens, n)
            I = np.repeat(np.arange(n), lens)
            cols = []
            for r, k in enumerate(lens):
                if kind == "banded":
                    lo, hi = max(0, r - 60), min(n, r + 61)
                    k = min(k, hi - lo)
                    cols.append(np.sort(rng.choice(np.arange(lo, hi), k, replace=False)))
                else:
                    cols.append(rng.choice(n, k, replace=False))
            J = np.concatenate(cols)
            I = np.repeat(np.arange(n), [len(c) for c in cols])
            V = rng.standard_normal(I.size) * 10.0 ** rng.integers(-4, 3, I.size)
            C_, sigma = int(rng.choice([4, 16, 32, 32, 64])), int(rng.choice([1, 64, 512]))
            method = str(rng.choice(["seg-rows", "seg-nnz"]))
            tot = pkg.Coo.from_arrays(n, n, I.astype(np.int32), J.astype(np.int32), V)
            wsa = pkg.seg_work_sharing_arr(tot, method, world)
            if np.any(np.diff(wsa) == 0):
                continue                                         # (a rank without rows: refused at conversion, like a matrix without rows)
            xg = rng.standard_normal(n)
            # single-rank product of the whole matrix (row chains do not depend on C / sigma / the partition)
            sg = pkg.convert_to_scs(tot, C_, sigma)
            ag = sg.arrays(); pkg.permute_scs_cols(sg, ag["old_to_new_idx"]); ag = sg.arrays()
            xpg = np.zeros(sg.n_rows_padded); xpg[:n] = pkg.apply_permutation(xg, ag["new_to_old_idx"])
            yg = pkg.apply_permutation(orc.spmv_scs(C_, sg.n_chunks, ag["chunk_ptrs"], ag["chunk_lengths"], ag["col_idxs"], ag["values"], xpg), ag["old_to_new_idx"])[:n]
            loc = B.seg_local_coo(tot, wsa, rank)
            d = pkg.DistNative(loc, wsa, C_, sigma, rank, world, hostcomm=hc, host_exchange=True)
            nl = int(wsa[rank + 1] - wsa[rank])
            want = yg[wsa[rank]:wsa[rank + 1]]
            for overlap, pad, fused in ((1, 0, 0), (0, 0, 0), (1, 1, 0), (1, 1, 1), (1, 0, 1)):
                d.set_option("overlap", overlap); d.set_option("pad_split", pad); d.set_option("fused_step", fused)
                x = d.new_x(xg[wsa[rank]:wsa[rank + 1]]); y = d.new_y()
                d.spmv(x, y); d.spmv(x, y); d.synchronize()
                assert np.array_equal(d.y_to_original_order(y)[:nl], want), (case, kind, n, C_, sigma, method, overlap, pad, fused)
            # block vectors on the same partition: message patterns x layouts, exchange-then-compute and the two-part step, gather kernels
            # and (dp b = 8, C 32 | 64) the phased block plan -- per column against the oracle's single-rank product
            d.set_option("pad_split", 0); d.set_option("fused_step", 0)
            b = int(rng.choice([2, 3, 4, 8, 8]))
            ld = d.padded_vec_size
            Xo = [xg[wsa[rank]:wsa[rank + 1]] * (1.0 + v / 8.0) for v in range(b)]
            wants = []
            for v in range(b):
                xv = np.zeros(sg.n_rows_padded); xv[:n] = pkg.apply_permutation(xg * (1.0 + v / 8.0), ag["new_to_old_idx"])
                yv = orc.spmv_scs(C_, sg.n_chunks, ag["chunk_ptrs"], ag["chunk_lengths"], ag["col_idxs"], ag["values"], xv)
                wants.append(pkg.apply_permutation(yv, ag["old_to_new_idx"])[:n][wsa[rank]:wsa[rank + 1]])
            for plan in ((0, b) if (b == 8 and C_ in (32, 64)) else (0,)):
                if plan: d.set_option("block_plan", plan)
                for overlap in (1, 0):
                    d.set_option("overlap", overlap)
                    for layout, mode in ((pkg.COLWISE, 0), (pkg.COLWISE, 1), (pkg.COLWISE, 2), (pkg.ROWWISE, 0)):
                        X = d.new_X(Xo, b, layout)
                        Y = torch.full((b * ld,), 7.0, dtype=torch.float64, device="cuda")
                        d.spmmv(X, Y, b, layout, mode); d.synchronize()
                        Yh = Y.cpu().numpy()
                        for v in range(b):
                            col = Yh[v:d.n_rows_padded * b:b] if layout == pkg.ROWWISE else Yh[v * ld:v * ld + d.n_rows_padded]
                            got = pkg.apply_permutation(np.ascontiguousarray(col), d.old_to_new)[:nl]
                            assert np.array_equal(got, wants[v]), (case, kind, n, C_, sigma, method, "spmmv", b, plan, overlap, layout, mode, v)
            d.barrier()
            d.close()
            done.append((kind, n, C_, sigma, method))
            if rank == 0:
                print(f"[fuzz] case {case}: {kind} n={n} C={C_} sigma={sigma} {method} b={b} ok", flush=True)
        hc.close()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))


def test_random_matrices_real_ranks_every_step_form(pkg):
    """Random matrices, chunk heights, sorting scopes and partitions (seg-rows / seg-nnz: unequal blocks, asymmetric lists, ranks without
    neighbours) with three real processes on one GPU (host-staged exchange): every arrangement of the step gives, bit for bit, the rows of
    the oracle's single-rank product.  USPMV_FUZZ_DIST_CASES raises the number of matrices."""
    world, n_cases = 3, int(os.environ.get("USPMV_FUZZ_DIST_CASES", "3"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    job = f"fz{os.getpid()}_{time.monotonic_ns()}"
    procs = [ctx.Process(target=_fuzz_worker, args=(r, world, q, job, n_cases, int(os.environ.get("USPMV_FUZZ_DIST_SEED", "424200")))) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in procs:
            res.append(q.get(timeout=300 + 20 * n_cases))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    for rank, msg in sorted(res):
        assert msg == "ok", f"rank {rank}: {msg}"


def test_loopback_refuses_unequal_blocks_cleanly(pkg):
    """seg-nnz blocks of different heights in LOOPBACK: the ids a rank asks block p for index p's rows, not its own -- the set-up
    must say so (USPMV_ERR_INVALID) instead of letting the pack kernel gather out of bounds."""
    import torch
    torch.cuda.set_device(0)
    shape, P = (12, 12, 30), 3
    counts = pkg.gen_stencil27_row_counts(*shape)
    wsa = pkg.seg_from_row_counts(counts, "seg-nnz", P)
    assert len(set(np.diff(wsa).tolist())) > 1, "the case needs unequal blocks"
    small = int(np.argmin(np.diff(wsa)))
    loc = pkg.gen_stencil27(*shape, row_begin=int(wsa[small]), row_end=int(wsa[small + 1]))
    with pytest.raises(pkg.UspmvError) as e:
        pkg.DistNative(loc, wsa, 32, 512, small, P, pkg.comm_unique_id(), comm_rank=0, comm_size=1)
    assert e.value.status == 1 and "equal block heights" in str(e.value)


def test_a_block_without_rows_is_refused_by_every_rank_alike(pkg):
    """work_sharing_arr with an empty block (seg-nnz around a very heavy row can produce one in the middle): every rank refuses at once,
    whichever block it plays -- not only the rank whose conversion would fail."""
    import torch
    torch.cuda.set_device(0)
    wsa = np.array([0, 40, 40, 96], np.int32)
    full = pkg.gen_stencil27(4, 4, 6)
    from ultimate_spmv_amd import binding as B
    for rank in (0, 2):
        loc = B.seg_local_coo(full, wsa, rank)
        with pytest.raises(pkg.UspmvError, match="owns no rows"):
            pkg.DistNative(loc, wsa, 32, 512, rank, 3, pkg.comm_unique_id(), comm_rank=0, comm_size=1)


def test_self_check_passes_in_loopback_and_catches_a_missing_exchange(pkg):
    import torch
    torch.cuda.set_device(0)
    shape, P, rank = (16, 16, 40), 4, 2
    counts = pkg.gen_stencil27_row_counts(*shape)
    wsa = pkg.seg_from_row_counts(counts, "seg-rows", P)
    loc = pkg.gen_stencil27(*shape, row_begin=int(wsa[rank]), row_end=int(wsa[rank + 1]))
    d = pkg.DistNative(loc, wsa, 32, 512, rank, P, pkg.comm_unique_id(), comm_rank=0, comm_size=1)
    x, y = d.new_x(np.zeros(d.n_local)), d.new_y()
    for ba in (0, 1):
        d.set_option("ba_synch", ba)
        bad, _ = d.check(loc, x, y)
        assert bad == 0
    d.set_option("diag_skip_exchange", 1)      # the halo tail stays zero: every boundary row must be reported
    bad, _ = d.check(loc, x, y)
    assert bad > 0
    d.set_option("diag_skip_exchange", 0)
    bad, _ = d.check(loc, x, y)
    assert bad == 0
    d.close()


def test_cli_real_ranks_host_exchange_mtx_scatter_and_check(pkg, tmp_path):
    """`uspmv bcsstk13.mtx scs -c 32 -s 512 -seg_nnz -comm_halos 1` as FOUR real rank processes (rank 0 reads and scatters the .mtx,
    the ranks meet in the host communicator), fixed-step protocol, -ba_synch 1, -check_y 1, JSON report."""
    h = np.load(os.path.join(GOLDEN, "halo.npz"))
    key = "bcsstk13_C32_s512_seg-nnz_P4"
    wsa = h[key + "_wsa"]
    js = str(tmp_path / "out.json")
    procs = []
    for rank in range(4):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="4", LOCAL_RANK="0", USPMV_EXCHANGE="host", USPMV_ID_DIR=str(tmp_path),
                   USPMV_JOB_ID=f"c{os.getpid()}", USPMV_HC_TIMEOUT="120", OMP_NUM_THREADS=os.environ.get("USPMV_TEST_OMP", "4"), OMP_WAIT_POLICY="passive")
        env.pop("USPMV_LOOPBACK", None)
        procs.append(subprocess.Popen([EXE, mtx_path("bcsstk13"), "scs", "-c", "32", "-s", "512", "-seg_nnz", "-comm_halos", "1", "-bench_steps", "5",
                                       "-bench_warmup", "2", "-check_y", "1", "-json", js, "-print_comm_vol", "1"], cwd=tmp_path, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    import json
    rep = json.load(open(js))
    assert rep["y_checked"] is True and rep["y_mismatches"] == 0 and rep["steps"] == 5 and rep["warmup"] == 2
    assert rep["ba_synch"] == 1 and rep["exchange"] == "host" and rep["ranks"] == 4 and rep["nnz"] == 83883
    assert rep["rank0"]["n_local"] == int(wsa[1] - wsa[0]) == 724 and rep["rank0"]["n_halo"] == 220
    assert "y checked bitwise on every rank: ok" in outs[0]
    assert not [f for f in os.listdir(tmp_path) if f.endswith(".uspmvcoo")], "block files must be gone"
    txt = open(tmp_path / "spmv_bench.txt").read()
    assert "seg_method: seg-nnz" in txt and "ba_synch: 1" in txt and "Per rank Elems Recvd" in txt


def test_cli_config5_shaped_workload_three_real_ranks_seg_nnz(pkg, tmp_path):
    """BASELINE config 5's shape on a workload larger than the goldens: `uspmv gen:96x96x96 scs -c 32 -s 512 -dp -seg_nnz -comm_halos 1
    -check_y 1` as THREE real rank processes sharing the GPU with the exchange staged through the host.  (Block k of 8 of the 304^3 matrix
    in RCCL loopback is impossible: seg-nnz blocks are unequal and loopback needs equal heights -- the set-up refuses it,
    test_loopback_refuses_unequal_blocks_cleanly.)  Mirrors the reference's loop code/main.cpp:458-474 over
    init/finalize_halo_exchange, code/classes_structs.hpp:857-995.  Every rank's row of the report is checked against the partition
    rule, and y of every local row bitwise against the entry-ordered chains (uspmv_dist_check)."""
    import json
    g, P = 96, 3
    counts = pkg.gen_stencil27_row_counts(g, g, g)
    wsa = pkg.seg_from_row_counts(counts, "seg-nnz", P)
    heights = np.diff(wsa)
    assert len(set(heights.tolist())) > 1, "seg-nnz blocks of this workload are unequal"
    js = str(tmp_path / "c5.json")
    procs = []
    for rank in range(P):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(P), LOCAL_RANK=str(rank), USPMV_EXCHANGE="host", USPMV_ID_DIR=str(tmp_path),
                   USPMV_JOB_ID=f"c5_{os.getpid()}", USPMV_HC_TIMEOUT="240", USPMV_STAGES="1", OMP_NUM_THREADS="4")
        env.pop("USPMV_LOOPBACK", None)
        procs.append(subprocess.Popen([EXE, f"gen:{g}x{g}x{g}", "scs", "-c", "32", "-s", "512", "-dp", "-seg_nnz", "-comm_halos", "1", "-ba_synch", "0",
                                       "-bench_steps", "20", "-bench_warmup", "5", "-check_y", "1", "-json", js], cwd=tmp_path, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    rep = json.load(open(js))
    assert rep["y_checked"] is True and rep["y_mismatches"] == 0 and rep["steps"] == 20 and rep["ranks"] == P
    assert rep["exchange"] == "host" and rep["rccl_nranks"] == 0 and rep["n_rows"] == g ** 3 and rep["nnz"] == int(counts.sum())
    assert rep["step_form"] in ("overlap", "plain") and set(rep["step_form_candidates_ms"]) == {"overlap", "plain"}
    rows = rep["per_rank"]
    assert [r["rank"] for r in rows] == list(range(P))
    assert [r["n_local"] for r in rows] == heights.tolist()
    assert sum(r["nnz"] for r in rows) == rep["nnz"]
    # a slab of a 96^3 27-point stencil needs whole planes from its neighbours (+ the padding column 0 from rank 0, code/mpi_funcs.hpp:279-306)
    assert rows[0]["n_halo"] >= g * g and rows[1]["n_halo"] >= 2 * g * g and rows[2]["n_halo"] >= g * g
    assert all(r["local_kernel_ms"] > 0 and r["interior"] > 0 and r["boundary"] > 0 for r in rows)
    assert rep["rank0"]["n_local"] == rows[0]["n_local"] and rep["rank0"]["n_halo"] == rows[0]["n_halo"]
    assert "[uspmv stage] timed region done" in outs[0] and "[uspmv stage] report written" in outs[0]


def test_cli_solve_mode_and_crs_across_real_ranks(pkg, orc, tmp_path):
    """`uspmv bcsstk13.mtx <scs -c 32 -s 512 | crs> <-dp | -sp> -mode s -rev 3 -seg_nnz` as three real rank processes (the multi-rank
    half of the reference's scripts/validate_master.sh): COMM - spmv - SWAP three times, every rank dumps y of its rows.  Each row is
    the same entry-ordered chain as in a single-rank run, so the dumps must equal, bit for bit, three applications of the oracle's
    product (in the run's precision) to the concatenated per-rank ramps (USPMV_DIST_X=ramp)."""
    tot = pkg.read_mtx(mtx_path("bcsstk13"))
    n, P = tot.n_rows, 3
    wsa = pkg.seg_work_sharing_arr(tot, "seg-nnz", P)
    for prec, code, ndt in (("-dp", pkg.F64, np.float64), ("-sp", pkg.F32, np.float32)):
        xg = np.concatenate([1.0 + 1e-3 * (np.arange(int(wsa[r + 1] - wsa[r])) % 1000) for r in range(P)]).astype(ndt)
        s = pkg.convert_to_scs(tot, 32, 512, code)
        a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"]); a = s.arrays()
        y = xg
        for _ in range(3):
            xp = np.zeros(s.n_rows_padded, ndt); xp[:n] = pkg.apply_permutation(y, a["new_to_old_idx"])
            y = pkg.apply_permutation(orc.spmv_scs(32, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], xp), a["old_to_new_idx"])[:n]
        assert y.dtype == ndt
        for fmt in (["scs", "-c", "32", "-s", "512"], ["crs"]):
            pre = str(tmp_path / ("y_" + fmt[0] + prec))
            procs = []
            for rank in range(P):
                env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(P), LOCAL_RANK="0", USPMV_EXCHANGE="host", USPMV_ID_DIR=str(tmp_path),
                           USPMV_JOB_ID=f"s{os.getpid()}{fmt[0]}{prec}", USPMV_HC_TIMEOUT="120", USPMV_DIST_X="ramp", OMP_NUM_THREADS=os.environ.get("USPMV_TEST_OMP", "4"), OMP_WAIT_POLICY="passive")
                env.pop("USPMV_LOOPBACK", None)
                procs.append(subprocess.Popen([EXE, mtx_path("bcsstk13")] + fmt + [prec, "-mode", "s", "-rev", "3", "-seg_nnz", "-comm_halos", "1", "-check_y", "1",
                                               "-dump_y", pre], cwd=tmp_path, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
            outs = [p.communicate(timeout=300)[0] for p in procs]
            for p, o in zip(procs, outs):
                assert p.returncode == 0, o
            assert "solve mode: 3 revision(s) done on 3 ranks, y checked bitwise on every rank: ok" in outs[0], outs[0]
            for rank in range(P):
                got = np.fromfile(pre + f".{rank}", ndt)
                assert np.array_equal(got, y[wsa[rank]:wsa[rank + 1]]), (prec, fmt[0], rank)
    # bench mode in single precision across ranks: the fixed-step protocol with the self-check
    procs = []
    for rank in range(P):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(P), LOCAL_RANK="0", USPMV_EXCHANGE="host", USPMV_ID_DIR=str(tmp_path),
                   USPMV_JOB_ID=f"b{os.getpid()}", USPMV_HC_TIMEOUT="120", OMP_NUM_THREADS=os.environ.get("USPMV_TEST_OMP", "4"), OMP_WAIT_POLICY="passive")
        env.pop("USPMV_LOOPBACK", None)
        procs.append(subprocess.Popen([EXE, mtx_path("bcsstk13"), "scs", "-c", "32", "-s", "512", "-sp", "-seg_nnz", "-comm_halos", "1", "-bench_steps", "5",
                                       "-bench_warmup", "2", "-check_y", "1"], cwd=tmp_path, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    assert "y checked bitwise on every rank: ok" in outs[0], outs[0]
    assert "data_type: float" in open(tmp_path / "spmv_bench.txt").read()


def test_cli_rand_x_across_real_ranks(pkg, orc, tmp_path):
    """-rand_x 1 | m across ranks as in the reference: min / max of |values| over the WHOLE matrix (rank 0 extracts and broadcasts,
    code/utilities.hpp:2502-2540), then every rank draws the SAME default-seeded mt19937 sequence for its padded local vector (:880-912).
    The expected x is rebuilt here from the generator's definition (numpy's legacy MT19937 stream = std::mt19937(5489); generate_canonical
    = two draws; the product-sum fused, exact arithmetic in fractions), y from the oracle."""
    from fractions import Fraction
    tot = pkg.read_mtx(mtx_path("bcsstk13"))
    n, P = tot.n_rows, 3
    wsa = pkg.seg_work_sharing_arr(tot, "seg-nnz", P)
    av = np.abs(tot.arrays()[2])
    vmin, vmax = float(av.min()), float(av.max())
    s = pkg.convert_to_scs(tot, 32, 512)
    a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"]); a = s.arrays()
    raw = np.random.RandomState(5489).randint(0, 2 ** 32, size=2 * 1024, dtype=np.uint32).astype(np.float64)
    u = (raw[0::2] + raw[1::2] * 4294967296.0) / 18446744073709551616.0            # generate_canonical<double, 53> on a 32-bit engine
    seq = np.array([float(Fraction(float(x)) * Fraction(vmax - vmin) + Fraction(vmin)) for x in u])   # fma(u, max - min, min)
    assert max(int(wsa[r + 1] - wsa[r]) for r in range(P)) <= seq.size
    for rx, xg in (("1", np.concatenate([seq[:int(wsa[r + 1] - wsa[r])] for r in range(P)])), ("m", np.full(n, vmin + (vmax - vmin) / 2.0))):
        xp = np.zeros(s.n_rows_padded); xp[:n] = pkg.apply_permutation(xg, a["new_to_old_idx"])
        y = pkg.apply_permutation(orc.spmv_scs(32, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], xp), a["old_to_new_idx"])[:n]
        pre = str(tmp_path / ("y_rx" + rx))
        procs = []
        for rank in range(P):
            env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(P), LOCAL_RANK="0", USPMV_EXCHANGE="host", USPMV_ID_DIR=str(tmp_path),
                       USPMV_JOB_ID=f"x{os.getpid()}{rx}", USPMV_HC_TIMEOUT="120", OMP_NUM_THREADS=os.environ.get("USPMV_TEST_OMP", "4"), OMP_WAIT_POLICY="passive")
            env.pop("USPMV_LOOPBACK", None); env.pop("USPMV_DIST_X", None)
            procs.append(subprocess.Popen([EXE, mtx_path("bcsstk13"), "scs", "-c", "32", "-s", "512", "-mode", "s", "-rev", "1", "-rand_x", rx, "-seg_nnz",
                                           "-comm_halos", "1", "-dump_y", pre], cwd=tmp_path, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        outs = [p.communicate(timeout=300)[0] for p in procs]
        for p, o in zip(procs, outs):
            assert p.returncode == 0, o
        for rank in range(P):
            got = np.fromfile(pre + f".{rank}", np.float64)
            assert np.array_equal(got, y[wsa[rank]:wsa[rank + 1]]), (rx, rank, np.abs(got - y[wsa[rank]:wsa[rank + 1]]).max())


def test_cli_equilibrate_across_real_ranks(pkg, orc, tmp_path):
    """-equilibrate 1 across ranks: every rank scales ITS block (rows, then the columns of the row-scaled block) after the segmentation,
    as the reference does (code/main.cpp:1117-1125 on local_mtx).  Expected: the blocks scaled one by one with numpy (division by the
    row / column maxima: order-independent, exact), stacked, through the oracle's product."""
    from ultimate_spmv_amd import binding as B
    tot = pkg.read_mtx(mtx_path("bcsstk13"))
    n, P = tot.n_rows, 3
    wsa = pkg.seg_work_sharing_arr(tot, "seg-nnz", P)
    I, J, V = (np.array(v) for v in tot.arrays())
    Veq = V.copy()
    for r in range(P):
        sel = (I >= wsa[r]) & (I < wsa[r + 1])
        v, ii, jj = Veq[sel], I[sel], J[sel]
        rmax = np.zeros(n); np.maximum.at(rmax, ii, np.abs(v)); v = v / rmax[ii]
        cmax = np.zeros(n); np.maximum.at(cmax, jj, np.abs(v)); v = v / cmax[jj]
        Veq[sel] = v
    eq = pkg.Coo.from_arrays(n, n, I, J, Veq)
    s = pkg.convert_to_scs(eq, 32, 512)
    a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"]); a = s.arrays()
    xg = np.concatenate([1.0 + 1e-3 * (np.arange(int(wsa[r + 1] - wsa[r])) % 1000) for r in range(P)])
    xp = np.zeros(s.n_rows_padded); xp[:n] = pkg.apply_permutation(xg, a["new_to_old_idx"])
    y = pkg.apply_permutation(orc.spmv_scs(32, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], xp), a["old_to_new_idx"])[:n]
    pre = str(tmp_path / "y_eq")
    procs = []
    for rank in range(P):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(P), LOCAL_RANK="0", USPMV_EXCHANGE="host", USPMV_ID_DIR=str(tmp_path),
                   USPMV_JOB_ID=f"e{os.getpid()}", USPMV_HC_TIMEOUT="120", USPMV_DIST_X="ramp", OMP_NUM_THREADS=os.environ.get("USPMV_TEST_OMP", "4"), OMP_WAIT_POLICY="passive")
        env.pop("USPMV_LOOPBACK", None)
        procs.append(subprocess.Popen([EXE, mtx_path("bcsstk13"), "scs", "-c", "32", "-s", "512", "-mode", "s", "-rev", "1", "-equilibrate", "1", "-seg_nnz",
                                       "-comm_halos", "1", "-check_y", "1", "-dump_y", pre], cwd=tmp_path, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    assert "y checked bitwise on every rank: ok" in outs[0]
    for rank in range(P):
        got = np.fromfile(pre + f".{rank}", np.float64)
        assert np.array_equal(got, y[wsa[rank]:wsa[rank + 1]]), (rank, np.abs(got - y[wsa[rank]:wsa[rank + 1]]).max())


def test_cli_loopback_graph_replay_with_ba_synch_and_check(pkg, tmp_path):
    """the captured step now also carries the per-step barrier (-ba_synch 1, the reference's default) and the self-check runs
    through the replayed graph"""
    env = dict(os.environ, USPMV_LOOPBACK="2", USPMV_LOOPBACK_RANK="1", USPMV_ID_DIR=str(tmp_path), USPMV_JOB_ID=f"g{os.getpid()}", OMP_NUM_THREADS=os.environ.get("USPMV_TEST_OMP", "4"), OMP_WAIT_POLICY="passive")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    for ba in ("1", "0"):
        r = subprocess.run([EXE, "gen:24x24x24", "scs", "-c", "32", "-s", "512", "-seg_rows", "-comm_halos", "1", "-bench_time", "0.05", "-ba_synch", ba,
                            "-check_y", "1", "-json", "-"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        import json
        rep = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert rep["graph_replay"] is True and rep["y_checked"] is True and rep["ba_synch"] == int(ba) and rep["loopback"] is True
        assert rep["versions"]["rccl_runtime"] >= rep["versions"]["rccl_build"] > 0


def test_cli_step_forms_all_check_and_auto_picks_one(pkg, tmp_path):
    """-step_form: every arrangement of the step (overlap | plain | pad | fused) passes the bitwise self-check through the harness, and
    auto times overlap | plain on this machine, reports them and keeps the fastest; auto_all adds the speculative pad / fused forms
    (graph replay: without the one-launch form)."""
    import json
    import subprocess
    env = dict(os.environ, USPMV_LOOPBACK="4", USPMV_LOOPBACK_RANK="2", USPMV_ID_DIR=str(tmp_path), USPMV_JOB_ID="sf", OMP_NUM_THREADS=os.environ.get("USPMV_TEST_OMP", "4"), OMP_WAIT_POLICY="passive")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    base = [EXE, "gen:32x32x64", "scs", "-c", "32", "-s", "512", "-seg_rows", "-comm_halos", "1", "-bench_steps", "20", "-bench_warmup", "3", "-check_y", "1"]
    for form, graph in (("overlap", 1), ("plain", 1), ("pad", 1), ("pad", 0), ("fused", 0)):
        js = tmp_path / f"{form}{graph}.json"
        r = subprocess.run(base + ["-step_form", form, "-graph", str(graph), "-json", str(js)], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        rep = json.load(open(js))
        assert rep["step_form"] == form and rep["y_checked"] is True and rep["y_mismatches"] == 0, rep
        assert rep["overlap"] == (form != "plain") and rep["graph_replay"] == bool(graph)
        assert f"step form: {form}" in r.stdout
    for graph, sf, cands in ((1, "auto_all", {"overlap", "plain", "pad"}), (0, "auto_all", {"overlap", "plain", "pad", "fused"}),
                             (1, "auto", {"overlap", "plain"}), (0, "auto", {"overlap", "plain"})):
        js = tmp_path / f"{sf}{graph}.json"
        r = subprocess.run(base + ["-graph", str(graph), "-step_form", sf, "-json", str(js)], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        rep = json.load(open(js))
        assert set(rep["step_form_candidates_ms"]) == cands and rep["step_form"] in cands, rep
        assert rep["step_form_candidates_ms"][rep["step_form"]] == min(rep["step_form_candidates_ms"].values())
        assert rep["y_checked"] is True and rep["y_mismatches"] == 0
    r = subprocess.run(base + ["-step_form", "sideways"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "step_form must be" in (r.stdout + r.stderr)


def test_cli_seg_metis_real_ranks(pkg, tmp_path):
    """-seg_metis through the harness (rank 0 partitions the matrix graph with the built-in partitioner -- METIS is not linked --, sorts the
    rows by part, permutes the matrix symmetrically and scatters the blocks; code/mpi_funcs.hpp:494-598), three real ranks, self-checked;
    then the same with the part vector coming from a gpmetis-style file."""
    m = pkg.read_mtx(mtx_path("bcsstk13"))
    part = pkg.graph_partition(m, 3)
    pf = tmp_path / "bcsstk13.part.3"
    pf.write_text("\n".join(str(int(v)) for v in part) + "\n")
    sizes = np.bincount(part, minlength=3)
    import json
    for extra in ([], ["-part_file", str(pf)]):
        js = str(tmp_path / f"out{len(extra)}.json")
        procs = []
        for rank in range(3):
            env = dict(os.environ, RANK=str(rank), WORLD_SIZE="3", LOCAL_RANK="0", USPMV_EXCHANGE="host", USPMV_ID_DIR=str(tmp_path),
                       USPMV_JOB_ID=f"m{os.getpid()}_{len(extra)}", USPMV_HC_TIMEOUT="120", OMP_NUM_THREADS=os.environ.get("USPMV_TEST_OMP", "4"), OMP_WAIT_POLICY="passive")
            env.pop("USPMV_LOOPBACK", None)
            procs.append(subprocess.Popen([EXE, mtx_path("bcsstk13"), "scs", "-c", "32", "-s", "512", "-seg_metis", "-comm_halos", "1", "-bench_steps", "3",
                                           "-bench_warmup", "1", "-check_y", "1", "-json", js] + extra, cwd=tmp_path, env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        outs = [p.communicate(timeout=300)[0] for p in procs]
        for p, o in zip(procs, outs):
            assert p.returncode == 0, o
        rep = json.load(open(js))
        assert rep["y_checked"] is True and rep["y_mismatches"] == 0 and rep["ranks"] == 3
        assert rep["rank0"]["n_local"] == int(sizes[0])
        assert "seg-metis: METIS is not linked" in outs[0]
    assert "seg_method: seg-metis" in open(tmp_path / "spmv_bench.txt").read()
    # The self-check runs in the PERMUTED numbering (a wrong symmetric permutation would still pass it): dump y with the ramp x, put it
    # back into the original numbering with <prefix>.perm and compare with the single-rank product of the ORIGINAL matrix -- the
    # permutation keeps the entry order inside a row, so every row's FMA chain is the same and the comparison is bitwise.
    from oracle import oracle as orc
    procs = []
    for rank in range(3):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="3", LOCAL_RANK="0", USPMV_EXCHANGE="host", USPMV_ID_DIR=str(tmp_path), USPMV_JOB_ID=f"mp{os.getpid()}",
                   USPMV_HC_TIMEOUT="120", USPMV_DIST_X="ramp", USPMV_DUMP_Y=str(tmp_path / "ymetis"), OMP_NUM_THREADS=os.environ.get("USPMV_TEST_OMP", "4"), OMP_WAIT_POLICY="passive")
        env.pop("USPMV_LOOPBACK", None)
        procs.append(subprocess.Popen([EXE, mtx_path("bcsstk13"), "scs", "-c", "32", "-s", "512", "-seg_metis", "-comm_halos", "1", "-bench_steps", "1", "-bench_warmup", "0"],
                                      cwd=tmp_path, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p_ in procs:
        o = p_.communicate(timeout=300)[0]
        assert p_.returncode == 0, o
    perm = np.fromfile(tmp_path / "ymetis.perm", np.int32)
    assert np.array_equal(np.sort(perm), np.arange(m.n_rows))
    _, wsa, _ = pkg.apply_partition(m, 3, part)
    y_new = np.concatenate([np.fromfile(tmp_path / f"ymetis.{r}") for r in range(3)])
    x_new = np.concatenate([1.0 + 1e-3 * (np.arange(int(wsa[r + 1] - wsa[r])) % 1000) for r in range(3)])     # USPMV_DIST_X=ramp: per rank, local numbering
    x_orig = np.empty(m.n_rows); x_orig[perm] = x_new
    I, J, V = m.arrays()
    s1 = pkg.convert_to_scs(m, 1, 1, pkg.F64)                                                             # (C = 1, sigma = 1: rows in original order, chains in entry order)
    a1 = s1.arrays()
    y_orig = orc.spmv_scs(1, s1.n_chunks, a1["chunk_ptrs"], a1["chunk_lengths"], a1["col_idxs"], a1["values"], x_orig)[:m.n_rows]
    assert np.array_equal(y_new, y_orig[perm])


def test_bench_launcher_starts_real_rank_processes_and_reports_every_rank(pkg, tmp_path):
    """`python3 bench.py --gpus 2` with NO outer launcher, the real `uspmv` harness on the GPU(s) of this box: the bench process starts
    the two rank processes itself, keeps its budget and prints one JSON line.  On a one-GPU box RCCL refuses two ranks on one device
    (tier 1 fails cleanly, tier 3 -- exchange staged through host memory -- measures); on a multi-GPU box tier 1 measures.  Either way:
    y of every local row checked bitwise, every rank's row in the line, the one-GPU time of the same matrix and the efficiency."""
    import json
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "USPMV_LOOPBACK", "USPMV_EXCHANGE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3", "--grid", "72", "--grid2", "48",
                        "--budget-s", "300"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=400)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    d = json.loads(lines[0])
    c = d["config"]
    assert d["value"] > 0 and d["n_gpus"] == 2 and d["scaling"] == "strong" and d["y_checked"] is True and d["y_mismatches"] == 0
    assert "itself" in c["launcher"] and [x["rank"] for x in c["per_rank"]] == [0, 1]
    assert sum(x["n_local"] for x in c["per_rank"]) == 72 ** 3 == c["n_rows"]
    assert all(x["local_kernel_ms"] > 0 and x["n_halo"] >= 72 * 72 for x in c["per_rank"])
    if c["exchange"] == "host":          # one GPU: the communicator was refused, and the line says so
        assert c["rccl_nranks"] == 0 and "fallback_reason" in d and [t["ok"] for t in d["budget"]["tiers"]][:2] == [False, True]
    else:                                # several GPUs: the production path
        assert c["exchange"] == "rccl" and c["rccl_nranks"] == 2 and d["budget"]["tiers"][0]["ok"] is True
    s1 = d["single_gpu_same_matrix"]
    assert s1["n_rows"] == 72 ** 3 and s1["ms_per_step"] > 0
    assert d["strong_scaling_efficiency"] == pytest.approx(s1["ms_per_step"] / (2 * d["ms_per_step"]), rel=1e-3)
    assert d["weak_scaling"]["value"] > 0 and d["weak_scaling"]["y_checked"] is True
    assert d["budget"]["used_s"] < 300
