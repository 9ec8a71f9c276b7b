"""BASELINE configurations 3, 4 and 4b at FULL size (SURVEY.md 8(d) stand-ins), HIP path against the oracle bit for bit, plus a
size-independent property (linearity in x: A(2x) == 2 A x exactly).  Config 2 at full size lives in
test_gpu_parity.py::test_full_size_nlpkkt200_class.  USPMV_FULL_SCALE < 1 shrinks the grids for quick local runs."""
import os
import time

import numpy as np
import pytest

from conftest import make_x

pytestmark = pytest.mark.gpu
SCALE = float(os.environ.get("USPMV_FULL_SCALE", "1.0"))


@pytest.fixture(scope="module")
def torch_cuda(pkg):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    torch.cuda.set_device(0)
    yield torch
    pkg.set_tuning(spmmv_variant=0, sweep=1, tlc=1)


def test_full_size_config3_queen_class_spmmv(pkg, orc, torch_cuda):
    """Queen_4147-class: 27-pt stencil, 3 dof per node on 111^3 nodes (n = 4.1 M, nnz = 3.26e8), SELL-32-512 dp, -block_vec_size 8,
    both block-vector layouts: the block-plan kernel uspmv_spmmv picks by itself == block_spmv_omp_scs_general (code/kernels.hpp:306-398)."""
    t = torch_cuda
    g = max(8, int(111 * SCALE))
    t0 = time.time()
    coo = pkg.gen_stencil27(g, g, g, dof=3)
    s = pkg.convert_to_scs(coo, 32, 512, pkg.F64)
    a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"]); a = s.arrays()
    b, ld = 8, s.n_rows_padded
    A = pkg.DeviceMatrix(s, block_tlc=b)
    assert A.block_staged == A.block_tiles > 0
    xp = np.zeros(ld); xp[:s.n_rows] = pkg.apply_permutation(make_x(s.n_rows), a["new_to_old_idx"])
    print(f"\n[cfg3] n={s.n_rows} nnz={s.nnz} set-up {time.time() - t0:.1f}s, block plan {A.block_staged}/{A.block_tiles} tiles", flush=True)
    for lay, rowwise in ((pkg.COLWISE, 0), (pkg.ROWWISE, 1)):
        X = np.zeros(b * ld)
        for v in range(b):
            col = xp * (1.0 + v / 8.0)
            if rowwise: X[np.arange(ld) * b + v] = col
            else: X[v * ld:(v + 1) * ld] = col
        dX = t.from_numpy(X).cuda(); dY = t.zeros(b * ld, dtype=t.float64, device="cuda")
        pkg.spmmv(A, dX, dY, b, ld, lay)
        t1 = time.time()
        Yo = orc.spmmv_scs(32, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], X, b, ld, rowwise)
        assert np.array_equal(dY.cpu().numpy(), Yo), ("layout", rowwise)
        dY2 = t.zeros_like(dY)
        pkg.spmmv(A, 2.0 * dX, dY2, b, ld, lay)
        assert t.equal(dY2, 2.0 * dY)
        print(f"[cfg3] {'rowwise' if rowwise else 'colwise'} bit-exact vs oracle (oracle {time.time() - t1:.1f}s)", flush=True)


def _ap_case(pkg, orc, t, coo, tag):
    t0 = time.time()
    dp, sp = pkg.partition_precisions(coo, 1e-3)
    assert dp.nnz > 0 and sp.nnz > 0
    ds = pkg.convert_to_scs(dp, 32, 512, pkg.F64)
    perm = ds.arrays()["old_to_new_idx"].copy()
    ss = pkg.convert_to_scs(sp, 32, 512, pkg.F32, fixed_permutation=perm)
    pkg.permute_scs_cols(ds, perm); pkg.permute_scs_cols(ss, perm)
    da, sa = ds.arrays(), ss.arrays()
    Ad, As = pkg.DeviceMatrix(ds), pkg.DeviceMatrix(ss)
    nt, nst = pkg.optimize_ap(Ad, As, ds, ss)          # tile-local-column plan, or the column-window sweep where that stages nothing
    kind, tiles, planned = Ad.plan_info()
    xp = np.zeros(ds.n_rows_padded); xp[:ds.n_rows] = pkg.apply_permutation(make_x(ds.n_rows), da["new_to_old_idx"])
    x = t.from_numpy(xp).cuda(); y = t.zeros(ds.n_rows_padded, dtype=t.float64, device="cuda")
    pkg.spmv_ap(Ad, As, x, y)
    print(f"\n[{tag}] n={ds.n_rows} dp nnz={dp.nnz} sp nnz={sp.nnz} set-up {time.time() - t0:.1f}s, plan kind {kind}: {planned}/{tiles} tiles", flush=True)
    y_or = orc.spmv_scs_ap_adv(32, ds.n_chunks, (da["chunk_ptrs"], da["chunk_lengths"], da["col_idxs"], da["values"]),
                               (sa["chunk_ptrs"], sa["chunk_lengths"], sa["col_idxs"], sa["values"]), xp)
    assert np.array_equal(y.cpu().numpy(), y_or)
    y2 = t.zeros_like(y)
    pkg.spmv_ap(Ad, As, 2.0 * x, y2)
    assert t.equal(y2, 2.0 * y)
    return kind, tiles, planned


def test_full_size_config4_hv15r_class_stencil_ap(pkg, orc, torch_cuda):
    """HV15R-class, regular variant: 5 dof per node on 74^3 nodes (n = 2.0 M, nnz = 2.7e8), magnitudes over 10 decades,
    -ap[dp_sp] -ap_threshold_1 1e-3: shared tile-local-column plan == scs_ap_impl_cpu<32> (code/ap_kernels.hpp:24-82)."""
    g = max(6, int(74 * SCALE))
    kind, tiles, planned = _ap_case(pkg, orc, torch_cuda, pkg.gen_stencil27(g, g, g, dof=5, magnitude_decades=10.0), "cfg4")
    assert kind == 1 and planned == tiles


def test_full_size_config4b_hv15r_class_banded_random_ap(pkg, orc, torch_cuda):
    """HV15R-class as SURVEY.md 8(d) specifies it: n = 2 017 169, 140 entries per row scattered over a +-50 000 band, 10 decades,
    -ap[dp_sp] -ap_threshold_1 1e-3.  No tile fits LDS, uspmv_dmat_optimize_ap installs the column-window sweep: every tile qualifies."""
    n = max(200000, int(2017169 * SCALE ** 3))
    kind, tiles, planned = _ap_case(pkg, orc, torch_cuda, pkg.gen_banded_random(n, 140, 50000, magnitude_decades=10.0), "cfg4b")
    assert kind == 2 and planned == tiles


def test_full_size_config4b_plain_dp_sweep(pkg, orc, torch_cuda):
    """The same banded-random matrix in plain double precision through uspmv_dmat_optimize: sweep kernel == scs_impl_cpu<32>."""
    t = torch_cuda
    n = max(200000, int(2017169 * SCALE ** 3))
    coo = pkg.gen_banded_random(n, 140, 50000, magnitude_decades=10.0)
    s = pkg.convert_to_scs(coo, 32, 512, pkg.F64)
    a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"]); a = s.arrays()
    A = pkg.DeviceMatrix(s, tlc=True)
    kind, tiles, planned = A.plan_info()
    assert kind == 2 and planned == tiles
    xp = np.zeros(s.n_rows_padded); xp[:s.n_rows] = pkg.apply_permutation(make_x(s.n_rows), a["new_to_old_idx"])
    y = t.zeros(s.n_rows_padded, dtype=t.float64, device="cuda")
    pkg.spmv(A, t.from_numpy(xp).cuda(), y)
    assert np.array_equal(y.cpu().numpy(), orc.spmv_scs(32, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], xp))


def test_full_size_kkt_nlpkkt200_class(pkg, orc, torch_cuda):
    """The KKT-structured member of the nlpkkt class (uspmv_gen_kkt, N = 200: n = 16 240 000 = nlpkkt200's size, nnz = 4.3e8, rows of
    5-28 entries, every state / multiplier row reaching into two index ranges N^3 apart): SELL-32-512 dp through the plan uspmv_dmat_optimize
    picks, bit for bit against scs_impl_cpu<32>'s restatement, plus linearity.  Shows the headline kernel off the friendly end of its class."""
    t = torch_cuda
    from ultimate_spmv_amd import binding as B
    N = max(6, int(200 * SCALE))
    t0 = time.time()
    coo = pkg.gen_kkt(N)
    assert coo.n_rows == 2 * N ** 3 + 6 * N ** 2
    s = pkg.convert_to_scs(coo, 32, 512, pkg.F64)
    a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"]); a = s.arrays()
    nnz = coo.nnz
    del coo
    A = pkg.DeviceMatrix(s, tlc=True)
    kind, nt, npl = A.plan_info()
    xp = np.zeros(s.n_rows_padded); xp[:s.n_rows] = pkg.apply_permutation(make_x(s.n_rows), a["new_to_old_idx"])
    x = t.from_numpy(xp).cuda(); y = t.zeros(s.n_rows_padded, dtype=t.float64, device="cuda")
    pkg.spmv(A, x, y)
    t.cuda.synchronize()
    ms = B.time_launches(0, 30, A=A, x=x, y=y)
    byts = s.n_elements * 12 + 8 * s.n_chunks + 8 * (s.n_rows + s.n_rows_padded)
    print(f"\n[kkt] N={N} n={s.n_rows} nnz={nnz} beta={nnz / s.n_elements:.4f} set-up {time.time() - t0:.1f}s plan kind {kind}: {npl}/{nt} tiles; "
          f"{ms:.4f} ms = {2.0 * nnz / ms / 1e6:.0f} GF/s, {byts / ms / 1e6:.0f} GB/s algorithmic = {byts / ms / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
    yo = orc.spmv_scs(32, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], xp)
    assert np.array_equal(y.cpu().numpy(), yo)
    y2 = t.zeros_like(y)
    pkg.spmv(A, 2.0 * x, y2)
    assert t.equal(y2, 2.0 * y)
    A0 = pkg.DeviceMatrix(s)            # the plain gather kernel on the same arrays
    y3 = t.zeros_like(y)
    pkg.spmv(A0, x, y3)
    assert t.equal(y3, y)
    print(f"[kkt] bit-exact vs oracle; gather kernel {B.time_launches(0, 30, A=A0, x=x, y=y3):.4f} ms", flush=True)
