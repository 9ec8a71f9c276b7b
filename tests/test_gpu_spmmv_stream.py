"""The phased SpMMV plan walked as a stream by persistent workgroups (tuning "spmmv_stream", csrc/spmmv_stream.hip): bit-identical to the reference's
block_spmv_omp_scs_general (code/kernels.hpp:306-398) in both block-vector layouts, for every number of workgroups per CU, both tile -> workgroup
mappings, both prefetch depths (one phase ahead with a full wait per phase; two phases ahead with partial waits), rows with partial last groups, empty rows and tiles, leading dimensions beyond the padded rows, special values; shapes the streaming
kernel does not cover (C != 32) fall back to the one-tile-per-workgroup kernel with the same bits."""
import numpy as np
import pytest

from conftest import make_x, mtx_path

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def t(pkg):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.cuda.set_device(0)
    return torch


def prep(pkg, m, C, sigma, dtype):
    s = pkg.convert_to_scs(m, C, sigma, dtype)
    a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"]); a = s.arrays()
    xp = np.zeros(s.n_rows_padded, a["values"].dtype)
    xp[:s.n_rows] = pkg.apply_permutation(make_x(s.n_rows).astype(a["values"].dtype), a["new_to_old_idx"])
    return s, a, xp


def block(xp, b, ld, rowwise):
    X = np.zeros(b * ld, xp.dtype)
    for v in range(b):
        col = xp * xp.dtype.type(1.0 + v / 8.0)
        if rowwise:
            X[v::b] = col
        else:
            X[v * ld:v * ld + xp.size] = col
    return X


def matrix(pkg, name):
    if name == "stencil3":
        return pkg.gen_stencil27(14, 13, 12, dof=3)
    if name == "stencil1":
        return pkg.gen_stencil27(24, 20, 18)
    if name == "mesh2d":
        return pkg.gen_stencil27(60, 50, 1, dof=2)
    if name == "band":
        return pkg.gen_banded_random(3000, 9, 60, magnitude_decades=3.0)
    return pkg.read_mtx(mtx_path(name))


CASES = [("stencil3", 32, 512, 4, 1, 1), ("stencil3", 32, 512, 3, 0, 2), ("stencil3", 32, 1, 1, 1, 2), ("stencil3", 32, 64, 5, 1, 1), ("stencil1", 32, 512, 4, 1, 2),
         ("mesh2d", 32, 128, 2, 1, 1), ("band", 32, 64, 4, 1, 2), ("FDM-2d-16", 32, 16, 4, 1, 2), ("impcol_e", 32, 64, 4, 0, 2), ("impcol_e", 32, 64, 4, 0, 1),
         ("stencil3", 64, 128, 4, 1, 1), ("stencil3", 32, 512, 99, 1, 1), ("mesh2d", 32, 128, 99, 1, 1), ("impcol_e", 32, 64, 99, 1, 1)]


@pytest.mark.parametrize("name,C,sigma,wgs,by_xcd,depth", CASES)
@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_stream_kernel_bitexact_both_layouts(pkg, orc, t, name, C, sigma, wgs, by_xcd, depth, dt):
    m = matrix(pkg, name)
    dtype = pkg.F64 if dt == "f64" else pkg.F32
    b = 8 if dt == "f64" else 16
    s, a, xp = prep(pkg, m, C, sigma, dtype)
    pkg.set_tuning(spmmv_stream=wgs, spmmv_stream_xcd=by_xcd, spmmv_stream_depth=depth)
    try:
        A = pkg.DeviceMatrix(s, block_tlc=b)
        info = A.block_plan_info()
        if C == 32 and info["phased_plan"] and info["idx8"]:
            assert info["stream_grid"] > 0 and info["stream_descriptors"] >= info["phases"]      # (the kernel under test is the one that answers)
        else:
            assert info["stream_grid"] == 0                                                      # (shape not covered: the other kernels answer)
        for ld in (s.n_rows_padded, s.n_rows_padded + 24):
            for rowwise in (True, False):
                if rowwise and ld != s.n_rows_padded:
                    continue
                X = block(xp, b, ld, rowwise)
                Yo = orc.spmmv_scs(C, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], X, b, ld, rowwise)
                dX = t.from_numpy(X).cuda(); dY = t.full((b * ld,), -7.0, dtype=dX.dtype, device="cuda")
                pkg.spmmv(A, dX, dY, b, ld, pkg.ROWWISE if rowwise else pkg.COLWISE)
                t.cuda.synchronize()
                got = dY.cpu().numpy()
                n = s.n_rows_padded
                if rowwise:
                    assert np.array_equal(got[:n * b], Yo[:n * b]), (name, C, sigma, rowwise)
                else:
                    for v in range(b):
                        assert np.array_equal(got[v * ld:v * ld + n], Yo[v * ld:v * ld + n]), (name, C, sigma, v, ld)
                        assert np.all(got[v * ld + n:(v + 1) * ld] == -7.0)                      # rows beyond the matrix are left alone
        del A
    finally:
        pkg.set_tuning(spmmv_stream=0, spmmv_stream_xcd=1, spmmv_stream_depth=1)


def test_stream_kernel_on_the_device_built_plan_and_special_values(pkg, orc, t):
    """A handle planned from its device arrays alone (uspmv_dmat_optimize_block_device: the function-pointer launchers' path) gets the schedule too;
    signed zeros, infinities and NaN in X travel like in the reference."""
    m = pkg.gen_stencil27(11, 10, 9, dof=3)
    s, a, xp = prep(pkg, m, 32, 512, pkg.F64)
    b, ld = 8, s.n_rows_padded
    rng = np.random.default_rng(11)
    X = block(xp, b, ld, True)
    sp = rng.choice(X.size, 200, replace=False)
    X[sp[:50]] = np.inf; X[sp[50:100]] = -np.inf; X[sp[100:150]] = np.nan; X[sp[150:]] = -0.0
    Yo = orc.spmmv_scs(32, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], X, b, ld, True)
    pkg.set_tuning(spmmv_stream=4, spmmv_stream_depth=2)
    try:
        A = pkg.DeviceMatrix(s)
        A.optimize_block_device(b)
        info = A.block_plan_info()
        assert info["device_built"] and info["stream_grid"] > 0
        dX = t.from_numpy(X).cuda(); dY = t.zeros(b * ld, dtype=t.float64, device="cuda")
        pkg.spmmv(A, dX, dY, b, ld, pkg.ROWWISE)
        t.cuda.synchronize()
        got = dY.cpu().numpy()
    finally:
        pkg.set_tuning(spmmv_stream=0, spmmv_stream_depth=1)
    both_nan = np.isnan(got) & np.isnan(Yo)
    assert np.array_equal(np.isnan(got), np.isnan(Yo))
    assert np.array_equal(got[~both_nan].view(np.uint64), Yo[~both_nan].view(np.uint64))
