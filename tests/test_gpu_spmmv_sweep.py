"""Block-vector column-window sweep (uspmv_dmat_optimize_block_sweep, csrc/spmmv_sweep.hip): SpMMV with 64-byte X rows over windows of X rows,
bit-identical to the reference's block_spmv_omp_scs_general (code/kernels.hpp:306-398) in both block-vector layouts -- column-major X and Y
taken as they are."""
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


CASES = [("stencil3", 32, 512, 9, 1024), ("stencil3", 32, 512, 8, 2048), ("stencil3", 32, 1, 10, 4096), ("stencil3", 64, 128, 9, 1024),
         ("stencil1", 32, 512, 9, 2048), ("bcsstk13", 32, 512, 9, 1024), ("band", 32, 64, 7, 1024), ("stencil3", 16, 512, 9, 1024)]


@pytest.mark.parametrize("name,C,sigma,wlog,tile_rows", CASES)
@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_block_sweep_bitexact_both_layouts(pkg, orc, t, name, C, sigma, wlog, tile_rows, dt):
    if name == "stencil3":
        m = pkg.gen_stencil27(14, 13, 12, dof=3)
    elif name == "stencil1":
        m = pkg.gen_stencil27(24, 20, 18)
    elif name == "band":
        m = pkg.gen_banded_random(5000, 40, 700, magnitude_decades=4.0)
    else:
        m = pkg.read_mtx(mtx_path(name))
    dtype = pkg.F64 if dt == "f64" else pkg.F32
    b = 8 if dt == "f64" else 16
    s, a, xp = prep(pkg, m, C, sigma, dtype)
    A = pkg.DeviceMatrix(s)
    nt, ns = A.optimize_block_sweep(s, b, wlog=wlog, tile_rows=tile_rows)
    assert nt > 0
    if ns != nt:
        pytest.skip(f"plan not installed for this shape ({ns} of {nt} tiles sweep)")
    pkg.set_tuning(spmmv_variant=9)                      # nothing but the sweep may answer
    try:
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
    finally:
        pkg.set_tuning(spmmv_variant=0)


def test_block_sweep_special_values_and_padding(pkg, orc, t):
    """Signed zeros, infinities and NaN travel like in the reference (a lane that sits a round out must not touch its accumulators; the
    stripped trailing padding is applied once per column), incl. a matrix whose padding column holds a non-finite X row."""
    m = pkg.gen_stencil27(10, 9, 8, dof=3)
    s, a, xp = prep(pkg, m, 32, 512, pkg.F64)
    b, ld = 8, s.n_rows_padded
    A = pkg.DeviceMatrix(s)
    nt, ns = A.optimize_block_sweep(s, b, wlog=9, tile_rows=1024)
    assert nt == ns
    rng = np.random.default_rng(5)
    X = block(xp, b, ld, True)
    sp = rng.choice(X.size, 200, replace=False)
    X[sp[:50]] = np.inf; X[sp[50:100]] = -np.inf; X[sp[100:150]] = np.nan; X[sp[150:]] = -0.0
    pad_col = int(a["col_idxs"][a["chunk_ptrs"][0] + (a["chunk_lengths"][0] - 1) * 32 + 31])
    X[pad_col * b:(pad_col + 1) * b] = [np.inf, -np.inf, np.nan, -0.0, 0.0, 1.0, -1.0, 5e-324]
    Yo = orc.spmmv_scs(32, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], X, b, ld, True)
    dX = t.from_numpy(X).cuda(); dY = t.zeros(b * ld, dtype=t.float64, device="cuda")
    pkg.set_tuning(spmmv_variant=9)
    try:
        pkg.spmmv(A, dX, dY, b, ld, pkg.ROWWISE)
        t.cuda.synchronize()
    finally:
        pkg.set_tuning(spmmv_variant=0)
    got = dY.cpu().numpy()
    both_nan = np.isnan(got) & np.isnan(Yo)
    assert np.array_equal(got.view(np.uint64)[~both_nan], Yo.view(np.uint64)[~both_nan]) and np.array_equal(np.isnan(got), np.isnan(Yo))
