"""The tile-local-column plan over single x ELEMENTS (uspmv_dmat_optimize's fallback when a tile's columns are scattered over too many 16-element lines:
x in a numbering that is only loosely related to the rows'): one 8-byte gather per distinct column of a tile, the same slot-ordered chains as
scs_impl_cpu (code/kernels.hpp:218-258) -- bit-identical to the oracle; regular numberings keep the line plan."""
import numpy as np
import pytest

from conftest import make_x

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def t(pkg):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.cuda.set_device(0)
    return torch


def scrambled_columns(pkg, g, dof, K, seed=3, rows_too=False):
    """27-point x dof stencil on g^3 nodes with the COLUMNS' nodes renumbered at random inside consecutive blocks of K nodes (rows keep their order
    unless rows_too: then the permutation is symmetric)."""
    base = pkg.gen_stencil27(g, g, g, dof=dof)
    I, J, V = (np.array(a) for a in base.arrays())
    n = base.n_rows
    nn = n // dof
    rng = np.random.default_rng(seed)
    p = np.arange(nn, dtype=np.int64)
    for s0 in range(0, nn, K):
        seg = p[s0:s0 + K].copy(); rng.shuffle(seg); p[s0:s0 + K] = seg
    J2 = (p[J // dof] * dof + J % dof).astype(np.int32)
    I2 = (p[I // dof] * dof + I % dof).astype(np.int32) if rows_too else I
    o = np.lexsort((J2, I2))
    return pkg.Coo.from_arrays(n, n, I2[o], J2[o], V[o])


@pytest.fixture(scope="module")
def scattered(pkg):
    return scrambled_columns(pkg, 40, 3, 8000)       # a 256-row tile's columns lie on ~2 000 lines: the line plan cannot stage it


@pytest.mark.parametrize("C,sigma,dt,cap", [(32, 512, "f64", 4096), (32, 1, "f32", 4096), (64, 128, "f64", 4096), (16, 64, "f64", 4096), (32, 512, "f64", 1100)])
def test_element_plan_bitexact(pkg, orc, t, scattered, C, sigma, dt, cap):
    m = scattered
    dtype = pkg.F64 if dt == "f64" else pkg.F32
    s = pkg.convert_to_scs(m, C, sigma, dtype)
    a = s.arrays()
    x = np.zeros(max(s.n_rows_padded, s.n_cols), a["values"].dtype)
    x[:s.n_cols] = make_x(s.n_cols).astype(a["values"].dtype)
    want = orc.spmv_scs(C, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], x)
    pkg.set_tuning(tlc_elem_cap=cap)
    try:
        A = pkg.DeviceMatrix(s, tlc=True)
    finally:
        pkg.set_tuning(tlc_elem_cap=4096)
    if cap >= 4096:
        assert A.plan_granularity() == 1 and A.tlc_staged == A.tlc_tiles, (A.plan_granularity(), A.tlc_staged, A.tlc_tiles)
    else:
        assert A.plan_granularity() != 1                  # the tiles list more elements than the cap allows: the other plans answer
    dx = t.from_numpy(x).cuda(); dy = t.full((s.n_rows_padded,), -3.0, dtype=dx.dtype, device="cuda")
    pkg.spmv(A, dx, dy)
    got = dy.cpu().numpy()
    assert np.array_equal(got[:s.n_rows_padded], want[:s.n_rows_padded])
    # ... and over a list of tiles (the distributed step's interior / boundary launches)
    if A.plan_granularity() == 1 and A.tlc_tiles >= 4 and C >= 32:          # (narrow chunks run on an internal re-chunking: no tile lists on the caller's handle)
        ids = t.tensor([1, 3, 0], dtype=t.int32, device="cuda")
        dy2 = t.full_like(dy, -3.0)
        pkg.spmv_tiles(A, ids, dx, dy2)
        got2 = dy2.cpu().numpy()
        R = A.tile_rows
        for tile in (0, 1, 3):
            assert np.array_equal(got2[tile * R:(tile + 1) * R], want[tile * R:(tile + 1) * R])
        assert np.all(got2[2 * R:3 * R] == -3.0)


def test_regular_numbering_keeps_the_line_plan_unless_forced(pkg, orc, t):
    s = pkg.convert_to_scs(pkg.gen_stencil27(16, 16, 16, dof=3), 32, 512, pkg.F64)
    a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"]); a = s.arrays()
    A = pkg.DeviceMatrix(s, tlc=True)
    assert A.plan_granularity() == 16 and A.tlc_staged == A.tlc_tiles
    pkg.set_tuning(tlc_elem=2)                            # (measurement aid: the element plan wherever it stages every tile)
    try:
        Ae = pkg.DeviceMatrix(s, tlc=True)
    finally:
        pkg.set_tuning(tlc_elem=1)
    assert Ae.plan_granularity() == 1
    x = np.zeros(s.n_rows_padded); x[:s.n_rows] = make_x(s.n_rows)
    want = orc.spmv_scs(32, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], x)
    for H in (A, Ae):
        dy = t.zeros(s.n_rows_padded, dtype=t.float64, device="cuda")
        pkg.spmv(H, t.from_numpy(x).cuda(), dy)
        assert np.array_equal(dy.cpu().numpy(), want)


@pytest.mark.parametrize("C,sigma,dt", [(32, 512, "f64"), (32, 64, "f32"), (64, 128, "f64")])
def test_rows_dealt_by_the_matrix_graph_then_the_element_plan(pkg, orc, t, C, sigma, dt):
    """Rows AND columns renumbered alike: a tile of consecutive rows is no compact piece of the mesh any more.  uspmv_dmat_optimize deals the rows to the tiles
    by the matrix graph (a private copy of the values, y through a row map) and lists single x elements: bit-identical to the oracle, every row of y."""
    m = scrambled_columns(pkg, 40, 3, 8000, rows_too=True)
    dtype = pkg.F64 if dt == "f64" else pkg.F32
    s = pkg.convert_to_scs(m, C, sigma, dtype)
    a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"]); a = s.arrays()
    x = np.zeros(s.n_rows_padded, a["values"].dtype)
    x[:s.n_rows] = pkg.apply_permutation(make_x(s.n_rows).astype(a["values"].dtype), a["new_to_old_idx"])
    want = orc.spmv_scs(C, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], x)
    A = pkg.DeviceMatrix(s, tlc=True)
    assert A.plan_granularity() == 1 and A.plan_rows_dealt() and A.tlc_staged * 20 >= A.tlc_tiles * 19, (A.plan_granularity(), A.tlc_staged, A.tlc_tiles)
    dx = t.from_numpy(x).cuda()
    for _ in range(2):
        dy = t.full((s.n_rows_padded,), -3.0, dtype=dx.dtype, device="cuda")
        pkg.spmv(A, dx, dy)
        assert np.array_equal(dy.cpu().numpy()[:s.n_rows], want[:s.n_rows])
    pkg.set_tuning(tlc_elem_rows=0)                       # without the row dealing the other plans answer -- same bits
    try:
        A0 = pkg.DeviceMatrix(s, tlc=True)
    finally:
        pkg.set_tuning(tlc_elem_rows=1)
    assert not A0.plan_rows_dealt()
    dy = t.full((s.n_rows_padded,), -3.0, dtype=dx.dtype, device="cuda")
    pkg.spmv(A0, dx, dy)
    assert np.array_equal(dy.cpu().numpy()[:s.n_rows], want[:s.n_rows])
