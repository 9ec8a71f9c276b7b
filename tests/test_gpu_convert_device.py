"""uspmv_convert_to_scs_device_from_arrays: convert_to_scs (code/utilities.hpp:1842-2104) from DEVICE-resident COO arrays -- row populations,
chunk lengths, chunk pointers, permutations and the scatter on the device; the sigma-window ordering by the reference's std::sort on the
host over the row counts (USPMV_SORT_HOST: every array bit-identical to the host path) or by a stable device ranking
(USPMV_SORT_DEVICE_STABLE: chunk arrays and y in original row order bit-identical, tie order differs)."""
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


def dev_coo(t, m):
    I, J, V = m.arrays()
    return t.from_numpy(np.array(I)).cuda(), t.from_numpy(np.array(J)).cuda(), t.from_numpy(np.array(V)).cuda()


def matrices(pkg):
    rng = np.random.default_rng(3)
    n = 300                                               # rows 0, 7, 14, ... and the last 40 rows are empty
    I = np.sort(rng.integers(0, n - 40, 3000)); I = I[I % 7 != 0]
    ragged = pkg.Coo.from_arrays(n, n, I, rng.integers(0, n, I.size), rng.standard_normal(I.size))
    return [pkg.read_mtx(mtx_path(nm)) for nm in ("FDM-2d-16", "impcol_e", "bcsstk13", "matrix1", "matrix_band_klein")] + [ragged, pkg.gen_stencil27(9, 11, 13)]


def test_host_sort_mode_equals_the_host_conversion_bit_for_bit(pkg, t):
    for m in matrices(pkg):
        dI, dJ, dV = dev_coo(t, m)
        for C, sigma in ((1, 1), (4, 4), (16, 512), (32, 512), (32, 1), (64, 128), (128, 256), (5, 7), (32, 100000)):
            for code in (pkg.F64, pkg.F32):
                for permute in (True, False):
                    s = pkg.convert_to_scs(m, C, sigma, code)
                    if permute:
                        pkg.permute_scs_cols(s, s.arrays()["old_to_new_idx"])
                    a = s.arrays()
                    lay, A, o2n, n2o = pkg.convert_to_scs_device_from_arrays(dI, dJ, dV, m.n_rows, m.n_cols, C, sigma, code, permute_cols=permute)
                    assert (lay.n_chunks, lay.n_elements, lay.n_rows_padded, lay.nnz) == (s.n_chunks, s.n_elements, s.n_rows_padded, s.nnz)
                    la = lay.arrays()
                    for k in ("chunk_ptrs", "chunk_lengths", "old_to_new_idx", "new_to_old_idx"):
                        assert np.array_equal(la[k], a[k]), (C, sigma, k)
                    assert np.array_equal(o2n.cpu().numpy(), a["old_to_new_idx"]) and np.array_equal(n2o.cpu().numpy(), a["new_to_old_idx"])
                    d = pkg.dmat_download(A)
                    for k in ("chunk_ptrs", "chunk_lengths", "col_idxs", "values"):
                        assert np.array_equal(d[k], a[k]), (C, sigma, code, permute, k)
        # fixed permutation (the sp struct of an ap[dp_sp] pair takes the dp struct's), identity-permutation quirk included
        s0 = pkg.convert_to_scs(m, 8, 32, pkg.F64)
        fp = s0.arrays()["old_to_new_idx"].copy()
        s1 = pkg.convert_to_scs(m, 8, 32, pkg.F32, fixed_permutation=fp)
        lay, A, o2n, _ = pkg.convert_to_scs_device_from_arrays(dI, dJ, dV, m.n_rows, m.n_cols, 8, 32, pkg.F32, fixed_permutation=t.from_numpy(fp).cuda(), permute_cols=False)
        d = pkg.dmat_download(A)
        for k in ("chunk_ptrs", "chunk_lengths", "col_idxs", "values"):
            assert np.array_equal(d[k], s1.arrays()[k]), ("fixed", k)
        assert np.array_equal(lay.arrays()["old_to_new_idx"], s1.arrays()["old_to_new_idx"]) and np.array_equal(o2n.cpu().numpy(), np.arange(m.n_rows))


def test_device_stable_sort_same_chunks_same_y_in_original_order(pkg, orc, t):
    """The stable device ranking orders rows of equal length differently from the reference's unstable std::sort; everything that does not
    depend on the tie order is bit-identical: chunk_lengths, chunk_ptrs, n_elements -- and y back in ORIGINAL row order, because a row's
    entry order (= its FMA chain) never changes.  The struct is a valid SELL-C-sigma struct: lengths descend inside every window."""
    for m in matrices(pkg):
        dI, dJ, dV = dev_coo(t, m)
        for C, sigma in ((4, 4), (16, 512), (32, 512), (32, 1), (64, 128), (5, 7)):
            s = pkg.convert_to_scs(m, C, sigma, pkg.F64)
            pkg.permute_scs_cols(s, s.arrays()["old_to_new_idx"])
            a = s.arrays()
            lay, A, o2n, n2o = pkg.convert_to_scs_device_from_arrays(dI, dJ, dV, m.n_rows, m.n_cols, C, sigma, pkg.F64, sort=pkg.SORT_DEVICE_STABLE)
            la = lay.arrays()
            assert np.array_equal(la["chunk_lengths"], a["chunk_lengths"]) and np.array_equal(la["chunk_ptrs"], a["chunk_ptrs"]) and lay.n_elements == s.n_elements
            p = la["old_to_new_idx"]
            assert np.unique(p).size == m.n_rows and p.min() >= 0 and p.max() < m.n_rows        # injective; stable: padding rows sort behind the real empty ones
            counts = np.bincount(m.arrays()[0], minlength=m.n_rows)
            lens = np.zeros(s.n_rows_padded, np.int64); lens[p] = counts
            for b in range(0, s.n_rows_padded, sigma):
                w = lens[b:b + sigma]
                assert np.all(w[:-1] >= w[1:])
                same = np.flatnonzero(w[:-1] == w[1:])                                           # stable: ties in original order
                inv = np.full(s.n_rows_padded, 1 << 30, np.int64); inv[p] = np.arange(m.n_rows)
                real = inv[b + same + 1] < (1 << 30)                                             # (padding rows come last among the empty ones)
                assert np.all(inv[b + same][real] < inv[b + same + 1][real])
            x = make_x(m.n_rows)
            nx = max(s.n_rows_padded, m.n_cols)
            xd = t.zeros(nx, dtype=t.float64, device="cuda")
            xd[o2n.long()] = t.from_numpy(x).cuda()                                              # x_permuted[old_to_new[i]] = x[i]
            y = t.zeros(s.n_rows_padded, dtype=t.float64, device="cuda")
            pkg.spmv(A, xd, y)
            y_orig = y[o2n.long()].cpu().numpy()
            xh = np.zeros(nx); xh[a["old_to_new_idx"]] = x
            y_ref = orc.spmv_scs(C, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], xh)[a["old_to_new_idx"]]
            assert np.array_equal(y_orig, y_ref), (C, sigma)


def test_refusals(pkg, t):
    I = t.tensor([2, 0, 1], dtype=t.int32, device="cuda"); J = t.tensor([0, 1, 2], dtype=t.int32, device="cuda"); V = t.ones(3, dtype=t.float64, device="cuda")
    with pytest.raises(pkg.UspmvError, match="sorted by row"):
        pkg.convert_to_scs_device_from_arrays(I, J, V, 3, 3, 2, 2)
    I2 = t.tensor([0, 1, 7], dtype=t.int32, device="cuda")
    with pytest.raises(pkg.UspmvError, match="outside"):
        pkg.convert_to_scs_device_from_arrays(I2, J, V, 3, 3, 2, 2)
    I3 = t.tensor([0, 1, 2], dtype=t.int32, device="cuda")
    with pytest.raises(pkg.UspmvError, match="8192"):
        pkg.convert_to_scs_device_from_arrays(t.arange(20000, dtype=t.int32, device="cuda"), t.zeros(20000, dtype=t.int32, device="cuda"), t.ones(20000, dtype=t.float64, device="cuda"),
                                              20000, 20000, 32, 16384, sort=pkg.SORT_DEVICE_STABLE)
    bad = t.tensor([0, 1, 9], dtype=t.int32, device="cuda")
    with pytest.raises(pkg.UspmvError, match="fixed_permutation"):
        pkg.convert_to_scs_device_from_arrays(I3, J, V, 3, 3, 2, 2, fixed_permutation=bad)
    lay, A, o2n, n2o = pkg.convert_to_scs_device_from_arrays(I3, J, V, 3, 3, 2, 2, want_layout=False)      # no host struct at all
    assert lay is None and A.n_chunks == 2 and A.n_elements == 4
    d = pkg.dmat_download(A)
    assert np.array_equal(d["chunk_lengths"], [1, 1])


def test_full_size_from_device_arrays_equals_the_host_layout_path(pkg, t):
    """nlpkkt200-class size (253^3 stencil, 4.3e8 entries): the handle built from device arrays equals the one of uspmv_convert_to_scs_device
    (host layout + device scatter) -- compared on the device, array by array -- in both ordering modes for the chunk arrays, and y."""
    import time
    g = 253
    m = pkg.gen_stencil27(g, g, g)
    lay0, A0 = pkg.convert_to_scs_device(m, 32, 512, pkg.F64)
    dI, dJ, dV = dev_coo(t, m)
    n, nc = m.n_rows, m.n_cols
    del m
    t.cuda.synchronize(); t0 = time.time()
    lay, A, o2n, n2o = pkg.convert_to_scs_device_from_arrays(dI, dJ, dV, n, nc, 32, 512, pkg.F64)
    t.cuda.synchronize(); t_host_sort = time.time() - t0
    d0, d1 = pkg.dmat_download(A0), pkg.dmat_download(A)
    for k in ("chunk_ptrs", "chunk_lengths", "col_idxs", "values"):
        assert np.array_equal(d0[k], d1[k]), k
    assert np.array_equal(lay.arrays()["old_to_new_idx"], lay0.arrays()["old_to_new_idx"])
    del d0, d1, A
    t0 = time.time()
    lay2, A2, o2n2, n2o2 = pkg.convert_to_scs_device_from_arrays(dI, dJ, dV, n, nc, 32, 512, pkg.F64, sort=pkg.SORT_DEVICE_STABLE, want_layout=False)
    t.cuda.synchronize(); t_dev_sort = time.time() - t0
    x = t.from_numpy(make_x(n)).cuda()
    xp0 = t.zeros(lay0.n_rows_padded, dtype=t.float64, device="cuda"); xp0[:n] = x[n2o.long()]
    xp2 = t.zeros(lay0.n_rows_padded, dtype=t.float64, device="cuda"); xp2[:n] = x[n2o2.long()]
    y0 = t.zeros_like(xp0); y2 = t.zeros_like(xp0)
    pkg.spmv(A0, xp0, y0); pkg.spmv(A2, xp2, y2)
    assert t.equal(y0[o2n.long()], y2[o2n2.long()])
    print(f"\nconvert from device arrays, 253^3 stencil ({dI.numel()} entries): host-sort mode {t_host_sort:.3f} s, device-stable mode {t_dev_sort:.3f} s")
