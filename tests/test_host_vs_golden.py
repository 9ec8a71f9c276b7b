"""Product host layer (libuspmv.so: MatrixMarket reader, convert_to_scs, permutations, precision
split, row partitioning, halo discovery) against golden vectors produced by the genuine reference.
Integer / index outputs must be bit-identical, including the std::sort tie order."""
import ctypes
import hashlib
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, golden, mtx_path
from test_oracle_vs_golden import _unit_fixture_cases, check_unit_fixture


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


FULL = ["FDM-2d-16", "impcol_e", "matrix1", "myBigMat", "mySymmMat", "matrix_band_klein", "bcsstk13"]


def test_library_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(ROOT, "include", "uspmv.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(uspmv_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 45
    L = ctypes.CDLL(pkg.library_path())
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, missing
    from ultimate_spmv_amd import binding
    assert declared == set(binding._SIGS), declared ^ set(binding._SIGS)


@pytest.mark.parametrize("name", FULL)
def test_read_mtx(pkg, name):
    g = golden(f"scs_{name}.npz")
    m = pkg.read_mtx(mtx_path(name))
    I, J, V = m.arrays()
    assert (m.n_rows, m.n_cols, m.nnz) == (int(g["n_rows"]), int(g["n_cols"]), int(g["nnz"]))
    assert np.array_equal(I, g["I"]) and np.array_equal(J, g["J"]) and np.array_equal(V, g["vals"])


def test_read_mtx_pattern_integer_and_errors(pkg, tmp_path):
    p = tmp_path / "pat.mtx"
    p.write_text("%%MatrixMarket matrix coordinate pattern symmetric\n% c\n3 3 3\n1 1\n3 1\n2 2\n")
    I, J, V = pkg.read_mtx(str(p)).arrays()
    assert I.tolist() == [0, 0, 1, 2] and J.tolist() == [0, 2, 1, 0] and V.tolist() == [0.01] * 4
    p.write_text("%%MatrixMarket matrix coordinate integer general\n2 2 2\n2 1 7\n1 2 -3\n")
    I, J, V = pkg.read_mtx(str(p)).arrays()
    assert I.tolist() == [0, 1] and J.tolist() == [1, 0] and V.tolist() == [-3.0, 7.0]
    for text, status in (("%%MatrixMarket matrix coordinate real general\n2 3 1\n1 1 1.0\n", 3),   # not square
                         ("%%MatrixMarket matrix coordinate complex general\n2 2 1\n1 1 1 0\n", 3),
                         ("%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n", 3),
                         ("%%MatrixMarket matrix coordinate real general\n2 2 2\n1 1 1.0\n", 2),      # premature EOF
                         ("hello\n", 2)):
        p.write_text(text)
        with pytest.raises(pkg.UspmvError) as e:
            pkg.read_mtx(str(p))
        assert e.value.status == status
    with pytest.raises(pkg.UspmvError) as e:
        pkg.read_mtx(str(tmp_path / "missing.mtx"))
    assert e.value.status == 2


@pytest.mark.parametrize("name", FULL)
@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_convert_to_scs_full_arrays(pkg, name, dt):
    g = golden(f"scs_{name}.npz")
    m = pkg.read_mtx(mtx_path(name))
    s = pkg.convert_to_scs(m, int(g["C"]), int(g["sigma"]), pkg.F64 if dt == "f64" else pkg.F32)
    a = s.arrays()
    assert (s.n_chunks, s.n_elements, s.nnz) == (int(g["n_chunks"]), int(g["n_elements"]), int(g["nnz"]))
    assert np.array_equal(a["chunk_ptrs"], g[f"{dt}_chunk_ptrs"])
    assert np.array_equal(a["chunk_lengths"], g[f"{dt}_chunk_lengths"])
    assert np.array_equal(a["old_to_new_idx"], g[f"{dt}_old_to_new"])      # std::sort tie order reproduced
    assert np.array_equal(a["new_to_old_idx"], g[f"{dt}_new_to_old"])
    assert np.array_equal(a["col_idxs"], g[f"{dt}_col_idxs_pre"])
    assert np.array_equal(a["values"], g[f"{dt}_values"])
    pkg.permute_scs_cols(s, a["old_to_new_idx"])
    assert np.array_equal(s.arrays()["col_idxs"], g[f"{dt}_col_idxs"])
    xp = pkg.apply_permutation(g["x"].astype(a["values"].dtype), a["new_to_old_idx"])
    assert np.array_equal(xp, g[f"{dt}_x_perm"][:s.n_rows])


def test_convert_to_scs_grid_hashes(pkg):
    """The (matrix, C, sigma) grid of scripts/validate_master.sh:16-23 -- every array, both dtypes."""
    grid = json.load(open(os.path.join(GOLDEN, "scs_grid_sha1.json")))
    mats = {}
    for key, ent in grid.items():
        name, Cc, sg = key.split("|")
        m = mats.setdefault(name, pkg.read_mtx(mtx_path(name)))
        for dt, code in (("f64", pkg.F64), ("f32", pkg.F32)):
            s = pkg.convert_to_scs(m, int(Cc), int(sg), code)
            a = s.arrays()
            pkg.permute_scs_cols(s, a["old_to_new_idx"])
            a = s.arrays()
            e = ent[dt]
            assert (s.n_elements, s.n_chunks) == (e["n_elements"], e["n_chunks"]), key
            for f, arr in (("chunk_lengths", a["chunk_lengths"]), ("chunk_ptrs", a["chunk_ptrs"]),
                           ("col_idxs", a["col_idxs"]), ("values", a["values"]), ("old_to_new", a["old_to_new_idx"])):
                assert sha(arr) == e[f], (key, dt, f)


def test_coo_binary_cache_roundtrip(pkg, tmp_path):
    """uspmv_coo_save / uspmv_coo_load: verbatim round trip, corrupt files refused."""
    m = pkg.read_mtx(mtx_path("bcsstk13"))
    f = str(tmp_path / "m.uspmvcoo")
    m.save(f)
    r = pkg.Coo.load(f)
    assert (r.n_rows, r.n_cols, r.nnz) == (m.n_rows, m.n_cols, m.nnz)
    for x, y in zip(m.arrays(), r.arrays()):
        assert np.array_equal(x, y)
    raw = open(f, "rb").read()
    open(f, "wb").write(raw[:len(raw) // 2])
    with pytest.raises(pkg.UspmvError):
        pkg.Coo.load(f)
    open(f, "wb").write(b"not a cache")
    with pytest.raises(pkg.UspmvError):
        pkg.Coo.load(f)
    with pytest.raises(pkg.UspmvError):
        pkg.Coo.load(str(tmp_path / "missing"))


def test_convert_errors(pkg):
    m = pkg.Coo.from_arrays(3, 3, [0, 1, 2], [0, 1, 2], [1.0, 2.0, 3.0])
    for C_, s_ in ((0, 1), (1, 0), (-4, 2)):
        with pytest.raises(pkg.UspmvError):
            pkg.convert_to_scs(m, C_, s_)
    with pytest.raises(pkg.UspmvError):
        pkg.Coo.from_arrays(3, 3, [0, 5], [0, 1], [1.0, 2.0])     # row outside the matrix
    with pytest.raises(pkg.UspmvError):
        pkg.convert_to_scs(m, 2, 2, pkg.F64, fixed_permutation=[0, 1, 9])


def test_unsorted_coo_matches_sorted(pkg):
    """convert_to_scs only relies on the order of entries INSIDE a row (code/utilities.hpp:2013-2036)."""
    g = golden("scs_impcol_e.npz")
    rng = np.random.default_rng(3)
    # shuffle whole rows as blocks: keeps the in-row order, breaks the global row order
    order = np.concatenate([np.flatnonzero(g["I"] == r) for r in rng.permutation(int(g["n_rows"]))])
    m = pkg.Coo.from_arrays(int(g["n_rows"]), int(g["n_cols"]), g["I"][order], g["J"][order], g["vals"][order])
    a = pkg.convert_to_scs(m, 32, 512).arrays()
    assert np.array_equal(a["col_idxs"], g["f64_col_idxs_pre"]) and np.array_equal(a["values"], g["f64_values"])


def test_reference_unit_fixtures(pkg):
    n = 0
    for name, coo, exp, cs in _unit_fixture_cases():
        def conv(nr, nc, I, J, v, Cc, sg, dt):
            m = pkg.Coo.from_arrays(nr, nc, I, J, v)
            s = pkg.convert_to_scs(m, Cc, sg, pkg.F32 if dt == np.float32 else pkg.F64)
            class R: pass
            r = R(); r.__dict__.update({k: v2.copy() for k, v2 in s.arrays().items()})
            return r
        check_unit_fixture(conv, name, coo, exp, cs)
        n += 1
    assert n >= 13
    fx = json.load(open(os.path.join(GOLDEN, "reference_unit_fixtures.json")))
    for base in ("M1", "M_big"):   # precision split at 1.0 (code/test_suite/tests.cpp:8-24)
        c = fx[base]
        dp, sp = pkg.partition_precisions(pkg.Coo.from_arrays(c["n_rows"], c["n_cols"], c["I"], c["J"], c["values"]), 1.0)
        hp, lp = fx[f"exp_{base}_hp"], fx[f"exp_{base}_lp"]
        I, J, V = dp.arrays()
        assert I.tolist() == hp["I"] and J.tolist() == hp["J"] and V.tolist() == hp["values"]
        I, J, V = sp.arrays()
        assert I.tolist() == lp["I"] and J.tolist() == lp["J"]
        assert V.astype(np.float32).tolist() == np.array(lp["values"], np.float32).tolist()


@pytest.mark.parametrize("name", ["bcsstk13", "impcol_e", "FDM-2d-16", "matrix1"])
def test_partition_and_fixed_permutation(pkg, name):
    a = golden("ap.npz")
    p = name + "_"
    m = pkg.read_mtx(mtx_path(name))
    dp, sp = pkg.partition_precisions(m, float(a[p + "th"]))
    I, J, V = dp.arrays()
    assert np.array_equal(I, a[p + "dp_I"]) and np.array_equal(J, a[p + "dp_J"]) and np.array_equal(V, a[p + "dp_V"])
    I, J, V = sp.arrays()
    assert np.array_equal(I, a[p + "sp_I"]) and np.array_equal(J, a[p + "sp_J"])
    assert np.array_equal(V.astype(np.float32), a[p + "sp_V"]) and np.array_equal(V, a[p + "sp_V"].astype(np.float64))
    Cc, sg = int(a[p + "C"]), int(a[p + "sigma"])
    ds = pkg.convert_to_scs(dp, Cc, sg, pkg.F64)
    da = ds.arrays()
    for f, k in (("chunk_ptrs", "dp_chunk_ptrs"), ("chunk_lengths", "dp_chunk_lengths"), ("col_idxs", "dp_col_idxs_pre"),
                 ("values", "dp_values"), ("old_to_new_idx", "old_to_new"), ("new_to_old_idx", "new_to_old")):
        assert np.array_equal(da[f], a[p + k]), f
    ss = pkg.convert_to_scs(sp, Cc, sg, pkg.F32, fixed_permutation=da["old_to_new_idx"])
    sa = ss.arrays()
    for f, k in (("chunk_ptrs", "sp_chunk_ptrs"), ("chunk_lengths", "sp_chunk_lengths"), ("col_idxs", "sp_col_idxs_pre"),
                 ("values", "sp_values"), ("old_to_new_idx", "sp_old_to_new")):
        assert np.array_equal(sa[f], a[p + k]), f
    pkg.permute_scs_cols(ds, da["old_to_new_idx"]); pkg.permute_scs_cols(ss, da["old_to_new_idx"])
    assert np.array_equal(ds.arrays()["col_idxs"], a[p + "dp_col_idxs"])
    assert np.array_equal(ss.arrays()["col_idxs"], a[p + "sp_col_idxs"])


def test_halo_setup(pkg):
    from ultimate_spmv_amd import binding as B
    h = golden("halo.npz")
    meta = json.load(open(os.path.join(GOLDEN, "halo_meta.json")))
    for key, m in meta.items():
        name, Cs, ss, method, Ps = key.rsplit("_", 4)
        Cc, sg, P = int(Cs[1:]), int(ss[1:]), int(Ps[1:])
        tot = pkg.read_mtx(mtx_path(name))
        wsa = pkg.seg_work_sharing_arr(tot, method, P)
        assert np.array_equal(wsa, h[key + "_wsa"]), key
        for r in range(P):
            loc = B.seg_local_coo(tot, wsa, r)
            assert (loc.n_rows, loc.nnz) == (m["n_local"][r], m["nnz"][r])
            s = pkg.convert_to_scs(loc, Cc, sg)
            assert s.n_elements == m["n_elements"][r]
            if key + "_y_global" in h.files:   # the host half of uspmv_dist_check: entry-ordered chains == the reference's y for the ramp x
                assert np.array_equal(pkg.dist_check_reference(loc, wsa, r, P), h[key + "_y_global"][wsa[r]:wsa[r + 1]]), (key, r)
            plan = pkg.HaloPlan(s, wsa, r, P)
            assert plan.n_halo == m["n_halo"][r] and plan.recv_counts.tolist() == m["recv_counts"][r]
            assert np.array_equal(plan.recv_counts_cumsum, h[f"{key}_r{r}_recv_cumsum"])
            assert np.array_equal(plan.recv_idxs, h[f"{key}_r{r}_recv_idxs"])
            a = s.arrays()
            assert np.array_equal(a["old_to_new_idx"], h[f"{key}_r{r}_old_to_new"])
            pkg.permute_scs_cols(s, a["old_to_new_idx"])
            assert np.array_equal(s.arrays()["col_idxs"], h[f"{key}_r{r}_col_idxs"])
            inner, bnd = s.split_chunks(plan.n_local)
            assert len(inner) + len(bnd) == s.n_chunks and np.array_equal(np.sort(np.concatenate([inner, bnd])), np.arange(s.n_chunks))
            ci, cp, cl = s.arrays()["col_idxs"], a["chunk_ptrs"], a["chunk_lengths"]
            for c in inner:
                assert ci[cp[c]:cp[c] + cl[c] * Cc].max(initial=0) < plan.n_local
            for c in bnd:
                assert ci[cp[c]:cp[c] + cl[c] * Cc].max() >= plan.n_local
            # ... and in three classes: halo columns only through the reference's (+0.0, one column) padding form a class of their own
            cls, pad_col = s.chunk_classes(plan.n_local)
            va = s.arrays()["values"]
            zero_halo = (ci >= plan.n_local) & (va.view(np.uint64 if va.dtype == np.float64 else np.uint32) == 0)
            want_col = int(ci[zero_halo].min()) if zero_halo.any() else None
            want = np.zeros(s.n_chunks, np.uint8)
            for c in range(s.n_chunks):
                sl = slice(cp[c], cp[c] + cl[c] * Cc)
                halo = ci[sl] >= plan.n_local
                if halo.any():
                    want[c] = 1 if (zero_halo[sl][halo] & (ci[sl][halo] == want_col)).all() else 2
            assert np.array_equal(cls, want), (key, r)
            assert pad_col == (want_col if (want == 1).any() else -1), (key, r, pad_col, want_col)
            assert np.array_equal(np.flatnonzero(cls == 0), inner) and np.array_equal(np.flatnonzero(cls > 0), bnd)
            if r > 0 and s.n_elements > s.nnz:                   # padded chunks exist: their column 0 is a halo column on ranks > 0
                assert (cls == 1).any() or (cls == 2).all() or pad_col == -1


def test_seg_errors(pkg):
    m = pkg.Coo.from_arrays(2, 2, [0, 1], [0, 1], [1.0, 1.0])
    with pytest.raises(pkg.UspmvError):
        pkg.seg_work_sharing_arr(m, "seg-rows", 3)      # n_rows < ranks (code/mpi_funcs.hpp:442-444)
    with pytest.raises(pkg.UspmvError):
        pkg.seg_work_sharing_arr(m, 7, 2)


def test_generator_is_symmetric_and_deterministic(pkg):
    m = pkg.gen_stencil27(5, 4, 3, dof=2, seed=7)
    I, J, V = m.arrays()
    n = 5 * 4 * 3 * 2
    import scipy.sparse as sp
    A = sp.coo_matrix((V, (I, J)), shape=(n, n)).tocsr()
    assert (A != A.T).nnz == 0 and np.all(np.diff(I) >= 0)
    assert A[0].nnz == 8 * 2 and A.diagonal().min() > 53
    part = pkg.gen_stencil27(5, 4, 3, dof=2, seed=7, row_begin=10, row_end=50)
    pI, pJ, pV = part.arrays()
    sel = (I >= 10) & (I < 50)
    assert np.array_equal(pI + 10, I[sel]) and np.array_equal(pJ, J[sel]) and np.array_equal(pV, V[sel])
    hv = pkg.gen_stencil27(4, 4, 4, magnitude_decades=10.0)
    _, _, W = hv.arrays()
    assert np.abs(W).min() < 1e-5 and np.abs(W).max() > 10


def test_device_entry_points_fail_loudly_without_gpu(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = pkg.gen_stencil27(4, 4, 4)
    s = pkg.convert_to_scs(m, 32, 512)
    h = ctypes.c_void_p()
    rc = pkg.lib().uspmv_dmat_upload(s.h, ctypes.byref(h))
    assert rc == 5 and b"no CPU fallback" in pkg.lib().uspmv_last_error()   # USPMV_ERR_NO_DEVICE
    assert pkg.device_count() == 0
