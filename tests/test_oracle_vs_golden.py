"""The oracle (oracle/uspmv_oracle.c) pinned against golden vectors produced by the GENUINE
reference (oracle/make_golden.py) and against the reference's own hand-written unit-test data."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import ADV_CS, GOLDEN, block_x, golden, make_x


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


FULL = ["FDM-2d-16", "impcol_e", "matrix1", "myBigMat", "mySymmMat", "matrix_band_klein", "bcsstk13"]


@pytest.mark.parametrize("name", FULL)
@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_spmv_kernel_bitexact(orc, name, dt):
    g = golden(f"scs_{name}.npz")
    y = orc.spmv_scs(int(g["C"]), int(g["n_chunks"]), g[f"{dt}_chunk_ptrs"], g[f"{dt}_chunk_lengths"],
                     g[f"{dt}_col_idxs"], g[f"{dt}_values"], g[f"{dt}_x_perm"])
    assert np.array_equal(y, g[f"{dt}_y_perm"])
    assert np.array_equal(orc.apply_permutation(y, g[f"{dt}_old_to_new"]), g[f"{dt}_y_orig"])


@pytest.mark.parametrize("name", FULL)
def test_convert_tie_independent_fields(orc, name):
    """chunk_lengths / chunk_ptrs / n_elements and y in ORIGINAL row order do not depend on the
    tie order of the sigma sort; they must match the reference exactly."""
    g = golden(f"scs_{name}.npz")
    s = orc.convert_to_scs(int(g["n_rows"]), int(g["n_cols"]), g["I"], g["J"], g["vals"], int(g["C"]),
                           int(g["sigma"]))
    assert s.n_elements == int(g["n_elements"])
    assert np.array_equal(s.chunk_lengths, g["f64_chunk_lengths"])
    assert np.array_equal(s.chunk_ptrs, g["f64_chunk_ptrs"])
    orc.permute_scs_cols(s, s.old_to_new_idx)
    xp = np.zeros(s.n_rows_padded)
    xp[:s.n_rows] = orc.apply_permutation(g["x"], s.new_to_old_idx)
    y = orc.spmv_scs(s.C, s.n_chunks, s.chunk_ptrs, s.chunk_lengths, s.col_idxs, s.values, xp)
    assert np.array_equal(orc.apply_permutation(y, s.old_to_new_idx), g["f64_y_orig"])


def test_permute_scs_cols(orc):
    g = golden("scs_bcsstk13.npz")
    class S: pass
    s = S(); s.n_elements = int(g["n_elements"]); s.n_rows = int(g["n_rows"])
    s.col_idxs = g["f64_col_idxs_pre"].copy()
    orc.permute_scs_cols(s, g["f64_old_to_new"])
    assert np.array_equal(s.col_idxs, g["f64_col_idxs"])


def test_grid_y_hashes(orc):
    """(matrix, C, sigma) grid of scripts/validate_master.sh:16-23: y (original order) of the oracle
    run on an oracle-converted matrix equals the reference's y bit for bit."""
    grid = json.load(open(os.path.join(GOLDEN, "scs_grid_sha1.json")))
    for name in ("FDM-2d-16", "impcol_e", "matrix1"):
        g = golden(f"scs_{name}.npz")
        for key, ent in grid.items():
            n, Cc, sg = key.split("|")
            if n != name:
                continue
            Cc, sg = int(Cc), int(sg)
            for dt, npdt in (("f64", np.float64), ("f32", np.float32)):
                s = orc.convert_to_scs(int(g["n_rows"]), int(g["n_cols"]), g["I"], g["J"], g["vals"], Cc, sg,
                                       dtype=npdt)
                assert s.n_elements == ent[dt]["n_elements"]
                assert sha(s.chunk_lengths) == ent[dt]["chunk_lengths"], key
                orc.permute_scs_cols(s, s.old_to_new_idx)
                xp = np.zeros(s.n_rows_padded, npdt)
                xp[:s.n_rows] = orc.apply_permutation(g["x"].astype(npdt), s.new_to_old_idx)
                y = orc.spmv_scs(s.C, s.n_chunks, s.chunk_ptrs, s.chunk_lengths, s.col_idxs, s.values, xp)
                assert sha(orc.apply_permutation(y, s.old_to_new_idx)) == ent[dt]["y_orig"], (key, dt)


@pytest.mark.parametrize("name", ["FDM-2d-16", "impcol_e", "matrix1", "bcsstk13"])
def test_csr(orc, name):
    """CRS kernel: the reference loop is omp-simd reassociated, so tolerance (1e-13 / 1e-5 of
    sum|a_ij x_j|, the reference's own max_rel_error, code/utilities.hpp:35-47)."""
    c = golden("csr.npz")
    g = golden(f"scs_{name}.npz")
    for dt, npdt, tol in (("f64", np.float64, 1e-13), ("f32", np.float32, 1e-5)):
        s = orc.convert_to_scs(int(g["n_rows"]), int(g["n_cols"]), g["I"], g["J"], g["vals"], 1, 1, dtype=npdt)
        assert np.array_equal(s.chunk_ptrs, c[f"{name}_{dt}_row_ptrs"])
        x = g["x"].astype(npdt)
        y = orc.spmv_csr(s.n_rows, s.chunk_ptrs, s.col_idxs, s.values, x)
        bound = orc.spmv_csr(s.n_rows, s.chunk_ptrs, s.col_idxs, np.abs(s.values), np.abs(x)).astype(np.float64)
        assert np.all(np.abs(y.astype(np.float64) - c[f"{name}_{dt}_y"]) <= tol * bound + 1e-300)
        for rowwise in (0, 1):
            X = block_x(x, s.n_rows, 4, s.n_rows, rowwise)
            Y = orc.spmmv_csr(s.n_rows, s.chunk_ptrs, s.col_idxs, s.values, X, 4, s.n_rows, rowwise)
            assert np.array_equal(Y, c[f"{name}_{dt}_Yb4_{'row' if rowwise else 'col'}"])


@pytest.mark.parametrize("name", ["FDM-2d-16", "impcol_e", "bcsstk13"])
def test_spmmv_bitexact(orc, name):
    g = golden(f"scs_{name}.npz")
    sp = golden("spmmv.npz")
    Cc, nch = int(g["C"]), int(g["n_chunks"])
    ld = nch * Cc
    for dt in ("f64", "f32"):
        for b in (2, 8):
            for rowwise in (0, 1):
                X = block_x(g[f"{dt}_x_perm"], ld, b, ld, rowwise)
                Y = orc.spmmv_scs(Cc, nch, g[f"{dt}_chunk_ptrs"], g[f"{dt}_chunk_lengths"], g[f"{dt}_col_idxs"],
                                  g[f"{dt}_values"], X, b, ld, rowwise)
                assert np.array_equal(Y, sp[f"{name}_{dt}_b{b}_{'row' if rowwise else 'col'}_Y"])


@pytest.mark.parametrize("name", ["bcsstk13", "impcol_e", "FDM-2d-16", "matrix1"])
def test_ap_bitexact(orc, name):
    a = golden("ap.npz")
    p = name + "_"
    Cc = int(a[p + "C"])
    nch = len(a[p + "dp_chunk_lengths"])
    # precision split
    g = golden(f"scs_{name}.npz")
    m = orc.partition_precisions_dpsp(g["vals"], float(a[p + "th"]))
    assert np.array_equal(g["vals"][m], a[p + "dp_V"]) and np.array_equal(g["I"][m], a[p + "dp_I"])
    assert np.array_equal(g["vals"][~m].astype(np.float32), a[p + "sp_V"]) and np.array_equal(g["J"][~m], a[p + "sp_J"])
    dp = (a[p + "dp_chunk_ptrs"], a[p + "dp_chunk_lengths"], a[p + "dp_col_idxs"], a[p + "dp_values"])
    sp = (a[p + "sp_chunk_ptrs"], a[p + "sp_chunk_lengths"], a[p + "sp_col_idxs"], a[p + "sp_values"])
    xp = a[p + "x_perm"]
    if Cc in ADV_CS:
        assert np.array_equal(orc.spmv_scs_ap_adv(Cc, nch, dp, sp, xp), a[p + "y_perm_adv"])
    assert np.array_equal(orc.spmv_scs_ap(Cc, nch, dp, sp, xp, xp.astype(np.float32)), a[p + "y_perm_gen"])
    # fixed-permutation conversion of the sp struct: tie-free, must match exactly
    s = orc.convert_to_scs(int(g["n_rows"]), int(g["n_cols"]), a[p + "sp_I"], a[p + "sp_J"],
                           a[p + "sp_V"].astype(np.float64), Cc, int(a[p + "sigma"]),
                           fixed_perm=a[p + "old_to_new"], dtype=np.float32)
    assert np.array_equal(s.chunk_lengths, a[p + "sp_chunk_lengths"])
    assert np.array_equal(s.col_idxs, a[p + "sp_col_idxs_pre"])
    assert np.array_equal(s.values, a[p + "sp_values"])
    assert np.array_equal(s.old_to_new_idx, a[p + "sp_old_to_new"])  # identity quirk of the reference


def test_halo_discovery_and_wsa(orc):
    h = golden("halo.npz")
    meta = json.load(open(os.path.join(GOLDEN, "halo_meta.json")))
    for key, m in meta.items():
        name, Cs, ss, method, Ps = key.rsplit("_", 4)
        Cc, sg, P = int(Cs[1:]), int(ss[1:]), int(Ps[1:])
        g = golden(f"scs_{name}.npz") if os.path.exists(os.path.join(GOLDEN, f"scs_{name}.npz")) else None
        I, J, V = g["I"], g["J"], g["vals"]
        wsa = orc.seg_work_sharing_arr(method, int(g["n_rows"]), I, P)
        assert np.array_equal(wsa, h[key + "_wsa"]), key
        for r in range(P):
            sel = (I >= wsa[r]) & (I < wsa[r + 1])
            s = orc.convert_to_scs(int(wsa[r + 1] - wsa[r]), int(g["n_cols"]), I[sel] - wsa[r], J[sel], V[sel], Cc, sg)
            assert s.n_elements == m["n_elements"][r] and np.array_equal(s.chunk_lengths, h[f"{key}_r{r}_chunk_lengths"])
            # use the reference's row order (tie order pinned by the golden) so col_idxs are comparable
            s = orc.convert_to_scs(int(wsa[r + 1] - wsa[r]), int(g["n_cols"]), I[sel] - wsa[r], J[sel], V[sel], Cc, sg,
                                   fixed_perm=h[f"{key}_r{r}_old_to_new"])
            nh, recv, cum = orc.collect_local_needed_heri(s.col_idxs, wsa, r, P, int(g["n_cols"]))
            assert nh == m["n_halo"][r] and [len(v) for v in recv] == m["recv_counts"][r]
            assert np.array_equal(cum, h[f"{key}_r{r}_recv_cumsum"])
            assert np.array_equal(np.concatenate(recv) if nh else np.zeros(0, np.int32), h[f"{key}_r{r}_recv_idxs"])
            orc.permute_scs_cols(s, h[f"{key}_r{r}_old_to_new"])
            assert np.array_equal(s.col_idxs, h[f"{key}_r{r}_col_idxs"])
            y = orc.spmv_scs(Cc, s.n_chunks, s.chunk_ptrs, s.chunk_lengths, s.col_idxs, s.values,
                             h[f"{key}_r{r}_x_local"])
            yo = orc.apply_permutation(y, h[f"{key}_r{r}_old_to_new"])
            assert np.array_equal(yo, h[key + "_y_global"][wsa[r]:wsa[r + 1]])


def test_pack_send_buf(orc):
    x = make_x(100)
    perm = np.random.default_rng(1).permutation(100).astype(np.int32)
    idx = np.array([5, 7, 7, 99, 0], np.int32)
    assert np.array_equal(orc.pack_send_buf(x, perm, idx), x[perm[idx]])


def _unit_fixture_cases():
    fx = json.load(open(os.path.join(GOLDEN, "reference_unit_fixtures.json")))
    cs = fx["_c_sigma"]
    for name, exp in sorted(fx.items()):
        if name.startswith("_") or exp["kind"] != "scs_explicit" or name.endswith("_compressed"):
            continue
        base = name.replace("explicit_exp_", "").rsplit("_scs_", 1)[0]
        if base.startswith("p0_") or base.startswith("p1_"):
            continue  # fake-rank variants encode an older column-compression convention
        coo = fx.get("exp_" + base) or fx.get(base)
        if coo is None:
            continue
        yield name, coo, exp, cs[name.replace("explicit_", "")]


def check_unit_fixture(convert, name, coo, exp, c_sigma):
    """convert(n_rows, n_cols, I, J, vals, C, sigma, np_dtype) -> object with the SCS arrays."""
    Cc, sg = c_sigma
    nnz = coo["nnz"]
    vals = list(coo["values"]) + [0.0] * (nnz - len(coo["values"]))  # (M1_be_row lists 5 of its 6 values)
    npdt = np.float32 if coo["vt"] == "float" else np.float64
    s = convert(coo["n_rows"], coo["n_cols"], coo["I"][:nnz], coo["J"][:nnz], vals, Cc, sg, npdt)
    assert list(s.chunk_lengths) == exp["chunk_lengths"], name
    assert list(s.chunk_ptrs) == exp["chunk_ptrs"], name
    # <= 16 rows per window: std::sort degenerates to (stable) insertion sort, so even the tie
    # order is comparable with a stable restatement
    assert list(s.col_idxs) == exp["col_idxs"], name
    assert list(s.old_to_new_idx) == exp["old_to_new_idx"], name
    assert list(s.new_to_old_idx) == exp["new_to_old_idx"], name
    n = len(coo["values"])
    assert np.array_equal(np.asarray(s.values)[:n], np.asarray(exp["values"], npdt)[:n]), name


def test_reference_unit_fixtures(orc):
    """The reference's own Catch2 expectations (code/test_suite/tests.cpp:8-275): precision split
    at threshold 1.0, convert_to_scs with C=1 on M1 variants (empty rows / columns) and on the
    hp / lp parts of M_big with sigma in {2,128}."""
    fx = json.load(open(os.path.join(GOLDEN, "reference_unit_fixtures.json")))
    n = 0
    for name, coo, exp, cs in _unit_fixture_cases():
        check_unit_fixture(lambda nr, nc, I, J, v, Cc, sg, dt: orc.convert_to_scs(nr, nc, I, J, v, Cc, sg, dtype=dt),
                           name, coo, exp, cs)
        n += 1
    assert n >= 13
    for base in ("M1", "M_big"):  # tests.cpp:8-24
        m = orc.partition_precisions_dpsp(np.array(fx[base]["values"]), 1.0)
        hp, lp = fx[f"exp_{base}_hp"], fx[f"exp_{base}_lp"]
        assert np.array(fx[base]["I"])[m].tolist() == hp["I"] and np.array(fx[base]["J"])[m].tolist() == hp["J"]
        assert np.array(fx[base]["values"])[m].tolist() == hp["values"]
        assert np.array(fx[base]["I"])[~m].tolist() == lp["I"] and np.array(fx[base]["J"])[~m].tolist() == lp["J"]
        assert np.array(fx[base]["values"])[~m].astype(np.float32).tolist() == np.array(lp["values"], np.float32).tolist()
