"""Randomised cross-checks against the genuine reference (oracle/_ref, built from /root/reference by oracle/Makefile):
random COO matrices with heavy length ties, empty rows and ragged shapes through convert_to_scs
(code/utilities.hpp:1842-2104, incl. the std::sort tie order), the dp+sp split (:2899-2911) and -- on the GPU --
the kernels.  Seeds are fixed; the CPU part needs no GPU."""
import numpy as np
import pytest

from oracle import refshim

@pytest.fixture(scope="module")
def torch_cuda(pkg):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    torch.cuda.set_device(0)
    return torch


pytestmark = pytest.mark.skipif(not refshim.available("colwise"), reason="oracle/_ref not built (reference sources absent)")


def random_coo(rng, n_rows, n_cols, density, tie_heavy):
    if tie_heavy:                      # few distinct row lengths -> many ties inside every sigma window
        lens = rng.choice([0, 1, 2, 2, 3, 3, 3, 7], n_rows)
    else:
        lens = rng.poisson(density * n_cols, n_rows)
    lens = np.minimum(lens, n_cols)
    I = np.repeat(np.arange(n_rows), lens)
    J = np.concatenate([rng.choice(n_cols, k, replace=False) for k in lens]) if I.size else np.zeros(0, np.int64)
    V = rng.standard_normal(I.size) * 10.0 ** rng.integers(-6, 3, I.size)
    return I.astype(np.int32), J.astype(np.int32), V


CASES = [(seed, n, C, sigma) for seed, (n, C, sigma) in enumerate([
    (1, 1, 1), (2, 4, 4), (17, 4, 8), (33, 32, 64), (64, 32, 512), (97, 16, 48), (130, 64, 128), (257, 128, 256),
    (300, 8, 8), (301, 5, 15), (511, 32, 32), (777, 2, 1000), (1000, 32, 512), (1025, 64, 64)])]


@pytest.mark.parametrize("seed,n,C,sigma", CASES)
def test_convert_and_split_match_reference(pkg, seed, n, C, sigma):
    rng = np.random.default_rng(1000 + seed)
    n_cols = n if seed % 3 else n + 7
    I, J, V = random_coo(rng, n, n_cols, 0.05, tie_heavy=bool(seed % 2))
    if I.size == 0:
        I, J, V = np.array([0], np.int32), np.array([0], np.int32), np.array([1.5])
    m = pkg.Coo.from_arrays(n, n_cols, I, J, V)
    rm = refshim.RefMtx.from_coo(n, n_cols, I, J, V)
    for dt, code in (("f64", pkg.F64), ("f32", pkg.F32)):
        s = pkg.convert_to_scs(m, C, sigma, code)
        r = refshim.convert_to_scs(rm, C, sigma, dt)
        a, ra = s.arrays(), r.arrays()
        assert (s.n_chunks, s.n_elements, s.n_rows_padded) == (r.n_chunks, r.n_elements, r.n_rows_padded)
        for k in ("chunk_ptrs", "chunk_lengths", "col_idxs", "values", "old_to_new_idx"):
            assert np.array_equal(a[k], ra[k]), (seed, dt, k)
        # new_to_old_idx: the reference leaves the slots that received a padding row uninitialised (raw new[],
        # code/utilities.hpp:2060-2069), so only the written ones are comparable
        written = a["old_to_new_idx"][a["old_to_new_idx"] < n]
        assert np.array_equal(a["new_to_old_idx"][written], ra["new_to_old_idx"][written]), (seed, dt, "new_to_old_idx")
    th = float(np.median(np.abs(V)))
    dp, sp = pkg.partition_precisions(m, th)
    rdp, _, (sI, sJ, sV) = refshim.partition_precisions_dpsp(rm, th)       # _ = raw MtxData<float,int>* of the sp part
    rI, rJ, rV = rdp.arrays()
    I2, J2, V2 = dp.arrays()
    assert np.array_equal(I2, rI) and np.array_equal(J2, rJ) and np.array_equal(V2, rV)
    I2, J2, V2 = sp.arrays()
    assert np.array_equal(I2, sI) and np.array_equal(J2, sJ) and np.array_equal(V2.astype(np.float32), sV)
    # the ap[dp_sp] structs: dp sorted on its own, sp placed with the dp permutation (code/main.cpp:1156-1160),
    # including the quirk that such a struct reports the identity as old_to_new_idx
    if dp.nnz and sp.nnz and n >= C:
        ds = pkg.convert_to_scs(dp, C, sigma, pkg.F64)
        rds = refshim.convert_to_scs(rdp, C, sigma, "f64")
        perm = ds.arrays()["old_to_new_idx"].copy()
        if np.all(perm < n):          # (a permutation that moves a row into a padding slot overruns the reference's chunk)
            try:
                ss = pkg.convert_to_scs(sp, C, sigma, pkg.F32, fixed_permutation=perm)
            except pkg.UspmvError:
                return                # the product refuses what the reference would overrun (non-empty row on a shorter chunk)
            rss = refshim.convert_to_scs(_, C, sigma, "f32", fixed_perm=rds.arrays()["old_to_new_idx"])
            for k in ("chunk_ptrs", "chunk_lengths", "col_idxs", "values", "old_to_new_idx"):
                assert np.array_equal(ss.arrays()[k], rss.arrays()[k]), (seed, "sp struct", k)


@pytest.mark.parametrize("name", ["bcsstk13", "impcol_e", "FDM-2d-16", None])
def test_equilibrate_matches_reference_and_oracle(pkg, orc, name):
    """uspmv_coo_equilibrate and the oracle's restatement against equilibrate_matrix of the genuine reference
    (code/utilities.hpp:2667-2685): bit-identical values."""
    from conftest import mtx_path
    if name is None:
        rng = np.random.default_rng(77)
        I, J, V = random_coo(rng, 200, 200, 0.05, tie_heavy=False)
        m = pkg.Coo.from_arrays(200, 200, I, J, V)
    else:
        m = pkg.read_mtx(mtx_path(name))
    I, J, V = [a.copy() for a in m.arrays()]
    rm = refshim.RefMtx.from_coo(m.n_rows, m.n_cols, I, J, V)
    if not hasattr(refshim.lib("colwise"), "ref_equilibrate_matrix"):
        pytest.skip("oracle/_ref predates ref_equilibrate_matrix")
    rm.equilibrate()
    ref_vals = rm.arrays()[2]
    assert np.array_equal(orc.equilibrate_matrix(m.n_rows, m.n_cols, I, J, V), ref_vals)
    m.equilibrate()
    assert np.array_equal(m.arrays()[2], ref_vals)
    assert np.array_equal(m.arrays()[0], I) and np.array_equal(m.arrays()[1], J)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,C,sigma", CASES[3:])
def test_kernels_match_reference_on_random_matrices(pkg, torch_cuda, seed, n, C, sigma):
    t = torch_cuda
    rng = np.random.default_rng(2000 + seed)
    I, J, V = random_coo(rng, n, n, 0.05, tie_heavy=bool(seed % 2))
    m = pkg.Coo.from_arrays(n, n, I, J, V)
    x0 = rng.standard_normal(n)
    for dt, code, kind in (("f64", pkg.F64, "adv" if C in (2, 4, 8, 16, 32, 64, 128) else "gen"), ("f32", pkg.F32, "gen")):
        s = pkg.convert_to_scs(m, C, sigma, code)
        pkg.permute_scs_cols(s, s.arrays()["old_to_new_idx"])
        a = s.arrays()
        xp = np.zeros(max(s.n_rows_padded, n), s.np_dtype)
        xp[:n] = pkg.apply_permutation(x0.astype(s.np_dtype), a["new_to_old_idx"])
        yr = refshim.spmv_scs(kind, C, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], xp)
        for tlc in (False, True):
            A = pkg.DeviceMatrix(s, tlc=tlc)
            y = t.full((s.n_rows_padded,), 7.0, dtype=A.torch_dtype, device="cuda")
            pkg.spmv(A, t.from_numpy(xp).cuda(), y)
            assert np.array_equal(y.cpu().numpy(), yr), (seed, dt, tlc)
        b, ld = 4, s.n_rows_padded + 3
        X = np.zeros(b * ld, s.np_dtype)
        for v in range(b):
            X[v * ld: v * ld + len(xp[:s.n_rows_padded])] = xp[:s.n_rows_padded] * (1 + v)
        Yr = refshim.spmmv_scs_general(C, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], X, b, ld, 0)
        Y = t.zeros(b * ld, dtype=A.torch_dtype, device="cuda")
        pkg.spmmv(A, t.from_numpy(X).cuda(), Y, b, ld, pkg.COLWISE)
        assert np.array_equal(Y.cpu().numpy().reshape(b, ld)[:, :s.n_rows_padded], Yr.reshape(b, ld)[:, :s.n_rows_padded]), (seed, dt, "spmmv")


@pytest.mark.skipif(not refshim.available("mpi"), reason="oracle/_ref mpi variant not built")
@pytest.mark.parametrize("seed,n,P,method,C,sigma", [(0, 300, 2, "seg-rows", 32, 512), (1, 301, 3, "seg-nnz", 16, 64), (2, 640, 5, "seg-nnz", 32, 32),
                                                    (3, 97, 4, "seg-rows", 4, 8), (4, 1000, 8, "seg-nnz", 64, 128), (5, 513, 2, "seg-nnz", 1, 1)])
def test_partition_and_halo_discovery_match_reference(pkg, seed, n, P, method, C, sigma):
    """seg_work_sharing_arr (code/mpi_funcs.hpp:424-622), seg_mtx_struct / localize_row_idx (:636-674, :862-877) and
    collect_local_needed_heri (:242-415) of the genuine reference (fake-rank driver, no MPI_Init) on random matrices:
    work-sharing array, local COO blocks, rewritten col_idxs, per-owner receive lists and their cumulative sums."""
    from ultimate_spmv_amd import binding as B
    rng = np.random.default_rng(3000 + seed)
    I, J, V = random_coo(rng, n, n, 0.04, tie_heavy=bool(seed % 2))
    # every row gets its diagonal: with empty rows in a block the reference itself is inconsistent (local n_rows =
    # number of DISTINCT NON-EMPTY rows, row ids shifted by the first non-empty row, code/mpi_funcs.hpp:770, :862-877,
    # while the vectors are sized by work_sharing_arr, code/main.cpp:1299) -- the product uses the block height
    has_diag = np.zeros(n, bool); has_diag[I[I == J]] = True
    add = np.flatnonzero(~has_diag)
    I = np.concatenate([I, add]); J = np.concatenate([J, add]); V = np.concatenate([V, np.full(add.size, 2.5)])
    o = np.lexsort((J, I)); I, J, V = I[o].astype(np.int32), J[o].astype(np.int32), V[o]
    m = pkg.Coo.from_arrays(n, n, I, J, V)
    rm = refshim.RefMtx.from_coo(n, n, I, J, V, "mpi")
    wsa_ref = refshim.seg_work_sharing_arr(rm, method, P)
    wsa = pkg.seg_work_sharing_arr(m, method, P)
    assert np.array_equal(wsa, wsa_ref)
    for r in range(P):
        if wsa[r + 1] == wsa[r]:
            continue
        loc_ref = refshim.seg_local_mtx(rm, wsa, r)
        loc = B.seg_local_coo(m, wsa, r)
        for a, b in zip(loc.arrays(), loc_ref.arrays()):
            assert np.array_equal(a, b)
        s = pkg.convert_to_scs(loc, C, sigma, pkg.F64)
        rs = refshim.convert_to_scs(loc_ref, C, sigma, "f64")
        nh, recv_ref, cum_ref = refshim.collect_local_needed_heri(rs, wsa, r, P)
        plan = B.HaloPlan(s, wsa, r, P)
        assert plan.n_halo == nh and np.array_equal(plan.recv_counts_cumsum, cum_ref)
        for p in range(P):
            assert np.array_equal(plan.recv_idxs_of(p), recv_ref[p]), (r, p)
        assert np.array_equal(s.arrays()["col_idxs"], rs.arrays()["col_idxs"]), r


def _same_coo(pkg, path):
    a = pkg.read_mtx(path)
    r = refshim.RefMtx.read(path)
    I, J, V = a.arrays()
    rI, rJ, rV = r.arrays()
    assert (a.n_rows, a.nnz) == (r.n_rows, r.nnz)
    assert np.array_equal(I, rI) and np.array_equal(J, rJ) and np.array_equal(V.view(np.uint64), rV.view(np.uint64))
    return a


def test_read_mtx_number_spellings_like_the_reference(pkg, tmp_path):
    """uspmv_read_mtx parses plain decimals with std::from_chars and everything else with strtod; the reference reads every value with
    fscanf("%lg") (code/mmio.h:132-263).  Spellings on either side of that split must give the same bits."""
    vals = ["1", "-1", "+1.5", "1.", ".5", "-.5e-3", "1e400", "-1e400", "1e-400", "4.9406564584124654e-324", "1.7976931348623157e308",
            "0.1", "0.30000000000000004", "123456789012345678901234567890", "0x1p3", "-0x1.8p-2", "inf", "-inf", "INF", "1E5", "1e+05",
            "00012.5", "-0", "0", "2.2250738585072011e-308", "9007199254740993", "5e-324", "1.0000000000000002"]
    n = len(vals)
    p = tmp_path / "spellings.mtx"
    with open(p, "w") as f:
        f.write(f"%%MatrixMarket matrix coordinate real general\n% odd spellings\n{n} {n} {n}\n")
        for i, v in enumerate(vals):
            f.write(f"{i + 1}   {(i * 7) % n + 1}\t{v}\n")
    _same_coo(pkg, str(p))


def test_read_mtx_large_symmetric_file_parallel_path(pkg, tmp_path):
    """A file large enough for the multi-threaded parse and the two-pass stable bucket sort (>= 1 MiB of entries, more rows than one
    bucket): identical COO -- same expansion order, same order inside every row -- as the reference's read_mtx; also with more lines
    than the header announces, and through our own writer."""
    rng = np.random.default_rng(77)
    n, k = 30000, 120000
    I = rng.integers(0, n, k); J = rng.integers(0, n, k)
    lo = np.maximum(I, J); hi = np.minimum(I, J)
    V = rng.standard_normal(k) * 10.0 ** rng.integers(-8, 8, k)
    p = tmp_path / "big_sym.mtx"
    with open(p, "w") as f:
        f.write(f"%%MatrixMarket matrix coordinate real symmetric\n{n} {n} {k - 5}\n")       # (five surplus lines at the end)
        for a, b, v in zip(lo, hi, V):
            f.write(f"{a + 1} {b + 1} {float(v)!r}\n")
    assert p.stat().st_size > (1 << 20)
    a = _same_coo(pkg, str(p))
    q = tmp_path / "rewritten.mtx"
    a.write_mtx(str(q), symmetric=False)
    b = pkg.read_mtx(str(q))
    for u, w in zip(a.arrays(), b.arrays()):
        assert np.array_equal(u, w)
    _same_coo(pkg, str(q))
