"""Seeded random matrices through every plan kind, against the oracle, bit for bit.  The golden matrices of the reference's
tests are small and regular; this adds what they do not have: empty rows, rows far longer than their neighbours, n that is
no multiple of C or of a tile, one-row and one-chunk matrices, unsorted entry order in the COO input, duplicate-free random
patterns from banded to uniform, values over many decades (for the ap split) -- with C in {1 .. 128}, sigma in {1, C, 512}.
Reference loops: code/kernels.hpp:216-258 (SpMV), :306-398 (SpMMV), code/ap_kernels.hpp:24-82 (ap)."""
import numpy as np
import pytest

from conftest import block_x

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda(pkg):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    torch.cuda.set_device(0)
    yield torch
    pkg.set_tuning(spmmv_variant=0, sweep=1, tlc=1)


def random_coo(pkg, rng, n, kind):
    """kind: 'uniform' (columns anywhere), 'banded', 'powerlaw' (a few very long rows), 'blocks' (dense diagonal blocks)."""
    rows, cols = [], []
    for i in range(n):
        if kind == "powerlaw":
            k = int(min(n, rng.pareto(1.2) * 3)) if rng.random() > 0.15 else 0
        elif kind == "blocks":
            k = 0
        else:
            k = int(rng.integers(0, 12)) if rng.random() > 0.1 else 0
        if kind == "banded":
            lo, hi = max(0, i - 40), min(n, i + 41)
            c = rng.choice(np.arange(lo, hi), size=min(k, hi - lo), replace=False)
        elif kind == "blocks":
            b0 = (i // 24) * 24
            c = np.arange(b0, min(n, b0 + 24))
        else:
            c = rng.choice(n, size=min(k, n), replace=False)
        rows.append(np.full(len(c), i)); cols.append(c)
    I = np.concatenate(rows) if rows else np.zeros(0, np.int64)
    J = np.concatenate(cols) if cols else np.zeros(0, np.int64)
    if len(I) == 0:
        I = np.array([0]); J = np.array([0])
    V = rng.uniform(-1.0, 1.0, len(I)) * 10.0 ** rng.uniform(-6.0, 3.0, len(I))
    p = rng.permutation(len(I))                      # entry order of the input must not matter
    return pkg.Coo.from_arrays(n, n, I[p], J[p], V[p])


def prep(pkg, coo, C, sigma, code, fixed=None):
    s = pkg.convert_to_scs(coo, C, sigma, code, fixed_permutation=fixed)
    a = s.arrays()
    pkg.permute_scs_cols(s, a["old_to_new_idx"] if fixed is None else fixed)
    return s, s.arrays()


CASES = [(0, 1, "uniform"), (1, 63, "uniform"), (2, 64, "banded"), (3, 65, "powerlaw"), (4, 700, "powerlaw"), (5, 1500, "banded"),
         (6, 2049, "uniform"), (7, 3000, "blocks"), (8, 5000, "powerlaw")]


@pytest.mark.parametrize("seed,n,kind", CASES)
def test_random_matrix_all_plans_bitexact(pkg, orc, torch_cuda, seed, n, kind):
    t = torch_cuda
    rng = np.random.default_rng(1000 + seed)
    coo = random_coo(pkg, rng, n, kind)
    for C, sigma in ((1, 1), (4, 1), (16, 16), (32, 512), (64, 64), (128, 512), (32, 1)):
        for code in (pkg.F64, pkg.F32):
            s, a = prep(pkg, coo, C, sigma, code)
            xp = np.zeros(s.n_rows_padded, s.np_dtype)
            xp[:s.n_rows] = rng.uniform(-2.0, 2.0, s.n_rows).astype(s.np_dtype)
            y_or = orc.spmv_scs(C, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], xp)
            x = t.from_numpy(xp).cuda()
            A = pkg.DeviceMatrix(s)
            tag = (seed, n, kind, C, sigma, code)
            # plain handle, automatic plan, forced sweep plan (where the struct admits one), plan built from the device arrays
            for what in ("none", "auto", "sweep", "device"):
                if what == "auto":
                    A.optimize(s)
                elif what == "sweep":
                    A.optimize_sweep(s, 8, 256)
                elif what == "device":
                    A = pkg.DeviceMatrix(s)
                    A.optimize_device()
                y = t.full((s.n_rows_padded,), -7.0, dtype=A.torch_dtype, device="cuda")
                pkg.spmv(A, x, y)
                got = y.cpu().numpy()
                if what == "device" and C < 32:          # (narrow chunks are re-chunked to C = 32 there: same chains, rows beyond n_rows_padded untouched)
                    assert np.array_equal(got[:s.n_rows_padded], y_or), tag + (what,)
                else:
                    assert np.array_equal(got, y_or), tag + (what,)
            # block vectors: the gather kernels and (C = 32 / 64) the block plans
            for b in (1, 3, 8, 16):
                ld = s.n_rows_padded + 3
                Ab = pkg.DeviceMatrix(s, block_tlc=b)
                for rowwise in (0, 1):
                    X = block_x(xp, s.n_rows_padded, b, ld, rowwise)
                    Yo = orc.spmmv_scs(C, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], X, b, ld, rowwise)
                    lay = pkg.ROWWISE if rowwise else pkg.COLWISE
                    Y = t.zeros((b * ld,), dtype=Ab.torch_dtype, device="cuda")
                    pkg.spmmv(Ab, t.from_numpy(X).cuda(), Y, b, ld, lay)
                    g = Y.cpu().numpy()
                    npad = s.n_rows_padded
                    if rowwise:
                        assert np.array_equal(g[:npad * b], Yo[:npad * b]), tag + ("spmmv", b, rowwise)
                    else:
                        assert np.array_equal(g.reshape(b, ld)[:, :npad], Yo.reshape(b, ld)[:, :npad]), tag + ("spmmv", b, rowwise)


@pytest.mark.parametrize("seed,n,kind", [c for c in CASES if c[1] >= 63])
def test_random_matrix_ap_bitexact(pkg, orc, torch_cuda, seed, n, kind):
    t = torch_cuda
    rng = np.random.default_rng(2000 + seed)
    coo = random_coo(pkg, rng, n, kind)
    dp, sp = pkg.partition_precisions(coo, 1e-2)
    if dp.nnz == 0 or sp.nnz == 0:
        pytest.skip("one-sided split")
    done = 0
    for C, sigma in ((32, 512), (64, 64), (16, 1), (4, 4)):
        ds = pkg.convert_to_scs(dp, C, sigma, pkg.F64)
        perm = ds.arrays()["old_to_new_idx"].copy()
        try:
            ss = pkg.convert_to_scs(sp, C, sigma, pkg.F32, fixed_permutation=perm)
        except pkg.UspmvError:                           # the dp part's permutation puts a non-empty sp row on a padded slot: the
            continue                                     # reference overruns its chunk there (code/utilities.hpp:1919-1922), we refuse
        done += 1
        pkg.permute_scs_cols(ds, perm); pkg.permute_scs_cols(ss, perm)
        da, sa = ds.arrays(), ss.arrays()
        xp = np.zeros(ds.n_rows_padded)
        xp[:ds.n_rows] = rng.uniform(-2.0, 2.0, ds.n_rows)
        y_or = orc.spmv_scs_ap_adv(C, ds.n_chunks, (da["chunk_ptrs"], da["chunk_lengths"], da["col_idxs"], da["values"]),
                                   (sa["chunk_ptrs"], sa["chunk_lengths"], sa["col_idxs"], sa["values"]), xp)
        x = t.from_numpy(xp).cuda()
        for what in ("none", "auto", "sweep", "device"):
            Ad, As = pkg.DeviceMatrix(ds), pkg.DeviceMatrix(ss)
            if what == "auto":
                pkg.optimize_ap(Ad, As, ds, ss)
            elif what == "sweep":
                pkg.optimize_sweep_ap(Ad, As, ds, ss, 8, 256)
            elif what == "device":
                pkg.optimize_device_ap(Ad, As)
            y = t.full((ds.n_rows_padded,), -7.0, dtype=t.float64, device="cuda")
            pkg.spmv_ap(Ad, As, x, y)
            assert np.array_equal(y.cpu().numpy()[:ds.n_rows_padded], y_or), (seed, n, kind, C, sigma, what)
    assert done > 0
