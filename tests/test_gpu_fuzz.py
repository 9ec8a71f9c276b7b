"""Randomised GPU parity: random matrices (ragged, empty rows, heavy ties, banded, a few very long rows, special values) x chunk
heights x sorting scopes x precisions through every way a handle can be set up -- plain upload, host plans, device plans, device
conversion, block plans, adaptive-precision pairs, the column-window sweep -- each result compared BITWISE with the oracle's loops
(oracle/uspmv_oracle.c = code/kernels.hpp:159-398, code/ap_kernels.hpp:24-82).  Seeds are fixed; USPMV_FUZZ_CASES raises the number
of cases for a long run (tools/README.md; profiles/r03/fuzz.txt holds one)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_CASES = int(os.environ.get("USPMV_FUZZ_CASES", "36"))
SEED0 = int(os.environ.get("USPMV_FUZZ_SEED", "20260"))


@pytest.fixture(scope="module")
def torch_cuda(pkg):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    torch.cuda.set_device(0)
    return torch


def random_matrix(rng):
    kind = rng.choice(["poisson", "ties", "banded", "longrows", "tiny", "stencil"])
    n = int(rng.choice([1, 2, 5, 31, 32, 33, 100, 257, 700, 1500, 4100]))
    if kind == "tiny":
        n = int(rng.integers(1, 12))
    n_cols = n
    if kind == "poisson":
        lens = rng.poisson(rng.choice([1.0, 4.0, 12.0]), n)
    elif kind == "ties":
        lens = rng.choice([0, 1, 2, 2, 3, 3, 3, 7], n)
    elif kind == "banded":
        lens = rng.integers(1, 9, n)
    elif kind == "longrows":
        lens = rng.poisson(3.0, n)
        for r in rng.choice(n, max(1, n // 200), replace=False):
            lens[r] = min(n_cols, int(rng.integers(100, 400)))
    elif kind == "stencil":
        lens = np.full(n, 5)
    else:
        lens = rng.integers(0, 4, n)
    lens = np.minimum(lens, n_cols).astype(np.int64)
    I = np.repeat(np.arange(n), lens)
    cols = []
    for r, k in enumerate(lens):
        if k == 0:
            continue
        if kind in ("banded", "stencil"):
            w = 40 if kind == "banded" else 3
            lo, hi = max(0, r - w), min(n_cols, r + w + 1)
            k2 = min(k, hi - lo)
            c = np.sort(rng.choice(np.arange(lo, hi), k2, replace=False))
            if k2 < k:
                c = np.concatenate([c, c[:k - k2]])[:k]
        else:
            c = rng.choice(n_cols, k, replace=False)
            if rng.random() < 0.5:
                c = np.sort(c)
        cols.append(c)
    J = np.concatenate(cols) if cols else np.zeros(0, np.int64)
    if J.size != I.size:            # (a banded row shorter than asked for)
        I = I[:J.size]
    V = rng.standard_normal(I.size) * 10.0 ** rng.integers(-6, 4, I.size)
    if I.size and rng.random() < 0.3:
        V[rng.choice(I.size, max(1, I.size // 50), replace=False)] = 0.0      # explicit zeros
    if I.size == 0:
        I, J, V = np.array([0]), np.array([0]), np.array([1.5])
    return kind, n, n_cols, I.astype(np.int32), J.astype(np.int32), V.astype(np.float64)


def _dev(t, a):
    return t.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("case", range(N_CASES))
def test_random_matrix_every_setup_bitexact(pkg, orc, torch_cuda, case):
    t = torch_cuda
    rng = np.random.default_rng(SEED0 + case)
    kind, n, n_cols, I, J, V = random_matrix(rng)
    C = int(rng.choice([1, 2, 4, 8, 16, 32, 32, 32, 64, 128, 10, 3]))
    sigma = int(rng.choice([1, C, 2 * C, 512, 1000]))
    f64 = rng.random() < 0.6
    m = pkg.Coo.from_arrays(n, n_cols, I, J, V)
    check_every_setup(pkg, orc, t, rng, m, C, sigma, f64, (case, kind, n, C, sigma, "f64" if f64 else "f32"))


GENERATED = [   # (name, generator, C, sigma, f64[, what the automatic choices must come out as]): sizes at which tiles, plans and the automatic choices between them are all in play
    ("stencil27 40x40x30", lambda pkg: pkg.gen_stencil27(40, 40, 30), 32, 512, True, dict(kind=1, tile_rows=256)),
    ("stencil27 24^3 x 3 dof", lambda pkg: pkg.gen_stencil27(24, 24, 24, dof=3), 32, 512, True, dict(b=8, phased=True)),
    ("stencil27 24^3 x 3 dof sp C64", lambda pkg: pkg.gen_stencil27(24, 24, 24, dof=3), 64, 64, False),
    ("stencil27 slab 150x150x2", lambda pkg: pkg.gen_stencil27(150, 150, 2), 32, 512, True),
    ("stencil27 30^3 C16", lambda pkg: pkg.gen_stencil27(30, 30, 30), 16, 16, True),
    ("stencil27 30^3 C8 sigma 64", lambda pkg: pkg.gen_stencil27(30, 30, 30), 8, 64, False),
    ("kkt N=14", lambda pkg: pkg.gen_kkt(14), 32, 512, True),
    ("kkt N=12 C64", lambda pkg: pkg.gen_kkt(12), 64, 512, True),
    ("banded 30000 x 30 +-2000 (1024-row tiles)", lambda pkg: pkg.gen_banded_random(30000, 30, 2000), 32, 512, True, dict(kind=1, tile_rows=1024)),
    ("banded 20000 x 140 +-9000 (sweep)", lambda pkg: pkg.gen_banded_random(20000, 140, 9000, magnitude_decades=8.0), 32, 512, True, dict(kind=2)),
    ("banded 20000 x 140 +-9000 sp", lambda pkg: pkg.gen_banded_random(20000, 140, 9000), 32, 512, False),
    ("banded 50000 x 12 +-300", lambda pkg: pkg.gen_banded_random(50000, 12, 300), 32, 1, True),
    # 1.3 M rows: the padding column lies > 2^20 columns below the band of the last tiles (the device builder's window mode)
    ("banded 1300000 x 5 +-100 (far padding column)", lambda pkg: pkg.gen_banded_random(1300000, 5, 100), 32, 512, True, dict(kind=1)),
    ("banded 1300000 x 5 +-100 sp C64", lambda pkg: pkg.gen_banded_random(1300000, 5, 100), 64, 128, False, dict(kind=1)),
]


@pytest.mark.parametrize("spec", GENERATED, ids=[g[0] for g in GENERATED])
def test_generated_matrix_every_setup_bitexact(pkg, orc, torch_cuda, spec):
    name, gen, C, sigma, f64 = spec[:5]
    rng = np.random.default_rng(sum(map(ord, name)))
    check_every_setup(pkg, orc, torch_cuda, rng, gen(pkg), C, sigma, f64, (name, C, sigma, "f64" if f64 else "f32"), spec[5] if len(spec) > 5 else {})


def check_every_setup(pkg, orc, t, rng, m, C, sigma, f64, tag, expect=None):
    expect = expect or {}
    code, ndt, tdt = (pkg.F64, np.float64, t.float64) if f64 else (pkg.F32, np.float32, t.float32)
    n, n_cols = m.n_rows, m.n_cols
    V = m.arrays()[2]
    s = pkg.convert_to_scs(m, C, sigma, code)
    a = s.arrays()
    pkg.permute_scs_cols(s, a["old_to_new_idx"])
    a = s.arrays()
    ld = s.n_rows_padded
    xo = (rng.standard_normal(n) * 10.0 ** rng.integers(-3, 3, n)).astype(ndt)
    xp = np.zeros(ld, ndt)
    xp[:n] = pkg.apply_permutation(xo, a["new_to_old_idx"])
    want = orc.spmv_scs(C, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], xp)

    def check_spmv(A, what):
        y = t.full((ld,), 3.0, dtype=tdt, device="cuda")
        pkg.spmv(A, _dev(t, xp), y)
        assert np.array_equal(y.cpu().numpy(), want), tag + (what,)

    # ---- single vector: plain upload, host plan, device plan, device conversion (+ device plan), forced sweep plan
    A = pkg.DeviceMatrix(s); check_spmv(A, "upload")
    A.optimize(s); check_spmv(A, "host plan")
    A2 = pkg.DeviceMatrix(s); A2.optimize_device(); check_spmv(A2, "device plan")
    assert A.plan_info() == A2.plan_info(), tag
    for i12 in (0, 2):                             # the plan's local indices as 16-bit words / packed to 12 bits wherever the tiles allow it
        pkg.set_tuning(tlc_idx12=i12)
        try:
            A8 = pkg.DeviceMatrix(s); A8.optimize(s) if i12 else A8.optimize_device()
        finally:
            pkg.set_tuning(tlc_idx12=1)
        assert A8.index_bits() in ((0, 16) if i12 == 0 else (0, 12, 16)), tag + (i12, A8.index_bits())
        check_spmv(A8, f"plan with tlc_idx12 = {i12} ({A8.index_bits()}-bit local indices)")
    if "kind" in expect: assert A.plan_info()[0] == expect["kind"], tag + (A.plan_info(),)
    if "tile_rows" in expect:
        import ctypes
        from ultimate_spmv_amd import binding
        tr = ctypes.c_int()
        assert binding.lib().uspmv_dmat_tile_rows(A.h, ctypes.byref(tr)) == 0 and tr.value == expect["tile_rows"], tag + (tr.value,)
    lay, A3 = pkg.convert_to_scs_device(m, C, sigma, code)
    assert np.array_equal(lay.arrays()["old_to_new_idx"], a["old_to_new_idx"]), tag
    check_spmv(A3, "device conversion"); A3.optimize_device(); check_spmv(A3, "device conversion + device plan")
    # ---- the conversion from DEVICE-resident COO arrays: the reference's tie order (every array bit-identical), and the stable device
    #      ordering (another tie order: compared through y in ORIGINAL row order, which does not depend on it)
    I_, J_, V_ = m.arrays()
    if np.all(np.diff(I_) >= 0):
        dI, dJ, dV = _dev(t, np.array(I_)), _dev(t, np.array(J_)), _dev(t, np.array(V_))
        lay6, A6, o2n6, n2o6 = pkg.convert_to_scs_device_from_arrays(dI, dJ, dV, n, n_cols, C, sigma, code)
        d6 = pkg.dmat_download(A6)
        for k in ("chunk_ptrs", "chunk_lengths", "col_idxs", "values"):
            assert np.array_equal(d6[k], a[k]), tag + ("conversion from device arrays", k)
        assert np.array_equal(o2n6.cpu().numpy(), a["old_to_new_idx"]), tag
        check_spmv(A6, "conversion from device arrays"); A6.optimize_device(); check_spmv(A6, "conversion from device arrays + device plan")
        if min(sigma, s.n_rows_padded) <= 8192:
            lay7, A7, o2n7, n2o7 = pkg.convert_to_scs_device_from_arrays(dI, dJ, dV, n, n_cols, C, sigma, code, sort=pkg.SORT_DEVICE_STABLE)
            assert np.array_equal(lay7.arrays()["chunk_ptrs"], a["chunk_ptrs"]) and np.array_equal(lay7.arrays()["chunk_lengths"], a["chunk_lengths"]), tag
            x7 = t.zeros(max(ld, n_cols), dtype=tdt, device="cuda"); x7[o2n7.long()] = _dev(t, xo)
            y7 = t.full((ld,), 3.0, dtype=tdt, device="cuda")
            A7.optimize_device()
            pkg.spmv(A7, x7, y7)
            xs_ = np.zeros(max(ld, n_cols), ndt); xs_[a["old_to_new_idx"]] = xo           # (scatter form: right even where the reference's tie order parks a real row behind n_rows)
            want7 = orc.spmv_scs(C, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], xs_)[a["old_to_new_idx"]]
            assert np.array_equal(y7[o2n7.long()].cpu().numpy(), want7), tag + ("stable device ordering",)
    if 256 % C == 0:
        A4 = pkg.DeviceMatrix(s)
        try:
            A4.optimize_sweep(s)
        except pkg.UspmvError:
            pass                                   # (shapes the sweep planner refuses keep their kernel)
        check_spmv(A4, "sweep plan (where tiles qualify)")
        A5 = pkg.DeviceMatrix(s)
        try:
            A5.optimize_sweep_device()
        except pkg.UspmvError:
            pass
        check_spmv(A5, "sweep plan built on the device")

    # ---- block vectors: both layouts, without a plan, with the host block plan, with the device block plan
    b = expect.get("b") or int(rng.choice([1, 2, 3, 4, 5, 8, 8, 13, 16]))
    cols_ = [(xp * ndt(1.0 + v / 8.0)).astype(ndt) for v in range(b)]
    handles = [("no plan", pkg.DeviceMatrix(s))]
    Ab = pkg.DeviceMatrix(s); Ab.optimize_block(s, b); handles.append(("host block plan", Ab))
    Ad = pkg.DeviceMatrix(s); Ad.optimize_block_device(b); handles.append(("device block plan", Ad))
    if b * (8 if f64 else 4) == 64 and C == 32:             # 64-byte X rows: the phased plan walked as a stream by persistent workgroups (both builders)
        sdepth = int(rng.choice([1, 2]))
        pkg.set_tuning(spmmv_stream=int(rng.choice([1, 2, 3, 4, 5, 99])), spmmv_stream_xcd=int(rng.choice([0, 1])))
        try:
            As = pkg.DeviceMatrix(s); As.optimize_block(s, b); handles.append(("host block plan, streamed", As))
            At = pkg.DeviceMatrix(s); At.optimize_block_device(b); handles.append(("device block plan, streamed", At))
        finally:
            pkg.set_tuning(spmmv_stream=0, spmmv_stream_xcd=1)
        # (the launch looks at the tuning too: the streamed handles are multiplied with it switched on, below)
    if b * (8 if f64 else 4) == 64 and 64 % C == 0:          # 64-byte X rows: the block-vector window sweep, where every tile qualifies
        Aw = pkg.DeviceMatrix(s)
        try:
            ntw, nsw = Aw.optimize_block_sweep(s, b, wlog=int(rng.choice([9, 10, 11])), tile_rows=int(rng.choice([1024, 2048, 4096])))
        except pkg.UspmvError:
            ntw, nsw = 1, 0
        if ntw == nsw:
            handles.append(("block window sweep", Aw))
    if expect.get("phased"):
        assert Ab.block_plan_info()["phased_plan"] == 1 and Ad.block_plan_info()["phased_plan"] == 1, tag + (Ab.block_plan_info(), Ad.block_plan_info())
    for lay_code, rowwise in ((pkg.COLWISE, False), (pkg.ROWWISE, True)):
        X = np.zeros(b * ld, ndt)
        for v in range(b):
            if rowwise: X[np.arange(ld) * b + v] = cols_[v]
            else: X[v * ld:(v + 1) * ld] = cols_[v]
        wantb = orc.spmmv_scs(C, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], X, b, ld, rowwise)
        for what, H in handles:
            Y = t.full((b * ld,), 3.0, dtype=tdt, device="cuda")
            pkg.set_tuning(spmmv_stream=1 if what.endswith("streamed") else 0, spmmv_stream_depth=sdepth if what.endswith("streamed") else 1)
            pkg.spmmv(H, _dev(t, X), Y, b, ld, lay_code)
            got = Y.cpu().numpy()
            n_out = s.n_rows_padded * b if rowwise else None
            if rowwise:
                assert np.array_equal(got[:n_out], wantb[:n_out]), tag + (what, "rowwise", b)
            else:
                for v in range(b):
                    assert np.array_equal(got[v * ld:v * ld + s.n_rows_padded], wantb[v * ld:v * ld + s.n_rows_padded]), tag + (what, "colwise", b, v)
                # the same X again, re-laid out once (uspmv_spmmv_x_prepared): same bits
                dXp = _dev(t, X)
                pkg.spmmv_x_prepared(H, dXp, b, ld)
                Y2 = t.full((b * ld,), 3.0, dtype=tdt, device="cuda")
                pkg.spmmv(H, dXp, Y2, b, ld, lay_code)
                pkg.spmmv_x_release(H)
                assert t.equal(Y2, Y), tag + (what, "colwise, X prepared", b)
    pkg.set_tuning(spmmv_stream=0, spmmv_stream_depth=1)

    # ---- adaptive precision pair (dp struct sorted on its own, sp struct placed with its permutation): plain, host plan, device plan
    if code == pkg.F64 and n >= C:
        th = float(np.median(np.abs(V))) if V.size else 1.0
        dp, sp = pkg.partition_precisions(m, th)
        if dp.nnz and sp.nnz:
            ds = pkg.convert_to_scs(dp, C, sigma, pkg.F64)
            perm = ds.arrays()["old_to_new_idx"].copy()
            if np.all(perm < n):
                try:
                    ss = pkg.convert_to_scs(sp, C, sigma, pkg.F32, fixed_permutation=perm)
                except pkg.UspmvError:
                    return                         # (a non-empty sp row on a shorter dp chunk: refused like the reference would overrun)
                pkg.permute_scs_cols(ds, perm); pkg.permute_scs_cols(ss, perm)
                da, sa = ds.arrays(), ss.arrays()
                xq = np.zeros(ds.n_rows_padded); xq[:n] = pkg.apply_permutation(xo.astype(np.float64), da["new_to_old_idx"])
                wanta = orc.spmv_scs_ap_adv(C, ds.n_chunks, (da["chunk_ptrs"], da["chunk_lengths"], da["col_idxs"], da["values"]),
                                            (sa["chunk_ptrs"], sa["chunk_lengths"], sa["col_idxs"], sa["values"]), xq)
                for what in ("plain", "host plan", "device plan"):
                    Pd, Ps = pkg.DeviceMatrix(ds), pkg.DeviceMatrix(ss)
                    if what == "host plan": pkg.optimize_ap(Pd, Ps, ds, ss)
                    elif what == "device plan": pkg.optimize_device_ap(Pd, Ps)
                    y = t.full((ds.n_rows_padded,), 3.0, dtype=t.float64, device="cuda")
                    pkg.spmv_ap(Pd, Ps, _dev(t, xq), y)
                    assert np.array_equal(y.cpu().numpy(), wanta), tag + ("ap", what)
