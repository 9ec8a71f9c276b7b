"""Column-window sweep kernels (csrc/sweep_kernels.hip) against the oracle, bit for bit: the HV15R-class banded-random
matrix of SURVEY.md 8(d) at reduced size (dp, sp, ap[dp_sp]), every window / tile / buffering / unroll variant, golden
matrices with partial coverage (rest chunks on the gather kernel), special values, and the automatic plan choice."""
import numpy as np
import pytest

from conftest import make_x, mtx_path

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda(pkg):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    torch.cuda.set_device(0)
    yield torch
    pkg.set_tuning(sweep=1, sweep_nbuf=1, sweep_unroll=8, sweep_remap=8, sweep_wlog=0, sweep_tile_rows=0, sweep_max_stage=0, tlc=1)


def _prep(pkg, coo, C, sigma, dtype, fixed=None, perm=None):
    s = pkg.convert_to_scs(coo, C, sigma, dtype, fixed_permutation=fixed)
    a = s.arrays()
    pkg.permute_scs_cols(s, a["old_to_new_idx"] if perm is None else perm)
    return s, s.arrays()


def _xp(pkg, s, a, special=False):
    xp = np.zeros(s.n_rows_padded, s.np_dtype)
    xp[:s.n_rows] = pkg.apply_permutation(make_x(s.n_rows, s.np_dtype), a["new_to_old_idx"])
    if special:
        xp[0] = -np.inf; xp[3] = -0.0
    return xp


@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_sweep_banded_random_bitexact(pkg, orc, torch_cuda, dt):
    t = torch_cuda
    coo = pkg.gen_banded_random(60000, 60, 5000)
    code = pkg.F64 if dt == "f64" else pkg.F32
    for C, sigma in ((32, 512), (16, 128), (64, 1)):
        s, a = _prep(pkg, coo, C, sigma, code)
        xp = _xp(pkg, s, a)
        y_or = orc.spmv_scs(C, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], xp)
        x = t.from_numpy(xp).cuda()
        A = pkg.DeviceMatrix(s)
        for wlog, rows in ((10, 256), (11, 512), (12, 1024), (12, 2048), (11, 4096), (0, 0)):
            if sigma > 1 and wlog and (1 << wlog) % sigma:
                continue
            nt, nsw = A.optimize_sweep(s, wlog, rows)
            assert nsw == nt and A.plan_info()[0] == 2, (C, sigma, wlog, rows, nt, nsw)
            for nbuf in (2, 1):
                for un in (8, 4):
                    pkg.set_tuning(sweep_nbuf=nbuf, sweep_unroll=un)
                    y = t.full((s.n_rows_padded,), -7.0, dtype=A.torch_dtype, device="cuda")
                    pkg.spmv(A, x, y)
                    assert np.array_equal(y.cpu().numpy(), y_or), (C, sigma, wlog, rows, nbuf, un)
        pkg.set_tuning(sweep_nbuf=1, sweep_unroll=8)


def test_sweep_ap_banded_random_bitexact(pkg, orc, torch_cuda):
    t = torch_cuda
    coo = pkg.gen_banded_random(50000, 70, 6000, magnitude_decades=10.0)
    dp, sp = pkg.partition_precisions(coo, 1e-3)
    assert dp.nnz > 0 and sp.nnz > 0
    for C, sigma in ((32, 512), (64, 64)):
        ds = pkg.convert_to_scs(dp, C, sigma, pkg.F64)
        perm = ds.arrays()["old_to_new_idx"].copy()
        ss = pkg.convert_to_scs(sp, C, sigma, pkg.F32, fixed_permutation=perm)
        pkg.permute_scs_cols(ds, perm); pkg.permute_scs_cols(ss, perm)
        da, sa = ds.arrays(), ss.arrays()
        for special in (False, True):
            xp = _xp(pkg, ds, da, special)
            y_or = orc.spmv_scs_ap_adv(C, ds.n_chunks, (da["chunk_ptrs"], da["chunk_lengths"], da["col_idxs"], da["values"]),
                                       (sa["chunk_ptrs"], sa["chunk_lengths"], sa["col_idxs"], sa["values"]), xp)
            x = t.from_numpy(xp).cuda()
            Ad, As = pkg.DeviceMatrix(ds), pkg.DeviceMatrix(ss)
            y0 = t.zeros(ds.n_rows_padded, dtype=t.float64, device="cuda")
            pkg.spmv_ap(Ad, As, x, y0)                                  # gather kernel
            for wlog, rows in ((11, 256), (12, 1024), (12, 2048), (0, 0)):
                nt, nsw = pkg.optimize_sweep_ap(Ad, As, ds, ss, wlog, rows)
                assert nsw == nt
                for nbuf in (2, 1):
                    pkg.set_tuning(sweep_nbuf=nbuf)
                    y = t.full((ds.n_rows_padded,), -7.0, dtype=t.float64, device="cuda")
                    pkg.spmv_ap(Ad, As, x, y)
                    got = y.cpu().numpy()
                    if special:   # NaN sign is the hardware's: compare NaN places and every other bit
                        assert np.array_equal(np.isnan(got), np.isnan(y_or)) and np.array_equal(got[~np.isnan(got)], y_or[~np.isnan(y_or)])
                        g0 = y0.cpu().numpy()
                        assert np.array_equal(np.isnan(got), np.isnan(g0)) and np.array_equal(got[~np.isnan(got)], g0[~np.isnan(g0)])
                    else:
                        assert np.array_equal(got, y_or), (C, sigma, wlog, rows, nbuf)
            pkg.set_tuning(sweep_nbuf=2)


@pytest.mark.parametrize("name", ["impcol_e", "bcsstk13", "matrix_band_klein", "FDM-2d-16", "myBigMat"])
def test_sweep_partial_coverage_on_golden_matrices(pkg, orc, torch_cuda, name):
    """Real matrices: some tiles qualify (column-sorted rows), some do not (symmetric files expanded out of order, more than
    255 entries of a row in one window): the sweep kernel and the gather kernel over the rest chunks together give the oracle's y."""
    t = torch_cuda
    m = pkg.read_mtx(mtx_path(name))
    pkg.set_tuning(sweep_max_stage=1 << 20)
    seen = 0
    for C, sigma, wlog in ((32, 512, 9), (32, 1, 8), (16, 64, 8), (64, 128, 10)):
        for code in (pkg.F64, pkg.F32):
            s, a = _prep(pkg, m, C, sigma, code)
            xp = _xp(pkg, s, a)
            y_or = orc.spmv_scs(C, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], xp)
            A = pkg.DeviceMatrix(s)
            nt, nsw = A.optimize_sweep(s, wlog, 256)
            seen += nsw
            y = t.full((s.n_rows_padded,), -7.0, dtype=A.torch_dtype, device="cuda")
            pkg.spmv(A, t.from_numpy(xp).cuda(), y)
            assert np.array_equal(y.cpu().numpy(), y_or), (name, C, sigma, code, nt, nsw)
    pkg.set_tuning(sweep_max_stage=0)
    if name in ("impcol_e", "matrix_band_klein"):
        assert seen > 0


def test_optimize_picks_the_sweep_for_wide_irregular_rows(pkg, orc, torch_cuda):
    t = torch_cuda
    coo = pkg.gen_banded_random(120000, 140, 50000)
    s, a = _prep(pkg, coo, 32, 512, pkg.F64)
    A = pkg.DeviceMatrix(s, tlc=True)                 # uspmv_dmat_optimize: TLC stages nothing here -> sweep plan
    kind, nt, nsw = A.plan_info()
    assert kind == 2 and nsw == nt
    xp = _xp(pkg, s, a)
    y = t.zeros(s.n_rows_padded, dtype=t.float64, device="cuda")
    pkg.spmv(A, t.from_numpy(xp).cuda(), y)
    assert np.array_equal(y.cpu().numpy(), orc.spmv_scs(32, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], xp))
    # a stencil keeps the tile-local-column plan
    s2, a2 = _prep(pkg, pkg.gen_stencil27(30, 30, 30), 32, 512, pkg.F64)
    A2 = pkg.DeviceMatrix(s2, tlc=True)
    assert A2.plan_info()[0] == 1


def test_device_sweep_plan_builder_equals_host_planner(pkg, orc, torch_cuda):
    """uspmv_dmat_optimize_sweep_device (csrc/sweep_plan_kernels.hip) against host/sweep_plan.cpp: every plan array bit for bit (digests
    of the device arrays), dp / sp / the ap[dp_sp] pair, full and partial coverage, several windows and tile heights; and y."""
    t = torch_cuda
    coo = pkg.gen_banded_random(60000, 60, 5000)
    for code in (pkg.F64, pkg.F32):
        for C, sigma in ((32, 512), (16, 128), (64, 1)):
            s, a = _prep(pkg, coo, C, sigma, code)
            xp = _xp(pkg, s, a)
            y_or = orc.spmv_scs(C, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], xp)
            for wlog, rows in ((10, 256), (12, 1024), (11, 4096), (0, 0)):
                if sigma > 1 and wlog and (1 << wlog) % sigma:
                    continue
                Ah, Ad = pkg.DeviceMatrix(s), pkg.DeviceMatrix(s)
                Ah.optimize_sweep(s, wlog, rows)
                nt, nsw = Ad.optimize_sweep_device(None, wlog, rows)
                dh, mh = Ah.sweep_plan_digest()
                dd, md = Ad.sweep_plan_digest()
                assert mh == md and mh[0] == 1 and nsw == mh[3] == nt, (C, sigma, wlog, rows, mh, md)
                assert dh == dd, (C, sigma, wlog, rows, [k for k in range(16) if dh[k] != dd[k]])
                y = t.full((s.n_rows_padded,), -7.0, dtype=Ad.torch_dtype, device="cuda")
                pkg.spmv(Ad, t.from_numpy(xp).cuda(), y)
                assert np.array_equal(y.cpu().numpy(), y_or)
    # partial coverage and stripped padding on real matrices (rest chunks, pad columns)
    pkg.set_tuning(sweep_max_stage=1 << 20)
    for name in ("impcol_e", "matrix_band_klein", "bcsstk13"):
        m = pkg.read_mtx(mtx_path(name))
        for C, sigma, wlog in ((32, 512, 9), (16, 64, 8)):
            s, a = _prep(pkg, m, C, sigma, pkg.F64)
            Ah, Ad = pkg.DeviceMatrix(s), pkg.DeviceMatrix(s)
            Ah.optimize_sweep(s, wlog, 256)
            Ad.optimize_sweep_device(None, wlog, 256)
            dh, mh = Ah.sweep_plan_digest()
            dd, md = Ad.sweep_plan_digest()
            assert mh == md and dh == dd, (name, C, sigma, mh, md, [k for k in range(16) if dh[k] != dd[k]])
    pkg.set_tuning(sweep_max_stage=0)
    # the ap[dp_sp] pair
    coo = pkg.gen_banded_random(50000, 70, 6000, magnitude_decades=10.0)
    dp, sp = pkg.partition_precisions(coo, 1e-3)
    ds = pkg.convert_to_scs(dp, 32, 512, pkg.F64)
    perm = ds.arrays()["old_to_new_idx"].copy()
    ss = pkg.convert_to_scs(sp, 32, 512, pkg.F32, fixed_permutation=perm)
    pkg.permute_scs_cols(ds, perm); pkg.permute_scs_cols(ss, perm)
    da, sa = ds.arrays(), ss.arrays()
    xp = _xp(pkg, ds, da)
    y_or = orc.spmv_scs_ap_adv(32, ds.n_chunks, (da["chunk_ptrs"], da["chunk_lengths"], da["col_idxs"], da["values"]),
                               (sa["chunk_ptrs"], sa["chunk_lengths"], sa["col_idxs"], sa["values"]), xp)
    for wlog, rows in ((12, 1024), (0, 0)):
        Ahd, Ahs, Add, Ads = pkg.DeviceMatrix(ds), pkg.DeviceMatrix(ss), pkg.DeviceMatrix(ds), pkg.DeviceMatrix(ss)
        pkg.optimize_sweep_ap(Ahd, Ahs, ds, ss, wlog, rows)
        Add.optimize_sweep_device(Ads, wlog, rows)
        dh, mh = Ahd.sweep_plan_digest()
        dd, md = Add.sweep_plan_digest()
        assert mh == md and dh == dd and mh[7] > 0, (wlog, rows, mh, md, [k for k in range(16) if dh[k] != dd[k]])
        y = t.full((ds.n_rows_padded,), -7.0, dtype=t.float64, device="cuda")
        pkg.spmv_ap(Add, Ads, t.from_numpy(xp).cuda(), y)
        assert np.array_equal(y.cpu().numpy(), y_or)


def test_optimize_device_reaches_the_sweep_without_a_host_struct(pkg, orc, torch_cuda):
    """uspmv_dmat_optimize_device[_ap] on wide irregular rows: the tile-local-column plan stages nothing, the sweep plan -- built on the device
    from the handle's arrays -- takes over (plan kind 2), as uspmv_dmat_optimize does with a host struct."""
    t = torch_cuda
    coo = pkg.gen_banded_random(120000, 140, 50000, magnitude_decades=10.0)
    s, a = _prep(pkg, coo, 32, 512, pkg.F64)
    A = pkg.DeviceMatrix(s)
    A.optimize_device()
    kind, nt, nsw = A.plan_info()
    assert kind == 2 and nsw == nt > 0
    xp = _xp(pkg, s, a)
    y = t.zeros(s.n_rows_padded, dtype=t.float64, device="cuda")
    pkg.spmv(A, t.from_numpy(xp).cuda(), y)
    assert np.array_equal(y.cpu().numpy(), orc.spmv_scs(32, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], xp))
    dp, sp = pkg.partition_precisions(coo, 1e-3)
    ds = pkg.convert_to_scs(dp, 32, 512, pkg.F64)
    perm = ds.arrays()["old_to_new_idx"].copy()
    ss = pkg.convert_to_scs(sp, 32, 512, pkg.F32, fixed_permutation=perm)
    pkg.permute_scs_cols(ds, perm); pkg.permute_scs_cols(ss, perm)
    da, sa = ds.arrays(), ss.arrays()
    Ad, As = pkg.DeviceMatrix(ds), pkg.DeviceMatrix(ss)
    pkg.optimize_device_ap(Ad, As)
    assert Ad.plan_info()[0] == 2
    xp = _xp(pkg, ds, da)
    y = t.zeros(ds.n_rows_padded, dtype=t.float64, device="cuda")
    pkg.spmv_ap(Ad, As, t.from_numpy(xp).cuda(), y)
    assert np.array_equal(y.cpu().numpy(), orc.spmv_scs_ap_adv(32, ds.n_chunks, (da["chunk_ptrs"], da["chunk_lengths"], da["col_idxs"], da["values"]),
                                                               (sa["chunk_ptrs"], sa["chunk_lengths"], sa["col_idxs"], sa["values"]), xp))
    # a stencil keeps the tile-local-column plan
    s2, _ = _prep(pkg, pkg.gen_stencil27(30, 30, 30), 32, 512, pkg.F64)
    A2 = pkg.DeviceMatrix(s2)
    A2.optimize_device()
    assert A2.plan_info()[0] == 1


def test_sweep_chain_variants_bitexact(pkg, orc, torch_cuda):
    """"sweep_pair" 0 (one chain at a time) / 1 (two chains of a lane side by side) / 2 (... with the FMAs under the rounds' lane masks
    instead of selects): same bits as the oracle for dp, sp and ap[dp_sp], special values included."""
    t = torch_cuda
    coo = pkg.gen_banded_random(50000, 70, 6000, magnitude_decades=10.0)
    try:
        for code in (pkg.F64, pkg.F32):
            s, a = _prep(pkg, coo, 32, 512, code)
            for special in (False, True):
                xp = _xp(pkg, s, a, special)
                y_or = orc.spmv_scs(32, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], xp)
                for rows in (1024, 2048, 4096):
                    A = pkg.DeviceMatrix(s)
                    A.optimize_sweep(s, 11, rows)
                    for pair in (0, 1, 2):
                        for un in (8, 4):
                            pkg.set_tuning(sweep_pair=pair, sweep_unroll=un)
                            y = t.full((s.n_rows_padded,), -7.0, dtype=A.torch_dtype, device="cuda")
                            pkg.spmv(A, t.from_numpy(xp).cuda(), y)
                            got = y.cpu().numpy()
                            assert np.array_equal(np.isnan(got), np.isnan(y_or)) and np.array_equal(got[~np.isnan(got)], y_or[~np.isnan(y_or)]), (code, special, rows, pair, un)
        dp, sp = pkg.partition_precisions(coo, 1e-3)
        ds = pkg.convert_to_scs(dp, 32, 512, pkg.F64)
        perm = ds.arrays()["old_to_new_idx"].copy()
        ss = pkg.convert_to_scs(sp, 32, 512, pkg.F32, fixed_permutation=perm)
        pkg.permute_scs_cols(ds, perm); pkg.permute_scs_cols(ss, perm)
        da, sa = ds.arrays(), ss.arrays()
        for special in (False, True):
            xp = _xp(pkg, ds, da, special)
            y_or = orc.spmv_scs_ap_adv(32, ds.n_chunks, (da["chunk_ptrs"], da["chunk_lengths"], da["col_idxs"], da["values"]),
                                       (sa["chunk_ptrs"], sa["chunk_lengths"], sa["col_idxs"], sa["values"]), xp)
            for rows in (1024, 4096):
                Ad, As = pkg.DeviceMatrix(ds), pkg.DeviceMatrix(ss)
                pkg.optimize_sweep_ap(Ad, As, ds, ss, 12, rows)
                for pair in (0, 1, 2):
                    pkg.set_tuning(sweep_pair=pair, sweep_unroll=8)
                    y = t.full((ds.n_rows_padded,), -7.0, dtype=t.float64, device="cuda")
                    pkg.spmv_ap(Ad, As, t.from_numpy(xp).cuda(), y)
                    got = y.cpu().numpy()
                    assert np.array_equal(np.isnan(got), np.isnan(y_or)) and np.array_equal(got[~np.isnan(got)], y_or[~np.isnan(y_or)]), (special, rows, pair)
    finally:
        pkg.set_tuning(sweep_pair=2, sweep_unroll=8)
