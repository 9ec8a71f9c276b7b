"""Device-memory hygiene: handles with every kind of plan, adaptive-precision pairs and the distributed object (with its optional step
forms and block plans) are created, used and freed repeatedly; the GPU's free memory after five more cycles is what it was after the
first (hipMemGetInfo through torch; the first cycle absorbs one-time runtime allocations)."""
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cycle_handles(pkg, t, m):
    s = pkg.convert_to_scs(m, 32, 512, pkg.F64)
    a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"])
    x = t.ones(s.n_rows_padded, dtype=t.float64, device="cuda"); y = t.zeros_like(x)
    for setup in ("host", "device", "sweep", "sweep_device"):
        A = pkg.DeviceMatrix(s)
        if setup == "host": A.optimize(s)
        elif setup == "device": A.optimize_device()
        elif setup == "sweep": A.optimize_sweep(s)
        else: A.optimize_sweep_device()
        pkg.spmv(A, x, y)
        del A
    for b in (4, 8):
        X = t.ones(b * s.n_rows_padded, dtype=t.float64, device="cuda"); Y = t.zeros_like(X)
        for dev in (False, True):
            A = pkg.DeviceMatrix(s)
            A.optimize_block_device(b) if dev else A.optimize_block(s, b)
            for lay in (pkg.COLWISE, pkg.ROWWISE):
                pkg.spmmv(A, X, Y, b, s.n_rows_padded, lay)
            del A
        del X, Y
    lay, Ad = pkg.convert_to_scs_device(m, 32, 512, pkg.F64)
    Ad.optimize_device(); pkg.spmv(Ad, x, y)
    del Ad, lay
    # round 4: conversion from device arrays (both orderings), block-vector window sweep, prepared X
    I_, J_, V_ = m.arrays()
    dI, dJ, dV = (t.from_numpy(np.array(v)).cuda() for v in (I_, J_, V_))
    for mode in (pkg.SORT_HOST, pkg.SORT_DEVICE_STABLE):
        lay2, A2, o2n, n2o = pkg.convert_to_scs_device_from_arrays(dI, dJ, dV, m.n_rows, m.n_cols, 32, 512, pkg.F64, sort=mode, want_layout=mode == pkg.SORT_HOST)
        A2.optimize_device(); pkg.spmv(A2, x, y)
        del A2, lay2, o2n, n2o
    del dI, dJ, dV
    X = t.ones(8 * s.n_rows_padded, dtype=t.float64, device="cuda"); Y = t.zeros_like(X)
    Aw = pkg.DeviceMatrix(s)
    Aw.optimize_block_sweep(s, 8, wlog=9, tile_rows=1024)
    Ab = pkg.DeviceMatrix(s); Ab.optimize_block(s, 8)
    for A in (Aw, Ab):
        pkg.spmmv_x_prepared(A, X, 8, s.n_rows_padded)
        for lay_ in (pkg.COLWISE, pkg.ROWWISE):
            pkg.spmmv(A, X, Y, 8, s.n_rows_padded, lay_)
        pkg.spmmv_x_release(A)
    del Aw, Ab, A, X, Y
    dp, sp = pkg.partition_precisions(m, 1e-1)
    ds = pkg.convert_to_scs(dp, 32, 512, pkg.F64)
    perm = ds.arrays()["old_to_new_idx"].copy()
    ss = pkg.convert_to_scs(sp, 32, 512, pkg.F32, fixed_permutation=perm)
    pkg.permute_scs_cols(ds, perm); pkg.permute_scs_cols(ss, perm)
    for how in ("host", "device"):
        Pd, Ps = pkg.DeviceMatrix(ds), pkg.DeviceMatrix(ss)
        pkg.optimize_ap(Pd, Ps, ds, ss) if how == "host" else pkg.optimize_device_ap(Pd, Ps)
        pkg.spmv_ap(Pd, Ps, x[:ds.n_rows_padded].contiguous(), y[:ds.n_rows_padded].contiguous())
        del Pd, Ps
    del x, y
    gc.collect()
    t.cuda.synchronize()
    t.cuda.empty_cache()


def _cycle_dist(pkg, t, mb, wsa):
    # the distributed object in loopback: every step form, a captured graph, the block plan and a two-part block step
    d = pkg.DistNative(mb, wsa, 32, 512, 1, 2, pkg.comm_unique_id(), comm_rank=0, comm_size=1)
    xd = d.new_x(np.full(d.n_local, 5.0)); yd = d.new_y()
    for pad, fused in ((0, 0), (1, 0), (1, 1)):
        d.set_option("pad_split", pad); d.set_option("fused_step", fused)
        d.run(xd, yd, 3, use_graph=False)
    d.set_option("pad_split", 0); d.set_option("fused_step", 0)
    d.run(xd, yd, 3, use_graph=True)
    d.set_option("block_plan", 8)
    Xd = t.ones(8 * d.padded_vec_size, dtype=t.float64, device="cuda"); Yd = t.zeros_like(Xd)
    for lay_ in (pkg.COLWISE, pkg.ROWWISE):
        d.spmmv(Xd, Yd, 8, lay_, 0)
    d.synchronize()
    d.close()
    del d, xd, yd, Xd, Yd
    gc.collect()
    t.cuda.synchronize()
    t.cuda.empty_cache()


def test_no_device_memory_is_lost_over_create_use_free_cycles(pkg):
    import torch as t
    t.cuda.set_device(0)
    m = pkg.gen_stencil27(40, 40, 40)
    _cycle_handles(pkg, t, m)
    free0, _ = t.cuda.mem_get_info()
    for _ in range(5):
        _cycle_handles(pkg, t, m)
    free1, _ = t.cuda.mem_get_info()
    assert free0 - free1 <= 2 << 20, f"handles: {(free0 - free1) / 2**20:.1f} MiB of device memory lost over five cycles"
    # The distributed object creates and destroys an RCCL communicator per cycle.  RCCL keeps the p2p channel buffers of the first few
    # communicators in a pool of its own (133 MiB each until the pool saturates) and ~1-2 MiB per communicator for good
    # (ncclCommInitRank / ncclCommDestroy alone show that); the object's own buffers must not add to it.
    shape = (24, 24, 48)
    counts = pkg.gen_stencil27_row_counts(*shape)
    wsa = pkg.seg_from_row_counts(counts, "seg-rows", 2)
    mb = pkg.gen_stencil27(*shape, row_begin=int(wsa[1]), row_end=int(wsa[2]))
    for _ in range(6):
        _cycle_dist(pkg, t, mb, wsa)
    free0, _ = t.cuda.mem_get_info()
    for _ in range(5):
        _cycle_dist(pkg, t, mb, wsa)
    free1, _ = t.cuda.mem_get_info()
    assert free0 - free1 <= 5 * (4 << 20), f"distributed object: {(free0 - free1) / 2**20:.1f} MiB of device memory lost over five cycles"
