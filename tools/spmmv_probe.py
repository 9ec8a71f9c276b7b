#!/usr/bin/env python3
"""SpMMV probe on the Queen_4147-class matrix: unroll / block / xcd variants, both layouts, b = 8 (and sp)."""
import os, sys, json, itertools
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
from ultimate_spmv_amd import binding as B
torch.cuda.set_device(0)
pkg.set_tuning(spmmv_list_plan=1)      # these probes time the older block-plan kernels too
g = int(sys.argv[1]) if len(sys.argv) > 1 else 111
sig = int(sys.argv[2]) if len(sys.argv) > 2 else 512
coo = pkg.gen_stencil27(g, g, g, dof=3)
for dt, tdt, vs in ((pkg.F64, torch.float64, 8), (pkg.F32, torch.float32, 4)):
    s = pkg.convert_to_scs(coo, 32, sig, dt); a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"])
    A = pkg.DeviceMatrix(s)
    b, ld = 8, s.n_rows_padded
    X = torch.rand(b * ld, dtype=tdt, device="cuda"); Y = torch.zeros_like(X)
    byts = s.n_elements * (vs + 4) + 8 * s.n_chunks + 2 * b * vs * ld
    for lay, nm in ((pkg.ROWWISE, "rowwise"), (pkg.COLWISE, "colwise")):
        for var, u, pf, blk in ((0, 0, 1, 256), (3, 4, 1, 256), (3, 8, 1, 256), (3, 8, 1, 128), (3, 8, 1, 64)):
            pkg.set_tuning(spmmv_variant=var, spmmv_unroll=u, spmmv_prefetch=pf, block=blk)
            Yr = torch.zeros_like(X)
            pkg.spmmv(A, X, Yr, b, ld, lay)
            if var == 0:
                Yref = Yr.clone()
            B.time_launches(5, 3, A=A, x=X, y=Y, b=b, ld=ld, layout=lay)
            ms = B.time_launches(5, 30, A=A, x=X, y=Y, b=b, ld=ld, layout=lay)
            print(json.dumps(dict(dtype=vs, sigma=sig, layout=nm, variant=var, unroll=u, prefetch=pf, block=blk, same=bool(torch.equal(Yr, Yref)), ms=round(ms, 4), GF=round(2 * s.nnz * b / ms / 1e6), GBs=round(byts / ms / 1e6))), flush=True)
    pkg.set_tuning(spmmv_prefetch=0)
    pkg.set_tuning(spmmv_variant=0, spmmv_unroll=0, block=256, xcd_remap=256)
    # LDS-staged block plan (uspmv_dmat_optimize_block)
    import time
    t0 = time.time()
    Ab = pkg.DeviceMatrix(s, block_tlc=b)
    print(json.dumps(dict(dtype=vs, block_plan_tiles=Ab.block_tiles, staged=Ab.block_staged, plan_s=round(time.time() - t0, 2))), flush=True)
    for lay, nm in ((pkg.ROWWISE, "rowwise"), (pkg.COLWISE, "colwise")):
        for batch, xr in ((1, 256), (0, 256), (0, 1024)):
            pkg.set_tuning(spmmv_swizzle=batch, xcd_remap=xr)
            Y2 = torch.zeros_like(X)
            pkg.set_tuning(spmmv_variant=4); pkg.spmmv(Ab, X, Y2, b, ld, lay); pkg.set_tuning(spmmv_variant=3); pkg.spmmv(A, X, Y, b, ld, lay); pkg.set_tuning(spmmv_variant=0)
            same = bool(torch.equal(Y, Y2))
            pkg.set_tuning(spmmv_variant=4)
            B.time_launches(5, 3, A=Ab, x=X, y=Y, b=b, ld=ld, layout=lay)
            ms = B.time_launches(5, 30, A=Ab, x=X, y=Y, b=b, ld=ld, layout=lay)
            print(json.dumps(dict(dtype=vs, layout=nm, variant="block_plan", swizzle=batch, xcd=xr, bitexact_vs_gather=same, ms=round(ms, 4), GF=round(2 * s.nnz * b / ms / 1e6), GBs=round(byts / ms / 1e6))), flush=True)
    pkg.set_tuning(spmmv_swizzle=1, xcd_remap=256, spmmv_variant=0)
    del Ab
    # single-vector SpMV in this dtype for reference (TLC)
    A.optimize(s)
    x = torch.rand(ld, dtype=tdt, device="cuda"); y = torch.zeros_like(x)
    ms = B.time_launches(0, 30, A=A, x=x, y=y)
    print(json.dumps(dict(dtype=vs, spmv_tlc_ms=round(ms, 4), GF=round(2 * s.nnz / ms / 1e6), GBs=round((s.n_elements * (vs + 4) + 8 * s.n_chunks + 2 * vs * ld) / ms / 1e6))), flush=True)
