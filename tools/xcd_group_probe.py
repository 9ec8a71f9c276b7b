#!/usr/bin/env python3
"""Tile -> XCD group size against the matrix's far stride.  `xcd_remap` G >= 2 lets XCD k process G consecutive tiles of every super-block
of 8 G tiles.  When 8 G equals the distance (in tiles) at which the matrix re-uses x -- one grid plane of a 3-D stencil -- XCD k meets in
super-block j+1 the x lines it fetched in super-block j: the plane-to-plane re-use is served by the XCD's own L2 instead of the fabric.
Usage: xcd_group_probe.py <2|3|5> [G ...]   (2: 253^3 SpMV, 3: 111^3 x 3 dof SpMMV b = 8 row-wise, 5: 304^3 SpMV on one GPU)"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
from ultimate_spmv_amd import binding as B
torch.cuda.set_device(0)
cfg = sys.argv[1] if len(sys.argv) > 1 else "2"
Gs = [int(v) for v in sys.argv[2:]]
if cfg == "3":
    g = 111
    coo = pkg.gen_stencil27(g, g, g, dof=3)
    s = pkg.convert_to_scs(coo, 32, 512, pkg.F64); a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"])
    b, ld = 8, s.n_rows_padded
    X = torch.rand(b * ld, dtype=torch.float64, device="cuda"); Y = torch.zeros_like(X)
    A = pkg.DeviceMatrix(s, block_tlc=b)
    byts = s.n_elements * 12 + 8 * s.n_chunks + b * 8 * s.n_rows + b * 8 * s.n_rows_padded
    plane_tiles = g * g * 3 / 64.0
    run = lambda n: B.time_launches(5, n, A=A, x=X, y=Y, b=b, ld=ld, layout=pkg.ROWWISE)
    Gs = Gs or [32, 48, 60, 66, 70, 72, 73, 74, 78, 84, 96, 128, 144, 256]
else:
    g = 253 if cfg == "2" else 304
    coo = pkg.gen_stencil27(g, g, g)
    s = pkg.convert_to_scs(coo, 32, 512, pkg.F64); a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"])
    if cfg == "5":
        pkg.set_tuning(tlc_measure_tile=0)
    A = pkg.DeviceMatrix(s, tlc=True)
    x = torch.full((s.n_rows_padded,), 5.0, dtype=torch.float64, device="cuda"); y = torch.zeros_like(x)
    byts = s.n_elements * 12 + 8 * s.n_chunks + 8 * s.n_rows + 8 * s.n_rows_padded
    plane_tiles = g * g / 256.0
    run = lambda n: B.time_launches(0, n, A=A, x=x, y=y)
    Gs = Gs or ([8, 16, 24, 28, 30, 31, 32, 33, 36, 40, 48, 62, 64, 125, 128, 256] if cfg == "2" else [16, 32, 40, 44, 45, 46, 48, 64, 90, 128, 256])
del coo
print(json.dumps(dict(config=cfg, plane_stride_tiles=round(plane_tiles, 2), plane_over_8=round(plane_tiles / 8, 2), plan=A.plan_info())), flush=True)
res = {}
for rep in range(2):
    for G in Gs + [256]:
        pkg.set_tuning(xcd_remap=G)
        run(5)
        ms = min(run(30) for _ in range(3))
        res.setdefault(G, []).append(ms)
for G in sorted(res):
    ms = min(res[G])
    print(json.dumps(dict(xcd_remap=G, ms=round(ms, 4), both=[round(v, 4) for v in res[G]], frac=round(byts / ms / 1e6 / 8000, 4))), flush=True)
pkg.set_tuning(xcd_remap=256)
