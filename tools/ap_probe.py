#!/usr/bin/env python3
"""AP kernel probe: time the dp+sp kernel for several thresholds / unrolls on the HV15R-class matrix."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
from ultimate_spmv_amd import binding as B
torch.cuda.set_device(0)
g = 74
coo = pkg.gen_stencil27(g, g, g, dof=5, magnitude_decades=10.0)
s = pkg.convert_to_scs(coo, 32, 512); a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"])
A = pkg.DeviceMatrix(s)
x = torch.full((s.n_rows_padded,), 5.0, dtype=torch.float64, device="cuda"); y = torch.zeros_like(x)
for u in (4, 8):
    pkg.set_tuning(unroll=u)
    print("plain dp unroll", u, round(B.time_launches(0, 50, A=A, x=x, y=y), 4), "ms  n_el", s.n_elements, flush=True)
for th in (0.0, 1e-3, 1e3):
    dp, sp = pkg.partition_precisions(coo, th)
    ds = pkg.convert_to_scs(dp, 32, 512) if dp.nnz else None
    if ds is None:
        continue
    perm = ds.arrays()["old_to_new_idx"].copy()
    ss = pkg.convert_to_scs(sp, 32, 512, pkg.F32, fixed_permutation=perm)
    pkg.permute_scs_cols(ds, perm); pkg.permute_scs_cols(ss, perm)
    Ad, As = pkg.DeviceMatrix(ds), pkg.DeviceMatrix(ss)
    for u in (2, 4, 8):
        pkg.set_tuning(unroll=u)
        ms = B.time_launches(4, 50, A=Ad, B=As, x=x, y=y)
        byts = 12 * ds.n_elements + 8 * ss.n_elements + 16 * ds.n_chunks + 16 * ds.n_rows_padded
        print(json.dumps(dict(th=th, unroll=u, ms=round(ms, 4), dp_el=ds.n_elements, sp_el=ss.n_elements, GBs=round(byts / ms / 1e6))), flush=True)
