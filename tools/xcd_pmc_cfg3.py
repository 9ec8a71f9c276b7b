#!/usr/bin/env python3
"""config 3 (phased SpMMV, row-wise) at xcd_remap 256 and 72 (= plane stride / 8), alternating, for rocprofv3 counter passes:
does the matched group size cut the X re-fetch from the fabric, and does the kernel care?  (summary: tools/placement_pmc_summary.py
reads 'scs_spmv_tlc' dispatches; here the kernel is scs_spmmv_quadph -- pass its name as argv[2] of the summary script)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
from ultimate_spmv_amd import binding as B
torch.cuda.set_device(0)
g = 111
coo = pkg.gen_stencil27(g, g, g, dof=3)
s = pkg.convert_to_scs(coo, 32, 512, pkg.F64); a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"])
b, ld = 8, s.n_rows_padded
X = torch.rand(b * ld, dtype=torch.float64, device="cuda"); Y = torch.zeros_like(X)
A = pkg.DeviceMatrix(s, block_tlc=b)
del coo
for G in (256, 72):
    pkg.set_tuning(xcd_remap=G)
    B.time_launches(5, 10, A=A, x=X, y=Y, b=b, ld=ld, layout=pkg.ROWWISE)
    print(json.dumps({"xcd_remap": G, "ms": round(min(B.time_launches(5, 30, A=A, x=X, y=Y, b=b, ld=ld, layout=pkg.ROWWISE) for _ in range(3)), 4)}), flush=True)
torch.cuda.synchronize()
for i in range(8):
    pkg.set_tuning(xcd_remap=256 if i % 2 == 0 else 72)
    pkg.spmmv(A, X, Y, b, ld, pkg.ROWWISE)
    torch.cuda.synchronize()
