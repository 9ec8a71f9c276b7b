#!/usr/bin/env python3
"""Config 3, phased SpMMV kernel (variant 8), measurement-only ablations (bits: 1 no X staging, 2 no arithmetic, 4 no value loads,
8 no index loads, 16 no list).  Results of the ablated launches are wrong by construction; only their times are printed."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
from ultimate_spmv_amd import binding as B
torch.cuda.set_device(0)
g = int(sys.argv[1]) if len(sys.argv) > 1 else 111
coo = pkg.gen_stencil27(g, g, g, dof=3)
s = pkg.convert_to_scs(coo, 32, 512, pkg.F64); a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"])
b, ld = 8, s.n_rows_padded
X = torch.rand(b * ld, dtype=torch.float64, device="cuda"); Y = torch.zeros_like(X)
A = pkg.DeviceMatrix(s, block_tlc=b)
names = {0: "full kernel", 1: "no X staging", 2: "no arithmetic", 4: "no value loads", 8: "no index loads", 17: "no list, no X staging",
         14: "list + X staging only", 3: "list + matrix stream only", 19: "matrix stream only (no list)",
         64: "X staging of CONSECUTIVE rows (same count, whole lines)", 78: "list + staging of consecutive rows only"}
for var, abls in ((8, (0, 1, 2, 4, 8, 17, 14, 3, 19, 64, 78)),):
    pkg.set_tuning(spmmv_variant=var)
    for abl in abls:
        pkg.set_tuning(ablate=abl)
        B.time_launches(5, 5, A=A, x=X, y=Y, b=b, ld=ld, layout=pkg.ROWWISE)
        ms = min(B.time_launches(5, 40, A=A, x=X, y=Y, b=b, ld=ld, layout=pkg.ROWWISE) for _ in range(3))
        print(json.dumps(dict(variant=var, ablate=abl, what=names[abl], ms=round(ms, 4))), flush=True)
pkg.set_tuning(ablate=0, spmmv_variant=0)
