#!/usr/bin/env python3
"""Config 3, phased SpMMV kernel: the per-group compare + branch form against the one-switch form (measurement-only ablate 32).
Both must give the same bits; alternating timings on one box."""
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
X = torch.rand(b * ld, dtype=torch.float64, device="cuda")
A = pkg.DeviceMatrix(s, block_tlc=b)
pkg.set_tuning(spmmv_variant=8)
ys = {}
for abl in (0, 32):
    pkg.set_tuning(ablate=abl)
    Y = torch.zeros_like(X); pkg.spmmv(A, X, Y, b, ld, pkg.ROWWISE); torch.cuda.synchronize(); ys[abl] = Y
print(json.dumps(dict(same_bits=bool(torch.equal(ys[0], ys[32])))), flush=True)
for rep in range(3):
    for abl in (0, 32):
        pkg.set_tuning(ablate=abl)
        B.time_launches(5, 5, A=A, x=X, y=Y, b=b, ld=ld, layout=pkg.ROWWISE)
        ms = min(B.time_launches(5, 40, A=A, x=X, y=Y, b=b, ld=ld, layout=pkg.ROWWISE) for _ in range(3))
        print(json.dumps(dict(ablate=abl, what="one switch per phase" if abl else "compare + branch per group (default)", ms=round(ms, 4))), flush=True)
pkg.set_tuning(ablate=0, spmmv_variant=0)
