#!/usr/bin/env python3
"""Config 3, phased SpMMV kernel (variant 8): tile -> XCD mapping (`xcd_remap`: 0 round-robin, 1 one contiguous eighth per XCD,
G > 1 groups of G consecutive tiles per XCD)."""
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
pkg.set_tuning(spmmv_variant=8)
for xr in (0, 1, 8, 32, 64, 128, 256, 512, 1024, 2048, 4096):
    pkg.set_tuning(xcd_remap=xr)
    B.time_launches(5, 5, A=A, x=X, y=Y, b=b, ld=ld, layout=pkg.ROWWISE)
    ms = min(B.time_launches(5, 40, A=A, x=X, y=Y, b=b, ld=ld, layout=pkg.ROWWISE) for _ in range(3))
    print(json.dumps(dict(xcd_remap=xr, ms=round(ms, 4))), flush=True)
pkg.set_tuning(xcd_remap=256, spmmv_variant=0)
