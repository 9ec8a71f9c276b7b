#!/usr/bin/env python3
"""Minimal workload for counter passes: config 3 (111^3 x 3 dof, SELL-32-512 dp, b = 8), 6 row-wise launches of the default SpMMV kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
torch.cuda.set_device(0)
g = 111
coo = pkg.gen_stencil27(g, g, g, dof=3)
s = pkg.convert_to_scs(coo, 32, 512, pkg.F64); a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"])
b, ld = 8, s.n_rows_padded
X = torch.rand(b * ld, dtype=torch.float64, device="cuda"); Y = torch.zeros_like(X)
A = pkg.DeviceMatrix(s, block_tlc=b)
for _ in range(6):
    pkg.spmmv(A, X, Y, b, ld, pkg.ROWWISE)
torch.cuda.synchronize()
