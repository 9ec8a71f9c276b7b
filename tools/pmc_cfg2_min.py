#!/usr/bin/env python3
"""Minimal workload for counter passes: config 2 (253^3 27-point stencil, SELL-32-512 dp), 6 launches of the default SpMV kernel (converted and planned on the device)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
torch.cuda.set_device(0)
pkg.set_tuning(tlc_measure_tile=0)
m = pkg.gen_stencil27(253, 253, 253)
I, J, V = m.arrays()
dI, dJ, dV = torch.from_numpy(np.array(I)).cuda(), torch.from_numpy(np.array(J)).cuda(), torch.from_numpy(np.array(V)).cuda()
lay, A, o2n, n2o = pkg.convert_to_scs_device_from_arrays(dI, dJ, dV, m.n_rows, m.n_cols, 32, 512, pkg.F64, sort=pkg.SORT_HOST, want_layout=False)
A.optimize_device()
x = torch.full((A.n_rows_padded,), 5.0, dtype=torch.float64, device="cuda"); y = torch.zeros_like(x)
for _ in range(6):
    pkg.spmv(A, x, y)
torch.cuda.synchronize()
