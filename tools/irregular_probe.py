#!/usr/bin/env python3
"""Gather-kernel tuning on the irregular (banded-random, HV15R-class) matrix of SURVEY.md 8(d): block size, XCD
remap, unroll, and the tile-local-column kernel's wide-footprint path (which runs at lower occupancy)."""
import itertools, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
from ultimate_spmv_amd import binding as B
torch.cuda.set_device(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2017169
band = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
coo = pkg.gen_banded_random(n, 140, band)
s = pkg.convert_to_scs(coo, 32, 512, pkg.F64); pkg.permute_scs_cols(s, s.arrays()["old_to_new_idx"])
A = pkg.DeviceMatrix(s)
x = torch.rand(s.n_rows_padded, dtype=torch.float64, device="cuda"); y = torch.zeros_like(x)
byts = s.n_elements * 12 + 8 * s.n_chunks + 8 * s.n_rows + 8 * s.n_rows_padded
def t(Ah):
    B.time_launches(0, 3, A=Ah, x=x, y=y)
    return B.time_launches(0, 20, A=Ah, x=x, y=y)
for blk, xr, u in itertools.product((128, 256, 512, 1024), (0, 1, 64, 256, 2048), (4, 8)):
    pkg.set_tuning(block=blk, xcd_remap=xr, unroll=u)
    ms = t(A)
    print(json.dumps(dict(kernel="rows", block=blk, xcd=xr, unroll=u, ms=round(ms, 4), GBs=round(byts / ms / 1e6))), flush=True)
pkg.set_tuning(block=256, xcd_remap=256, unroll=8)
for ml in (0, 64, 1280):
    A2 = pkg.DeviceMatrix(s); nt, ns = A2.optimize(s, ml)
    for xr in (0, 256):
        pkg.set_tuning(xcd_remap=xr)
        print(json.dumps(dict(kernel="tlc", max_lines=ml, staged=[ns, nt], xcd=xr, ms=round(t(A2), 4))), flush=True)
pkg.set_tuning(xcd_remap=256)
