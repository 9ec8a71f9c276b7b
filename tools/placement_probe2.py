#!/usr/bin/env python3
"""Follow-up of placement_probe.py (the x / y vectors, not the matrix, decided the kernel's time): x and y carved out of raw buffers at
CONTROLLED offsets.  One matrix (253^3 stencil, SELL-32-512, tile-local-column plan), kernel ms by HIP events (best of 3 x 30)."""
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
pkg.set_tuning(tlc_measure_tile=0)
g = int(sys.argv[1]) if len(sys.argv) > 1 else 253
coo = pkg.gen_stencil27(g, g, g)
s = pkg.convert_to_scs(coo, 32, 512, pkg.F64); a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"])
del coo
A = pkg.DeviceMatrix(s, tlc=True)
n = s.n_rows_padded
nb = n * 8


def timeit(x, y):
    B.time_launches(0, 10, A=A, x=x, y=y)
    return round(min(B.time_launches(0, 30, A=A, x=x, y=y) for _ in range(3)), 4)


def carve(R, off):
    v = R[off:off + nb].view(torch.float64)
    return v


MB = 1 << 20
bufs = [torch.zeros(768 * MB, dtype=torch.uint8, device="cuda") for _ in range(3)]
for i, R in enumerate(bufs):
    assert R.data_ptr() % (2 * MB) == 0
    x = carve(R, 0); x.fill_(5.0)
    y = carve(R, 256 * MB)
    print(json.dumps({"test": "three raw buffers, x at +0, y at +256 MiB", "buffer": i, "base": hex(R.data_ptr()), "ms": timeit(x, y)}), flush=True)
R = bufs[0]
x = carve(R, 0); x.fill_(5.0)
for oy in (0, 0x80, 0x100, 0x200, 0x400, 0x800, 0x1000, 0x2000, 0x8000, 0x10000, 0x100000, 0x800 + 0x100000):
    y = carve(R, 256 * MB + oy)
    print(json.dumps({"test": "y offset from a 2 MiB boundary (x aligned)", "y_off": hex(oy), "ms": timeit(x, y)}), flush=True)
y = carve(R, 256 * MB)
for ox in (0, 0x80, 0x800, 0x1000, 0x10000, 0x100000):
    x = carve(R, ox); x.fill_(5.0)
    print(json.dumps({"test": "x offset from a 2 MiB boundary (y aligned)", "x_off": hex(ox), "ms": timeit(x, y)}), flush=True)
x = carve(R, 0); x.fill_(5.0)
for d in (nb, (nb + 0xfff) & ~0xfff, (nb + 2 * MB - 1) & ~(2 * MB - 1), 128 * MB, 128 * MB + 0x800, 192 * MB, 256 * MB, 384 * MB, 512 * MB):
    y = carve(R, d)
    print(json.dumps({"test": "distance y - x (x at +0)", "distance": hex(d), "ms": timeit(x, y)}), flush=True)
# x and y in DIFFERENT raw buffers
x = carve(bufs[1], 0); x.fill_(5.0)
for oy in (0, 0x800):
    y = carve(bufs[2], oy)
    print(json.dumps({"test": "x and y in different raw buffers", "y_off": hex(oy), "ms": timeit(x, y)}), flush=True)
