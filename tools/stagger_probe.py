#!/usr/bin/env python3
"""xcd_stagger against the y-placement lottery (profiles/r04/placement_*.txt): one matrix, several raw buffers (each a fresh physical
allocation, all held), y at a 2 MiB boundary and 1 MiB behind it, XCD stagger S in {0, 8, 29, 32, 64}.  Usage: stagger_probe.py [2|3]"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
from ultimate_spmv_amd import binding as B
torch.cuda.set_device(0)
pkg.set_tuning(tlc_measure_tile=0)
cfg = sys.argv[1] if len(sys.argv) > 1 else "2"
MB = 1 << 20
if cfg == "3":
    g = 111
    coo = pkg.gen_stencil27(g, g, g, dof=3)
    s = pkg.convert_to_scs(coo, 32, 512, pkg.F64); a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"])
    b, ld = 8, s.n_rows_padded
    A = pkg.DeviceMatrix(s, block_tlc=b)
    nb = b * ld * 8
    run = lambda x, y, n: B.time_launches(5, n, A=A, x=x, y=y, b=b, ld=ld, layout=pkg.ROWWISE)
else:
    g = 253
    coo = pkg.gen_stencil27(g, g, g)
    s = pkg.convert_to_scs(coo, 32, 512, pkg.F64); a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"])
    A = pkg.DeviceMatrix(s, tlc=True)
    nb = s.n_rows_padded * 8
    run = lambda x, y, n: B.time_launches(0, n, A=A, x=x, y=y)
del coo
span = ((nb + 2 * MB - 1) // (2 * MB) + 1) * 2 * MB


def timeit(x, y):
    run(x, y, 10)
    return round(min(run(x, y, 30) for _ in range(3)), 4)


bufs = [torch.zeros(2 * span + 2 * MB, dtype=torch.uint8, device="cuda") for _ in range(4)]
for i, R in enumerate(bufs):
    x = R[0:nb].view(torch.float64); x.fill_(1.5)
    row = {"config": cfg, "buffer": i}
    for S in (0, 8, 29, 32, 64):
        pkg.set_tuning(xcd_stagger=S)
        row[f"S={S}"] = [timeit(x, R[span:span + nb].view(torch.float64)), timeit(x, R[span + MB:span + MB + nb].view(torch.float64))]
    pkg.set_tuning(xcd_stagger=0)
    print(json.dumps(row), flush=True)
