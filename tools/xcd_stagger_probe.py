#!/usr/bin/env python3
"""XCD group size matched to the matrix's plane stride (tools/xcd_group_probe.py: fewer bytes from the fabric, PMC-confirmed, but no faster)
together with a stagger of the XCDs inside their groups (remap_block): do the eight XCDs, no longer in step, turn the saved bytes into time?
Usage: xcd_stagger_probe.py <2|3>"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
from ultimate_spmv_amd import binding as B
torch.cuda.set_device(0)
pkg.set_tuning(tlc_measure_tile=0)
cfg = sys.argv[1] if len(sys.argv) > 1 else "3"
if cfg == "3":
    g = 111
    coo = pkg.gen_stencil27(g, g, g, dof=3)
    s = pkg.convert_to_scs(coo, 32, 512, pkg.F64); a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"])
    b, ld = 8, s.n_rows_padded
    X = torch.rand(b * ld, dtype=torch.float64, device="cuda"); Y = torch.zeros_like(X)
    A = pkg.DeviceMatrix(s, block_tlc=b)
    run = lambda n: B.time_launches(5, n, A=A, x=X, y=Y, b=b, ld=ld, layout=pkg.ROWWISE)
    grid = [(256, 0), (256, 32), (72, 0), (72, 9), (72, 4), (72, 18), (72, 27), (144, 18), (144, 9), (36, 4), (36, 9), (289, 36), (578, 72), (578, 0)]
else:
    g = 253
    coo = pkg.gen_stencil27(g, g, g)
    s = pkg.convert_to_scs(coo, 32, 512, pkg.F64); a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"])
    A = pkg.DeviceMatrix(s, tlc=True)
    x = torch.full((s.n_rows_padded,), 5.0, dtype=torch.float64, device="cuda"); y = torch.zeros_like(x)
    run = lambda n: B.time_launches(0, n, A=A, x=x, y=y)
    grid = [(256, 0), (256, 32), (31, 0), (31, 4), (31, 8), (62, 8), (125, 16), (250, 31), (250, 0), (125, 0)]
del coo
res = {}
for rep in range(2):
    for G, S in grid:
        pkg.set_tuning(xcd_remap=G, xcd_stagger=S)
        run(5)
        res.setdefault((G, S), []).append(min(run(30) for _ in range(3)))
for (G, S), v in res.items():
    print(json.dumps({"config": cfg, "xcd_remap": G, "xcd_stagger": S, "ms": round(min(v), 4), "both": [round(t, 4) for t in v]}), flush=True)
