#!/usr/bin/env python3
"""Config 3 (Queen_4147-class, dp b = 8): the block-vector window sweep against the phased plan, both layouts; windows / tile rows swept.
Prints the plan's staging (X rows per matrix row) through USPMV_VERBOSE."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
from ultimate_spmv_amd import binding as B
torch.cuda.set_device(0)
g = int(sys.argv[1]) if len(sys.argv) > 1 else 111
coo = pkg.gen_stencil27(g, g, g, dof=3)
s = pkg.convert_to_scs(coo, 32, 512, pkg.F64); a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"])
del coo
b, ld = 8, s.n_rows_padded
byts = s.n_elements * 12 + 8 * s.n_chunks + b * 8 * s.n_rows + b * 8 * s.n_rows_padded
X = torch.rand(b * ld, dtype=torch.float64, device="cuda"); Y = torch.zeros_like(X)
A0 = pkg.DeviceMatrix(s, block_tlc=b)


def run(A, lay, n):
    return B.time_launches(5, n, A=A, x=X, y=Y, b=b, ld=ld, layout=lay)


for lay, nm in ((pkg.ROWWISE, "rowwise"), (pkg.COLWISE, "colwise")):
    run(A0, lay, 20)
    ms = min(run(A0, lay, 30) for _ in range(3))
    print(json.dumps({"plan": "phased", "layout": nm, "ms": round(ms, 4), "frac": round(byts / ms / 1e6 / 8000, 4)}), flush=True)
Yref = {}
for lay in (pkg.ROWWISE, pkg.COLWISE):
    Y.zero_(); pkg.spmmv(A0, X, Y, b, ld, lay); torch.cuda.synchronize(); Yref[lay] = Y.clone()
del A0
for wlog, tr, nbuf in ((9, 2048, 2), (10, 2048, 2), (10, 4096, 2), (10, 4096, 1), (11, 4096, 1), (11, 2048, 1)):
    pkg.set_tuning(sweep_nbuf=nbuf)
    A = pkg.DeviceMatrix(s)
    t0 = time.time()
    nt, ns = A.optimize_block_sweep(s, b, wlog=wlog, tile_rows=tr)
    tp = time.time() - t0
    row = {"plan": "sweep", "wlog": wlog, "tile_rows": tr, "nbuf": nbuf, "tiles": nt, "sweep_tiles": ns, "plan_s": round(tp, 2)}
    if nt == ns:
        pkg.set_tuning(spmmv_variant=9)
        for lay, nm in ((pkg.ROWWISE, "rowwise"), (pkg.COLWISE, "colwise")):
            Y.zero_(); pkg.spmmv(A, X, Y, b, ld, lay); torch.cuda.synchronize()
            row[nm + "_bitexact"] = bool(torch.equal(Y, Yref[lay]))
            run(A, lay, 20)
            ms = min(run(A, lay, 30) for _ in range(3))
            row[nm + "_ms"] = round(ms, 4); row[nm + "_frac"] = round(byts / ms / 1e6 / 8000, 4)
        pkg.set_tuning(spmmv_variant=0)
    print(json.dumps(row), flush=True)
    del A
pkg.set_tuning(sweep_nbuf=2)
