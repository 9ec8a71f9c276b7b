#!/usr/bin/env python3
"""The tile-local-column plan over 16-element lines against the plan over single x elements ("tlc_elem" 2 forces it) on the nlpkkt200-class stencil:
one process, one pair of vectors, alternating."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
from ultimate_spmv_amd import binding as B
torch.cuda.set_device(0)
g = int(sys.argv[1]) if len(sys.argv) > 1 else 253
pkg.set_tuning(tlc_measure_tile=0)
m = pkg.gen_stencil27(g, g, g)
s = pkg.convert_to_scs(m, 32, 512, pkg.F64); a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"])
hs = []
for e in (1, 2):
    pkg.set_tuning(tlc_elem=e)
    A = pkg.DeviceMatrix(s, tlc=True)
    hs.append((f"elements per list entry {A.plan_granularity()}, index bits {A.index_bits()}", A))
pkg.set_tuning(tlc_elem=1)
x = torch.rand(s.n_rows_padded, dtype=torch.float64, device="cuda"); y = torch.zeros_like(x); y0 = None
res = {n: [] for n, _ in hs}
for n, A in hs:
    pkg.spmv(A, x, y)
    if y0 is None: y0 = y.clone()
    print(n, "bit-identical:", bool(torch.equal(y, y0)), flush=True)
    B.time_launches(0, 30, A=A, x=x, y=y)
for r in range(5):
    for n, A in hs:
        res[n].append(round(B.time_launches(0, 40, A=A, x=x, y=y), 4))
for n, _ in hs:
    print(json.dumps(dict(plan=n, ms=res[n])), flush=True)
