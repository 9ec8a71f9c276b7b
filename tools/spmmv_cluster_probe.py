#!/usr/bin/env python3
"""Config 3, phased SpMMV kernel: plan options side by side -- rows per phase 256 | 512 (`spmmv_phase_rows`), private row order
1 = ties undone | 2 = rows clustered per tile (`spmmv_reorder`); both layouts, every line checked against the gather kernel."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
from ultimate_spmv_amd import binding as B
torch.cuda.set_device(0)
pkg.set_tuning(spmmv_list_plan=1)      # these probes time the older block-plan kernels too
g = int(sys.argv[1]) if len(sys.argv) > 1 else 111
coo = pkg.gen_stencil27(g, g, g, dof=3)
s = pkg.convert_to_scs(coo, 32, 512, pkg.F64); a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"])
b, ld = 8, s.n_rows_padded
X = torch.rand(b * ld, dtype=torch.float64, device="cuda"); Y = torch.zeros_like(X)
A0 = pkg.DeviceMatrix(s)
Y0 = {}
for lay in (pkg.ROWWISE, pkg.COLWISE):
    pkg.set_tuning(spmmv_variant=3); y = torch.zeros_like(X); pkg.spmmv(A0, X, y, b, ld, lay); Y0[lay] = y
pkg.set_tuning(spmmv_variant=8)
for reorder in (1, 2):
    for rows in (256, 512):
        pkg.set_tuning(spmmv_reorder=reorder, spmmv_phase_rows=rows)
        t0 = time.time()
        A = pkg.DeviceMatrix(s, block_tlc=b)
        plan_s = time.time() - t0
        for lay, nm in ((pkg.ROWWISE, "rowwise"), (pkg.COLWISE, "colwise")):
            Y.fill_(-1.0); pkg.spmmv(A, X, Y, b, ld, lay)
            same = bool(torch.equal(Y, Y0[lay]))
            B.time_launches(5, 5, A=A, x=X, y=Y, b=b, ld=ld, layout=lay)
            ms = min(B.time_launches(5, 40, A=A, x=X, y=Y, b=b, ld=ld, layout=lay) for _ in range(3))
            print(json.dumps(dict(reorder=reorder, phase_rows=rows, layout=nm, plan_s=round(plan_s, 1), bitexact=same, ms=round(ms, 4))), flush=True)
        del A
pkg.set_tuning(spmmv_reorder=1, spmmv_phase_rows=256, spmmv_variant=0)
