#!/usr/bin/env python3
"""Config 3, phased SpMMV kernel: the plan's private row order side by side -- 1 = ties undone (what the device-side builder does),
2 = balls over all slots, 3 = flat bricks of `lines` mesh lines (needs the line stride: measurement aid), 4 = flat patches grown along the
slots of one phase (the host planner's default) -- with the phases cut
greedily or by dynamic programming (`spmmv_phase_dp`); both layouts, every line checked against the lane-per-row kernel."""
import os, sys, json, time
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
A0 = pkg.DeviceMatrix(s)
Y0 = {}
for lay in (pkg.ROWWISE, pkg.COLWISE):
    pkg.set_tuning(spmmv_variant=3); y = torch.zeros_like(X); pkg.spmmv(A0, X, y, b, ld, lay); Y0[lay] = y
pkg.set_tuning(spmmv_variant=0)
del A0
cases = [(1, 0, 4), (4, 24, 4), (3, 24, 4), (2, 24, 4), (1, 24, 4), (4, 0, 4), (4, 24, 4), (1, 0, 4)]
for reorder, dp, lines in cases:
    pkg.set_tuning(spmmv_reorder=reorder, spmmv_phase_dp=dp, spmmv_brick_stride=3 * g, spmmv_brick_lines=lines)
    t0 = time.time()
    A = pkg.DeviceMatrix(s, block_tlc=b)
    plan_s = time.time() - t0
    for lay, nm in ((pkg.ROWWISE, "rowwise"), (pkg.COLWISE, "colwise")):
        Y.fill_(-1.0); pkg.spmmv(A, X, Y, b, ld, lay)
        same = bool(torch.equal(Y, Y0[lay]))
        B.time_launches(5, 20, A=A, x=X, y=Y, b=b, ld=ld, layout=lay)
        ms = sorted(B.time_launches(5, 40, A=A, x=X, y=Y, b=b, ld=ld, layout=lay) for _ in range(5))
        print(json.dumps(dict(reorder=reorder, phase_dp=dp, lines=lines, rows_staged=A.block_plan_info()["rows_staged"], layout=nm, plan_s=round(plan_s, 1), bitexact=same, ms_min=round(ms[0], 4), ms_med=round(ms[2], 4))), flush=True)
    del A
pkg.set_tuning(spmmv_reorder=4, spmmv_phase_dp=24, spmmv_brick_stride=0)
