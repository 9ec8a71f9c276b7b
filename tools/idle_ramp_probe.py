#!/usr/bin/env python3
"""Does a kernel's time depend on how long the GPU idled before it?  After `idle` seconds without GPU work, consecutive batches of 5
launches are timed (HIP events) -- if the first batches are slower, short measurement loops behind host-side work (bench.py's
other_configs: numpy set-up, CPU baseline legs) report the power-state ramp, not the kernel.  Usage: idle_ramp_probe.py <2|3>"""
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
from ultimate_spmv_amd import binding as B
torch.cuda.set_device(0)
cfg = sys.argv[1] if len(sys.argv) > 1 else "3"
if cfg == "3":
    g = 111
    coo = pkg.gen_stencil27(g, g, g, dof=3)
    s = pkg.convert_to_scs(coo, 32, 512, pkg.F64); a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"])
    b, ld = 8, s.n_rows_padded
    X = torch.rand(b * ld, dtype=torch.float64, device="cuda"); Y = torch.zeros_like(X)
    A = pkg.DeviceMatrix(s, block_tlc=b)
    run = lambda n: B.time_launches(5, n, A=A, x=X, y=Y, b=b, ld=ld, layout=pkg.ROWWISE)
else:
    g = 253
    coo = pkg.gen_stencil27(g, g, g)
    s = pkg.convert_to_scs(coo, 32, 512, pkg.F64); a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"])
    A = pkg.DeviceMatrix(s, tlc=True)
    x = torch.full((s.n_rows_padded,), 5.0, dtype=torch.float64, device="cuda"); y = torch.zeros_like(x)
    run = lambda n: B.time_launches(0, n, A=A, x=x, y=y)
del coo


def smi():
    out = {}
    import glob
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_*clk"):
        try:
            act = [ln.strip() for ln in open(f).read().splitlines() if ln.strip().endswith("*")]
            out[os.path.basename(f)[7:]] = act[0] if act else None
        except OSError:
            pass
    return out


run(30)
for idle in (0.0, 0.2, 1.0, 5.0, 0.0):
    torch.cuda.synchronize()
    time.sleep(idle)
    before = smi()
    batches = [round(run(3), 4) for _ in range(12)]      # (each call: 2 untimed + 3 timed launches)
    print(json.dumps(dict(config=cfg, idle_s=idle, clocks_before=before, ms_per_launch_batches_of_3=batches, clocks_after=smi())), flush=True)
