#!/usr/bin/env python3
"""Config 4b (HV15R-class banded-random, SELL-32-512): the column-window sweep with ONE 128-KiB-window workgroup per CU (the default) against
TWO 64-KiB-window workgroups per CU (one's staging + barrier behind the other's FMAs), for ap[dp_sp] and plain dp; window / threads / tile
rows / buffers via the tuning keys, the plan rebuilt for every setting.  Bit-exactness of the settings is tests/test_gpu_sweep.py's job."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
from ultimate_spmv_amd import binding as B
torch.cuda.set_device(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2017169
coo = pkg.gen_banded_random(n, 140, 50000, magnitude_decades=10.0)
dp, sp = pkg.partition_precisions(coo, 1e-3)
ds = pkg.convert_to_scs(dp, 32, 512, B.F64)
perm = ds.arrays()["old_to_new_idx"].copy()
ss = pkg.convert_to_scs(sp, 32, 512, B.F32, fixed_permutation=perm)
pkg.permute_scs_cols(ds, perm); pkg.permute_scs_cols(ss, perm)
s = pkg.convert_to_scs(coo, 32, 512, B.F64); a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"])
x = torch.full((ds.n_rows_padded,), 1.5, dtype=torch.float64, device="cuda"); y = torch.zeros_like(x)
b_ap = 12 * ds.n_elements + 8 * ss.n_elements + 16 * ds.n_chunks + 8 * (ds.n_rows + ds.n_rows_padded)
b_dp = s.n_elements * 12 + 8 * s.n_chunks + 8 * (s.n_rows + s.n_rows_padded)
settings = [dict(), dict(sweep_wlog=13, sweep_nbuf=1), dict(sweep_wlog=13, sweep_nbuf=1, sweep_threads=512), dict(sweep_wlog=13, sweep_nbuf=1, sweep_threads=512, sweep_tile_rows=2048),
            dict(sweep_wlog=13, sweep_nbuf=1, sweep_threads=512, sweep_tile_rows=1024), dict(sweep_wlog=13, sweep_nbuf=1, sweep_tile_rows=2048), dict(sweep_wlog=13, sweep_nbuf=2),
            dict(sweep_wlog=14, sweep_threads=512), dict(sweep_wlog=12, sweep_nbuf=1, sweep_threads=512, sweep_tile_rows=2048), dict(sweep_wlog=12, sweep_nbuf=1, sweep_threads=256, sweep_tile_rows=1024)]
base = dict(sweep_wlog=0, sweep_nbuf=2, sweep_threads=0, sweep_tile_rows=0)
for st in settings:
    pkg.set_tuning(**base)
    if "sweep_nbuf" not in st:
        pass
    pkg.set_tuning(**st)
    row = {"setting": st or "default"}
    try:
        Ad, As = pkg.DeviceMatrix(ds), pkg.DeviceMatrix(ss)
        pkg.optimize_ap(Ad, As, ds, ss)
        kind = Ad.plan_info()
        B.time_launches(4, 10, A=Ad, B=As, x=x, y=y)
        ms = min(B.time_launches(4, 30, A=Ad, B=As, x=x, y=y) for _ in range(3))
        row.update(ap_ms=round(ms, 4), ap_frac=round(b_ap / ms / 1e6 / 8000, 4), ap_plan=list(kind))
        del Ad, As
        A = pkg.DeviceMatrix(s, tlc=True)
        kind = A.plan_info()
        B.time_launches(0, 10, A=A, x=x, y=y)
        ms = min(B.time_launches(0, 30, A=A, x=x, y=y) for _ in range(3))
        row.update(dp_ms=round(ms, 4), dp_frac=round(b_dp / ms / 1e6 / 8000, 4), dp_plan=list(kind))
        del A
    except Exception as e:
        row["error"] = str(e)[:200]
    print(json.dumps(row), flush=True)
pkg.set_tuning(**base)
