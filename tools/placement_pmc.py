#!/usr/bin/env python3
"""Fast / slow placement pair of the headline kernel under rocprofv3 counters: the same matrix, x fixed, y at a 2 MiB boundary (fast) and
1 MiB behind it (slow), alternating launches F S F S ... so that one counter pass sees both.  Run as
  rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <dir> -- python3 tools/placement_pmc.py
and summarise with tools/placement_pmc_summary.py <dir> (dispatch parity = placement).  Without a profiler it just prints the two times."""
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
g = 253
coo = pkg.gen_stencil27(g, g, g)
s = pkg.convert_to_scs(coo, 32, 512, pkg.F64); a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"])
del coo
A = pkg.DeviceMatrix(s, tlc=True)
nb = s.n_rows_padded * 8
MB = 1 << 20
R = torch.zeros(768 * MB, dtype=torch.uint8, device="cuda")
x = R[0:nb].view(torch.float64); x.fill_(5.0)
yf = R[256 * MB:256 * MB + nb].view(torch.float64)
ys = R[257 * MB:257 * MB + nb].view(torch.float64)
for _ in range(3):
    pkg.spmv(A, x, yf); pkg.spmv(A, x, ys)
torch.cuda.synchronize()
tf = min(B.time_launches(0, 20, A=A, x=x, y=yf) for _ in range(2))
ts = min(B.time_launches(0, 20, A=A, x=x, y=ys) for _ in range(2))
print(json.dumps({"fast_ms": round(tf, 4), "slow_ms": round(ts, 4)}), flush=True)
torch.cuda.synchronize()
# marker: a stream-read launch, then 8 alternating SpMV launches (even = fast, odd = slow)
part = torch.empty(8192, dtype=torch.float64, device="cuda")
B.time_launches(3, 1, x=R.view(torch.float64), y=part, n=1 << 20)
torch.cuda.synchronize()
for i in range(8):
    pkg.spmv(A, x, yf if i % 2 == 0 else ys)
    torch.cuda.synchronize()
