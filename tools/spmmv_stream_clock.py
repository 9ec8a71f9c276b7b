#!/usr/bin/env python3
"""Per-workgroup start / end times of ONE launch of the streaming SpMMV kernel (USPMV_STREAM_CLOCK, 100 MHz counter): does a static split of equal
work end at the same time on every CU?  Prints the spread of the end times overall and per XCD (workgroup w runs on XCD w % 8)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = "/tmp/uspmv_stream_clock.txt"
os.environ["USPMV_STREAM_CLOCK"] = out
import torch, numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
torch.cuda.set_device(0)
g = int(sys.argv[1]) if len(sys.argv) > 1 else 111
wgs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
coo = pkg.gen_stencil27(g, g, g, dof=3)
s = pkg.convert_to_scs(coo, 32, 512, pkg.F64); a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"])
b, ld = 8, s.n_rows_padded
X = torch.rand(b * ld, dtype=torch.float64, device="cuda"); Y = torch.zeros_like(X)
pkg.set_tuning(spmmv_stream=wgs)
A = pkg.DeviceMatrix(s, block_tlc=b)
for k in range(12):
    pkg.spmmv(A, X, Y, b, ld, pkg.ROWWISE)          # (every launch rewrites the file; the last one is read)
torch.cuda.synchronize()
d = np.loadtxt(out)
w, t0, t1 = d[:, 0].astype(int), d[:, 1] / 100.0, d[:, 2] / 100.0      # microseconds
dur = t1 - t0
q = lambda v: [round(float(x), 1) for x in np.percentile(v, [0, 10, 50, 90, 100])]
print(json.dumps(dict(wgs_per_cu=wgs, grid=len(w), start_us_pct=q(t0), end_us_pct=q(t1), duration_us_pct=q(dur))))
for x in range(8):
    m = (w % 8) == x
    print(json.dumps(dict(xcd=x, end_us_pct=q(t1[m]), duration_us_pct=q(dur[m]))))
