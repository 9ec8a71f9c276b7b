#!/usr/bin/env python3
"""Does the tie order of the sigma sort cost the tile-local-column kernel anything?  ONE process, ONE pair of vectors (x, y: the placement effect of
DESIGN 9.1 is held fixed), handles of the nlpkkt200-class stencil converted on the device with sigma in {1, 512} and the ties either as std::sort
leaves them (the reference's order) or in original order (stable ranking); 256- and 512-row tiles; timed alternately, several rounds."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
from ultimate_spmv_amd import binding as B
torch.cuda.set_device(0)
pkg.set_tuning(tlc_measure_tile=0)
g = int(sys.argv[1]) if len(sys.argv) > 1 else 253
m = pkg.gen_stencil27(g, g, g)
I, J, V = m.arrays()
dI, dJ, dV = torch.from_numpy(np.array(I)).cuda(), torch.from_numpy(np.array(J)).cuda(), torch.from_numpy(np.array(V)).cuda()
n, nc = m.n_rows, m.n_cols
del m, I, J, V
handles = []
for sigma, mode, name in ((512, pkg.SORT_HOST, "s512 std::sort ties"), (512, pkg.SORT_DEVICE_STABLE, "s512 ties in original order"), (1, pkg.SORT_HOST, "s1")):
    for tr in (256, 512):
        pkg.set_tuning(tlc_tile_rows=tr)
        lay, A, o2n, n2o = pkg.convert_to_scs_device_from_arrays(dI, dJ, dV, n, nc, 32, sigma, pkg.F64, sort=mode, want_layout=False)
        A.optimize_device()
        handles.append((f"{name}, {tr}-row tiles", A))
pkg.set_tuning(tlc_tile_rows=0)
npad = handles[0][1].n_rows_padded
x = torch.full((npad,), 5.0, dtype=torch.float64, device="cuda"); y = torch.zeros_like(x)
for name, A in handles:
    B.time_launches(0, 30, A=A, x=x, y=y)
res = {name: [] for name, _ in handles}
for rnd in range(6):
    for name, A in handles:
        res[name].append(round(B.time_launches(0, 40, A=A, x=x, y=y), 4))
for name, _ in handles:
    v = sorted(res[name])
    print(json.dumps(dict(handle=name, ms_min=v[0], ms_med=v[len(v) // 2], ms=res[name])), flush=True)
