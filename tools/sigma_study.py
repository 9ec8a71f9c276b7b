#!/usr/bin/env python3
"""SURVEY 8(f)2's locality study: what the sorting scope sigma does to the tile-local-column kernel -- x lines a tile needs (the plan's own
count), padding (beta) and kernel time -- for sigma in {1, 32, 128, 512, 4096, 65536} on the 27-point stencil (253^3) and the KKT generator
(N = 200), SELL-32-sigma dp.  The matrix is converted ON THE DEVICE from device-resident COO arrays for every sigma
(uspmv_convert_to_scs_device_from_arrays), planned on the device, timed with HIP events.  Also: the conversion's own time."""
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
pkg.set_tuning(tlc_measure_tile=0)
which = sys.argv[1:] or ["stencil", "kkt"]
for name in which:
    m = pkg.gen_stencil27(253, 253, 253) if name == "stencil" else pkg.gen_kkt(200)
    I, J, V = m.arrays()
    dI, dJ, dV = torch.from_numpy(np.array(I)).cuda(), torch.from_numpy(np.array(J)).cuda(), torch.from_numpy(np.array(V)).cuda()
    n, nc, nnz = m.n_rows, m.n_cols, m.nnz
    del m, I, J, V
    for sigma in (1, 32, 128, 512, 4096, 65536):
        for mode, mname in ((pkg.SORT_HOST, "host std::sort on the counts"), (pkg.SORT_DEVICE_STABLE, "stable device ranking")):
            if mode == pkg.SORT_DEVICE_STABLE and sigma > 8192:
                continue
            torch.cuda.synchronize(); t0 = time.time()
            lay, A, o2n, n2o = pkg.convert_to_scs_device_from_arrays(dI, dJ, dV, n, nc, 32, sigma, pkg.F64, sort=mode, want_layout=False)
            torch.cuda.synchronize(); t_conv = time.time() - t0
            t0 = time.time()
            nt, ns = A.optimize_device()
            torch.cuda.synchronize(); t_plan = time.time() - t0
            x = torch.full((A.n_rows_padded,), 5.0, dtype=torch.float64, device="cuda"); y = torch.zeros_like(x)
            B.time_launches(0, 5, A=A, x=x, y=y)
            ms = min(B.time_launches(0, 30, A=A, x=x, y=y) for _ in range(3))
            byts = A.n_elements * 12 + 8 * A.n_chunks + 8 * n + 8 * A.n_rows_padded
            import ctypes
            meta = (ctypes.c_int64 * 4)()
            info = None
            if B.lib().uspmv_dmat_plan_download(A.h, meta, None, None, None, None) == 0 and meta[0]:
                info = {"mean": round(meta[1] / meta[0], 1), "max": int(meta[3])}
            print(json.dumps(dict(matrix=name, n=n, nnz=nnz, sigma=sigma, ordering=mname, beta=round(nnz / A.n_elements, 5), convert_s=round(t_conv, 3), plan_s=round(t_plan, 3),
                                  tiles=nt, staged=ns, rows_per_tile=A.tile_rows, x_lines_per_tile=info, kernel_ms=round(ms, 4), frac_of_8TBs=round(byts / ms / 1e6 / 8000, 4))), flush=True)
            del A, x, y
    del dI, dJ, dV
    torch.cuda.empty_cache()
