import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch as t
import __graft_entry__ as ge
pkg = ge.load_package()
from ultimate_spmv_amd import binding as B
t.cuda.set_device(0)
for name, gen in (("banded_4M_30_2000", lambda: pkg.gen_banded_random(4000000, 30, 2000)),
                  ("banded_2M_60_8000", lambda: pkg.gen_banded_random(2000000, 60, 8000)),
                  ("banded_2M_100_20000", lambda: pkg.gen_banded_random(2000000, 100, 20000)),
                  ("kkt_160", lambda: pkg.gen_kkt(160))):
    coo = gen()
    s = pkg.convert_to_scs(coo, 32, 512, pkg.F64)
    a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"])
    x = t.ones(s.n_rows_padded, dtype=t.float64, device="cuda"); y = t.zeros_like(x)
    byts = s.n_elements * 12 + 8 * s.n_chunks + 8 * s.n_rows + 8 * s.n_rows_padded
    out = dict(matrix=name, n=s.n_rows, nnz=s.nnz)
    A = pkg.DeviceMatrix(s); A.optimize(s); pkg.spmv(A, x, y); y0 = y.clone()
    ms = B.time_launches(0, 30, A=A, x=x, y=y)
    out["auto"] = dict(plan=list(A.plan_info()), ms=round(ms, 5), frac=round(byts / (ms * 1e-3) / 8e12, 4))
    A2 = pkg.DeviceMatrix(s)
    try:
        nt, ns = A2.optimize_sweep(s)
        pkg.spmv(A2, x, y)
        ms = B.time_launches(0, 30, A=A2, x=x, y=y)
        out["sweep"] = dict(tiles=[nt, ns], plan=list(A2.plan_info()), ms=round(ms, 5), frac=round(byts / (ms * 1e-3) / 8e12, 4), same=bool(t.equal(y, y0)))
    except Exception as e:
        out["sweep"] = str(e)[:100]
    print(json.dumps(out), flush=True)
