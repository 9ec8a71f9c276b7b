#!/usr/bin/env python3
"""Config 3 (Queen_4147-class, dp b = 8 and sp b = 16): the phased-plan kernel (variant 8), both layouts; every line checked
against the gather kernel (variant 3)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
from ultimate_spmv_amd import binding as B
torch.cuda.set_device(0)
g = int(sys.argv[1]) if len(sys.argv) > 1 else 111
coo = pkg.gen_stencil27(g, g, g, dof=3)
for dt, tdt, vs, b in ((pkg.F64, torch.float64, 8, 8), (pkg.F32, torch.float32, 4, 16)):
    s = pkg.convert_to_scs(coo, 32, 512, dt); a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"])
    ld = s.n_rows_padded
    X = torch.rand(b * ld, dtype=tdt, device="cuda"); Y = torch.zeros_like(X)
    A0 = pkg.DeviceMatrix(s)
    byts = s.n_elements * (vs + 4) + 8 * s.n_chunks + 2 * b * vs * ld
    for wgs in (8,):
        A = pkg.DeviceMatrix(s, block_tlc=b)
        for lay, nm in ((pkg.ROWWISE, "rowwise"), (pkg.COLWISE, "colwise")):
            pkg.set_tuning(spmmv_variant=3); Y0 = torch.zeros_like(X); pkg.spmmv(A0, X, Y0, b, ld, lay)
            for var in (8,):
                pkg.set_tuning(spmmv_variant=var)
                Y.fill_(-1.0); pkg.spmmv(A, X, Y, b, ld, lay)
                same = bool(torch.equal(Y, Y0))
                B.time_launches(5, 5, A=A, x=X, y=Y, b=b, ld=ld, layout=lay)
                ms = min(B.time_launches(5, 40, A=A, x=X, y=Y, b=b, ld=ld, layout=lay) for _ in range(3))
                print(json.dumps(dict(dtype=vs, b=b, layout=nm, variant=var, index_bytes=1, wgs_per_cu=wgs, bitexact=same, ms=round(ms, 4), TF=round(2 * s.nnz * b / ms / 1e9, 2), frac=round(byts / ms / 1e6 / 8000, 3))), flush=True)
        pkg.set_tuning(spmmv_variant=0)
        del A
