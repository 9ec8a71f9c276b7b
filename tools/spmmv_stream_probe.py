#!/usr/bin/env python3
"""Config 3 (and small checks): the phased SpMMV kernel, one tile per workgroup (scs_spmmv_quadph) against persistent workgroups walking the
same plan as a stream (scs_spmmv_pstream, "spmmv_stream" = workgroups per CU) -- alternating on one box, both layouts, every result compared
bit for bit with the lane-per-row kernel."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
from ultimate_spmv_amd import binding as B
torch.cuda.set_device(0)


def check_small():
    ok = True
    for name, gen in (("stencil 14^3 x 3", lambda: pkg.gen_stencil27(14, 14, 14, dof=3)), ("stencil 9x7x5", lambda: pkg.gen_stencil27(9, 7, 5, dof=1)),
                      ("stencil 20x17x3 x 2", lambda: pkg.gen_stencil27(20, 17, 3, dof=2))):
        coo = gen()
        for sigma in (1, 64, 512):
            s = pkg.convert_to_scs(coo, 32, sigma, pkg.F64); a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"])
            b, ld = 8, s.n_rows_padded
            X = torch.rand(b * ld, dtype=torch.float64, device="cuda") - 0.5
            for wgs, depth in ((1, 1), (4, 1), (1, 2), (3, 2), (99, 1)):
                pkg.set_tuning(spmmv_stream=0, spmmv_variant=3)
                A0 = pkg.DeviceMatrix(s)
                ref = {}
                for lay in (pkg.ROWWISE, pkg.COLWISE):
                    y = torch.zeros_like(X); pkg.spmmv(A0, X, y, b, ld, lay); ref[lay] = y
                pkg.set_tuning(spmmv_variant=0, spmmv_stream=wgs, spmmv_stream_depth=depth)
                A = pkg.DeviceMatrix(s, block_tlc=b)
                for lay in (pkg.ROWWISE, pkg.COLWISE):
                    Y = torch.full_like(X, -7.0); pkg.spmmv(A, X, Y, b, ld, lay)
                    same = bool(torch.equal(Y[: b * s.n_rows] if lay == pkg.ROWWISE else Y, ref[lay][: b * s.n_rows] if lay == pkg.ROWWISE else ref[lay]))
                    if lay == pkg.COLWISE:
                        same = all(bool(torch.equal(Y[v * ld: v * ld + s.n_rows], ref[lay][v * ld: v * ld + s.n_rows])) for v in range(b))
                    ok &= same
                    print(json.dumps(dict(check=name, sigma=sigma, wgs_per_cu=wgs, depth=depth, layout="row" if lay == pkg.ROWWISE else "col", bitexact=same)), flush=True)
                del A, A0
    pkg.set_tuning(spmmv_stream=0, spmmv_stream_depth=1, spmmv_stream_waves=4)
    return ok


if not os.environ.get("SKIP_SMALL") and not check_small():
    print("SMALL CHECKS FAILED", flush=True)
    sys.exit(1)

g = int(sys.argv[1]) if len(sys.argv) > 1 else 111
coo = pkg.gen_stencil27(g, g, g, dof=3)
s = pkg.convert_to_scs(coo, 32, 512, pkg.F64); a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"])
b, ld = 8, s.n_rows_padded
X = torch.rand(b * ld, dtype=torch.float64, device="cuda"); Y = torch.zeros_like(X)
pkg.set_tuning(spmmv_stream=0, spmmv_variant=3)
A0 = pkg.DeviceMatrix(s)
Y0 = {}
for lay in (pkg.ROWWISE, pkg.COLWISE):
    y = torch.zeros_like(X); pkg.spmmv(A0, X, y, b, ld, lay); Y0[lay] = y
pkg.set_tuning(spmmv_variant=0)
del A0
# cases: workgroups per CU (0 = the one-tile-per-workgroup kernel); a negative number = that many with tile t -> workgroup t % grid
# "n:2" = depth 2 (X rows and entries two phases ahead, three LDS buffers, partial waits)
raw_cases = sys.argv[2].split(",") if len(sys.argv) > 2 else "0,3,3:2,0,3:2".split(",")
# "99" = one tile per workgroup (only the phases of a tile pipelined); "n:1:5" = register budget of five waves per SIMD
def _case(c):
    f = c.split(":")
    return int(f[0]), int(f[1]) if len(f) > 1 else 1, int(f[2]) if len(f) > 2 else 4
cases = [_case(c) for c in raw_cases]
abl = [int(c) for c in (sys.argv[3].split(",") if len(sys.argv) > 3 else [])]
for wgs, depth, waves in cases:
    pkg.set_tuning(spmmv_stream=abs(wgs), spmmv_stream_xcd=1 if wgs >= 0 else 0, spmmv_stream_depth=depth, spmmv_stream_waves=waves)
    A = pkg.DeviceMatrix(s, block_tlc=b)
    for lay, nm in ((pkg.ROWWISE, "rowwise"), (pkg.COLWISE, "colwise")):
        Y.fill_(-1.0); pkg.spmmv(A, X, Y, b, ld, lay)
        same = bool(torch.equal(Y, Y0[lay]))
        B.time_launches(5, 20, A=A, x=X, y=Y, b=b, ld=ld, layout=lay)
        ms = sorted(B.time_launches(5, 40, A=A, x=X, y=Y, b=b, ld=ld, layout=lay) for _ in range(5))
        print(json.dumps(dict(wgs_per_cu=wgs, depth=depth if wgs else 0, waves=waves if wgs else 0, kernel="pstream" if wgs else "quadph", layout=nm, bitexact=same, ms_min=round(ms[0], 4), ms_med=round(ms[2], 4))), flush=True)
    if wgs and depth == 1:
        for ab in abl:
            pkg.set_tuning(ablate=ab, spmmv_variant=8)
            B.time_launches(5, 20, A=A, x=X, y=Y, b=b, ld=ld, layout=pkg.ROWWISE)
            ms = sorted(B.time_launches(5, 40, A=A, x=X, y=Y, b=b, ld=ld, layout=pkg.ROWWISE) for _ in range(3))
            print(json.dumps(dict(wgs_per_cu=wgs, ablate=ab, ms_min=round(ms[0], 4))), flush=True)
        pkg.set_tuning(ablate=0, spmmv_variant=0)
    del A
pkg.set_tuning(spmmv_stream=0, spmmv_stream_depth=1, spmmv_stream_waves=4)
