import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
from ultimate_spmv_amd import binding as B
torch.cuda.set_device(0)
pkg.set_tuning(spmmv_list_plan=1)      # these probes time the older block-plan kernels too
coo = pkg.gen_stencil27(111, 111, 111, dof=3)
s = pkg.convert_to_scs(coo, 32, 512, pkg.F32); pkg.permute_scs_cols(s, s.arrays()["old_to_new_idx"])
b, ld = 8, s.n_rows_padded
X = torch.rand(b * ld, dtype=torch.float32, device="cuda"); Y = torch.zeros_like(X); Y0 = torch.zeros_like(X)
A0 = pkg.DeviceMatrix(s)
pkg.spmmv(A0, X, Y0, b, ld, pkg.ROWWISE)
print(json.dumps(dict(gather_ms=round(B.time_launches(5, 20, A=A0, x=X, y=Y, b=b, ld=ld, layout=pkg.ROWWISE), 4))), flush=True)
for tr in (64, 32):
    pkg.set_tuning(spmmv_tile_rows=tr)
    A = pkg.DeviceMatrix(s, block_tlc=b)
    for lay, nm in ((pkg.ROWWISE, "rowwise"), (pkg.COLWISE, "colwise")):
        pkg.spmmv(A, X, Y, b, ld, lay)
        ms = B.time_launches(5, 20, A=A, x=X, y=Y, b=b, ld=ld, layout=lay)
        pkg.spmmv(A0, X, Y0, b, ld, lay)
        print(json.dumps(dict(tile_rows=tr, layout=nm, staged=[A.block_staged, A.block_tiles], ms=round(ms, 4), same=bool(torch.equal(Y, Y0)))), flush=True)
    del A
pkg.set_tuning(spmmv_tile_rows=0)
