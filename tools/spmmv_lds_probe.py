import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
from ultimate_spmv_amd import binding as B
torch.cuda.set_device(0)
pkg.set_tuning(spmmv_list_plan=1)      # these probes time the older block-plan kernels too
coo = pkg.gen_stencil27(111, 111, 111, dof=3)
s = pkg.convert_to_scs(coo, 32, 512, pkg.F64); pkg.permute_scs_cols(s, s.arrays()["old_to_new_idx"])
b, ld = 8, s.n_rows_padded
X = torch.rand(b * ld, dtype=torch.float64, device="cuda"); Y = torch.zeros_like(X); Y0 = torch.zeros_like(X)
A0 = pkg.DeviceMatrix(s)
pkg.spmmv(A0, X, Y0, b, ld, pkg.ROWWISE)
print(json.dumps(dict(gather_ms=round(B.time_launches(5, 20, A=A0, x=X, y=Y, b=b, ld=ld, layout=pkg.ROWWISE), 4))), flush=True)
for kb in (0, 44, 36, 30, 24):
    pkg.set_tuning(spmmv_lds_kb=kb)
    A = pkg.DeviceMatrix(s, block_tlc=b)
    pkg.set_tuning(spmmv_variant=4)
    pkg.spmmv(A, X, Y, b, ld, pkg.ROWWISE)
    ms = B.time_launches(5, 20, A=A, x=X, y=Y, b=b, ld=ld, layout=pkg.ROWWISE)
    pkg.set_tuning(spmmv_variant=0)
    print(json.dumps(dict(lds_kb=kb, staged=[A.block_staged, A.block_tiles], ms=round(ms, 4), same=bool(torch.equal(Y, Y0)))), flush=True)
    del A
pkg.set_tuning(spmmv_lds_kb=0)
