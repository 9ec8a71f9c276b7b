#!/usr/bin/env python3
"""SURVEY 8(f)1 at scale, on the CPU (this needs /root/reference through oracle/_ref, so it runs in the build container): a symmetric
MatrixMarket file with >= 1e8 stored entries (the KKT generator's matrix, lower triangle) is read by uspmv_read_mtx and by the reference's
own read_mtx (code/utilities.hpp:2148-2309 over code/mmio.h:132-263), the two COOs are compared element by element, and the rank-0 side of
a P = 8 run is timed: the seg-nnz cut + the eight per-rank blocks (here: binary block files; reference: seg_work_sharing_arr +
seg_mtx_struct per rank, code/mpi_funcs.hpp:424-622, :739-860, without its MPI_Send).
Usage: mtx_scale_probe.py [N=156] [dir=/tmp]"""
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("USPMV_NO_TORCH", "1")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from oracle import refshim

N = int(sys.argv[1]) if len(sys.argv) > 1 else 156
d = sys.argv[2] if len(sys.argv) > 2 else "/tmp"
path = os.path.join(d, f"kkt_{N}_sym.mtx")
cores = len(os.sched_getaffinity(0))
out = {"matrix": f"uspmv_gen_kkt N={N}, lower triangle as a 'symmetric' MatrixMarket file", "host_cores": cores}
t0 = time.time()
m = pkg.gen_kkt(N)
out["generated_entries"] = m.nnz
m.write_mtx(path, symmetric=True)
out["write_s"] = round(time.time() - t0, 1)
out["file_GB"] = round(os.path.getsize(path) / 1e9, 3)
del m
with open(path) as f:
    f.readline()
    out["stored_entries"] = int(f.readline().split()[2])
os.system(f"cat {path} > /dev/null")                       # both readers start from a warm page cache
t0 = time.time(); a = pkg.read_mtx(path); t_own_first = time.time() - t0
del a
t0 = time.time(); r = refshim.RefMtx.read(path); t_ref = time.time() - t0
t0 = time.time(); a = pkg.read_mtx(path); t_own = time.time() - t0      # (second call in the process: the kernel has huge pages at hand)
I, J, V = a.arrays()
rI, rJ, rV = r.arrays()
same = bool(a.n_rows == r.n_rows and a.nnz == r.nnz and np.array_equal(I, rI) and np.array_equal(J, rJ) and np.array_equal(V.view(np.uint64), rV.view(np.uint64)))
out.update({"n_rows": a.n_rows, "nnz_expanded": a.nnz, "uspmv_read_mtx_first_call_s": round(t_own_first, 2), "uspmv_read_mtx_s": round(t_own, 2), "reference_read_mtx_s": round(t_ref, 2), "speedup": round(t_ref / t_own, 1),
            "uspmv_read_mtx_Mentries_per_s": round(out["stored_entries"] / t_own / 1e6, 1), "reference_Mentries_per_s": round(out["stored_entries"] / t_ref / 1e6, 1), "identical_coo": same})
print(json.dumps(out), flush=True)
assert same, "the two readers disagree"
# ---- rank 0's share of a P = 8 set-up
P = 8
t0 = time.time()
wsa = pkg.seg_work_sharing_arr(a, "seg-nnz", P)
blocks = []
for rk in range(P):
    blk = pkg.seg_local_coo(a, wsa, rk)
    blk.save(os.path.join(d, f"kkt_{N}_block{rk}.uspmvcoo"))
    blocks.append((blk.n_rows, blk.nnz))
    del blk
t_cut = time.time() - t0
t0 = time.time()
wsa_r = refshim.seg_work_sharing_arr(r, "seg-nnz", P)
for rk in range(P):
    lb = refshim.seg_local_mtx(r, wsa_r, rk)
    del lb
t_cut_ref = time.time() - t0
t0 = time.time()
b0 = pkg.Coo.load(os.path.join(d, f"kkt_{N}_block3.uspmvcoo"))
t_load = time.time() - t0
out2 = {"P": P, "wsa_equal": bool(np.array_equal(wsa, wsa_r)), "cut_and_write_8_block_files_s": round(t_cut, 2), "reference_cut_8_blocks_in_memory_s": round(t_cut_ref, 2),
        "one_rank_loads_its_block_s": round(t_load, 2), "block_rows_nnz": blocks}
print(json.dumps(out2), flush=True)
for rk in range(P):
    os.unlink(os.path.join(d, f"kkt_{N}_block{rk}.uspmvcoo"))
os.unlink(path)
