#!/usr/bin/env python3
"""Is the headline kernel's time a property of WHERE its arrays lie?  (tools/sigma_study.py timed two bit-identical structs -- sigma = 1, both
ordering modes -- 6 % apart in one process.)  One host struct (253^3 stencil, SELL-32-512), uploaded + planned again and again under different
allocation histories; per instance: kernel ms (HIP events, best of 3 x 30), the device addresses of its arrays.
  round A: build, time, free, build again ...            (same sizes re-allocated: the allocator hands the same ranges back?)
  round B: build a second instance while the first lives (different ranges), time both alternately
  round C: same instance, fresh x / y vectors each time"""
import ctypes
import json
import os
import sys
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
coo = pkg.gen_stencil27(g, g, g)
s = pkg.convert_to_scs(coo, 32, 512, pkg.F64); a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"])
del coo


def timeit(A, x, y):
    B.time_launches(0, 10, A=A, x=x, y=y)
    return round(min(B.time_launches(0, 30, A=A, x=x, y=y) for _ in range(3)), 4)


def addrs(A, x, y):
    d = {"values": A.values.data_ptr(), "col_idxs": A.col_idxs.data_ptr(), "x": x.data_ptr(), "y": y.data_ptr()}
    p = (ctypes.c_uint64 * 8)()
    if hasattr(B.lib(), "uspmv_dmat_plan_addresses") and B.lib().uspmv_dmat_plan_addresses(A.h, p) == 0:
        d.update({"tlc_col16": p[0], "tlc_lines": p[1], "tlc_line_ptr": p[2], "tlc_c16_ptrs": p[3]})
    return {k: hex(v) for k, v in d.items()}


def vecs():
    x = torch.full((s.n_rows_padded,), 5.0, dtype=torch.float64, device="cuda")
    return x, torch.zeros_like(x)


print(json.dumps({"free_total_GB": [round(v / 1e9, 2) for v in torch.cuda.mem_get_info()]}), flush=True)
for i in range(5):
    A = pkg.DeviceMatrix(s, tlc=True); x, y = vecs()
    print(json.dumps({"round": "A", "instance": i, "ms": timeit(A, x, y), **addrs(A, x, y)}), flush=True)
    del A, x, y
    torch.cuda.empty_cache()
A1 = pkg.DeviceMatrix(s, tlc=True); x1, y1 = vecs()
A2 = pkg.DeviceMatrix(s, tlc=True); x2, y2 = vecs()
for i in range(3):
    print(json.dumps({"round": "B", "pass": i, "first": timeit(A1, x1, y1), "second": timeit(A2, x2, y2), "first_with_second_vectors": timeit(A1, x2, y2),
                      "second_with_first_vectors": timeit(A2, x1, y1)}), flush=True)
print(json.dumps({"round": "B", "first": addrs(A1, x1, y1), "second": addrs(A2, x2, y2)}), flush=True)
del A2, x2, y2
keep = []
for i in range(4):
    x, y = vecs()
    print(json.dumps({"round": "C", "vectors": i, "ms": timeit(A1, x, y), "x": hex(x.data_ptr()), "y": hex(y.data_ptr())}), flush=True)
    keep.append((x, y))          # (held: every pair lies somewhere else)
