#!/usr/bin/env python3
"""Set-up time of the host path (convert_to_scs + permute_scs_cols + upload) against uspmv_convert_to_scs_device
on the nlpkkt200-class matrix; also checks that both produce the same device arrays."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
torch.cuda.set_device(0)
g = int(sys.argv[1]) if len(sys.argv) > 1 else 253
t0 = time.time(); coo = pkg.gen_stencil27(g, g, g); t_gen = time.time() - t0
for rep in range(2):
    t0 = time.time()
    s = pkg.convert_to_scs(coo, 32, 512, pkg.F64); t1 = time.time()
    pkg.permute_scs_cols(s, s.arrays()["old_to_new_idx"]); t2 = time.time()
    A = pkg.DeviceMatrix(s); torch.cuda.synchronize(); t3 = time.time()
    lay, Ad = pkg.convert_to_scs_device(coo, 32, 512, pkg.F64); t4 = time.time()
    Ad.optimize_device(); torch.cuda.synchronize(); t5 = time.time()
    A.optimize(s); torch.cuda.synchronize(); t6 = time.time()
    same = None
    if rep == 1:
        d = pkg.dmat_download(Ad); a = s.arrays()
        same = bool(np.array_equal(d["col_idxs"], a["col_idxs"]) and np.array_equal(d["values"], a["values"]) and np.array_equal(d["chunk_ptrs"], a["chunk_ptrs"]))
    print(json.dumps(dict(grid=g, nnz=coo.nnz, gen_s=round(t_gen, 2), host_convert_s=round(t1 - t0, 2), host_permute_cols_s=round(t2 - t1, 2),
                          upload_s=round(t3 - t2, 2), host_path_total_s=round(t3 - t0, 2), device_path_total_s=round(t4 - t3, 2), device_plan_s=round(t5 - t4, 2), host_plan_s=round(t6 - t5, 2),
                          plan_tiles=[Ad.tlc_staged, Ad.tlc_tiles, A.tlc_staged, A.tlc_tiles], identical=same)), flush=True)
    if rep == 1:      # the raw-array entry point (the reference's function-pointer seam) with and without the plan cache
        from ultimate_spmv_amd import binding as B
        x = torch.rand(s.n_rows_padded, dtype=torch.float64, device="cuda"); y = torch.zeros_like(x)
        A0 = pkg.DeviceMatrix(s)
        for cache in (0, 1):
            pkg.set_tuning(raw_plan_cache=cache)
            for _ in range(5):
                pkg.uspmv_scs_gpu(32, s.n_chunks, A0.chunk_ptrs, A0.chunk_lengths, A0.col_idxs, A0.values, x, y)
            torch.cuda.synchronize(); t7 = time.time()
            for _ in range(50):
                pkg.uspmv_scs_gpu(32, s.n_chunks, A0.chunk_ptrs, A0.chunk_lengths, A0.col_idxs, A0.values, x, y)
            torch.cuda.synchronize()
            print(json.dumps(dict(raw_entry_point_plan_cache=cache, ms_per_call=round((time.time() - t7) / 50 * 1e3, 4))), flush=True)
        pkg.set_tuning(raw_plan_cache=0); pkg.lib().uspmv_raw_plan_cache_clear()
    del A, Ad, s, lay
