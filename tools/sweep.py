#!/usr/bin/env python3
"""Interleaved A/B timing of kernel variants on one device, one process (HIP events on the launch
stream).  python tools/sweep.py --grid 253 --rounds 5"""
import argparse
import itertools
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=253)
    ap.add_argument("--dof", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("-c", type=int, default=32)
    ap.add_argument("-s", type=int, default=512)
    ap.add_argument("--xmode", default="const")
    ap.add_argument("--variants", default="full")
    args = ap.parse_args()
    import torch
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from ultimate_spmv_amd import binding as B
    torch.cuda.set_device(0)
    t0 = time.time()
    m = pkg.gen_stencil27(args.grid, args.grid, args.grid, dof=args.dof)
    s = pkg.convert_to_scs(m, args.c, args.s)
    a = s.arrays()
    pkg.permute_scs_cols(s, a["old_to_new_idx"])
    ap.add_argument("--tile-rows", type=int, default=256) if False else None
    pkg.set_tuning(tlc_tile_rows=int(os.environ.get("TLC_TILE_ROWS", "256")))
    A = pkg.DeviceMatrix(s, tlc=True, tlc_max_lines=int(os.environ.get("TLC_MAX_LINES", "0")))
    print(f"tlc: {A.tlc_staged} of {A.tlc_tiles} tiles staged", flush=True)
    print(f"n={s.n_rows} nnz={s.nnz} n_el={s.n_elements} beta={s.nnz / s.n_elements:.4f} setup {time.time() - t0:.1f}s", flush=True)
    x = torch.full((s.n_rows_padded,), 5.0, dtype=torch.float64, device="cuda")
    if args.xmode == "rand":
        x = torch.rand(s.n_rows_padded, dtype=torch.float64, device="cuda")
    y = torch.zeros(s.n_rows_padded, dtype=torch.float64, device="cuda")
    bytes_ = s.n_elements * 12 + 8 * s.n_chunks + 8 * s.n_rows + 8 * s.n_rows_padded
    if args.variants == "full":
        var = [dict(spmv_variant=0, unroll=u, nontemporal=nt, xcd_remap=xr, block=b)
               for u, nt, xr, b in itertools.product((1, 2, 4, 8), (0, 1), (0, 1), (128, 256, 512))]
        if args.c == 32:
            var += [dict(spmv_variant=1, unroll=4, nontemporal=nt, xcd_remap=xr, block=b)
                    for nt, xr, b in itertools.product((0, 1), (0, 1), (256, 512))]
    else:
        var = [dict(kv.split("=") for kv in v.split(",")) for v in args.variants.split(";")]
        var = [{k: int(v) for k, v in d.items()} for d in var]
    res = {json.dumps(v, sort_keys=True): [] for v in var}
    for r in range(args.rounds):
        for v in var:
            pkg.set_tuning(spmv_variant=0, unroll=8, nontemporal=1, xcd_remap=0, block=256, ablate=0, tail_batch=0, tlc=0)  # keys are sticky
            pkg.set_tuning(**v)
            B.time_launches(0, 2, A=A, x=x, y=y)
            res[json.dumps(v, sort_keys=True)].append(B.time_launches(0, args.reps, A=A, x=x, y=y))
        print(f"round {r} done", flush=True)
    rows = sorted(((np.median(t), min(t), k) for k, t in res.items()))
    print(f"{'median ms':>10} {'min ms':>8} {'GB/s':>8} {'GF/s':>8}  variant")
    for med, mn, k in rows:
        print(f"{med:10.4f} {mn:8.4f} {bytes_ / med / 1e6:8.0f} {2 * s.nnz / med / 1e6:8.0f}  {k}")
    n = 1 << 27
    sa = torch.empty(n, dtype=torch.float64, device="cuda"); sb = torch.ones(2 * n, dtype=torch.float64, device="cuda")
    part = torch.empty(8192, dtype=torch.float64, device="cuda")
    for what, nm, bpe in ((1, "copy", 16), (2, "triad", 24), (3, "read", 8)):
        B.time_launches(what, 3, x=sb, y=sa if what != 3 else part, n=n)
        ms = B.time_launches(what, 20, x=sb, y=sa if what != 3 else part, n=n)
        print(f"stream {nm}: {bpe * n / ms / 1e6:.0f} GB/s")


if __name__ == "__main__":
    main()
