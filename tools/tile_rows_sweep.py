#!/usr/bin/env python3
"""Rows per tile of the tile-local-column plan (256 | 512 | 1024) against kernel time, per matrix class -- the data behind the
rule in csrc/uspmv_api.hip (plan_tile_rows).  One line per (matrix, tile_rows): lines staged, the largest tile, kernel ms."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--only", default="")
    ap.add_argument("--tiles", default="256,512,1024")
    args = ap.parse_args()
    import torch as t
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from ultimate_spmv_amd import binding as B
    t.cuda.set_device(0)
    classes = {
        "stencil27_253": lambda: pkg.gen_stencil27(253, 253, 253),
        "kkt_200": lambda: pkg.gen_kkt(200),
        "stencil27_111_dof3": lambda: pkg.gen_stencil27(111, 111, 111, dof=3),
        "stencil27_74_dof5": lambda: pkg.gen_stencil27(74, 74, 74, dof=5),
        "stencil9_2d_4000": lambda: pkg.gen_stencil27(4000, 4000, 1),
        "stencil27_slab_1000x1000x16": lambda: pkg.gen_stencil27(1000, 1000, 16),
        "banded_4M_30_2000": lambda: pkg.gen_banded_random(4000000, 30, 2000),
        "banded_8M_12_300": lambda: pkg.gen_banded_random(8000000, 12, 300),
    }
    for name, gen in classes.items():
        if args.only and name not in args.only.split(","):
            continue
        coo = gen()
        s = pkg.convert_to_scs(coo, 32, 512, pkg.F64)
        a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"]); a = s.arrays()
        x = t.ones(s.n_rows_padded, dtype=t.float64, device="cuda"); y = t.zeros_like(x)
        byts = s.n_elements * 12 + 8 * s.n_chunks + 8 * s.n_rows + 8 * s.n_rows_padded
        y0 = None
        for R in [int(v) for v in args.tiles.split(",")]:
            pkg.set_tuning(tlc_tile_rows=R)
            A = pkg.DeviceMatrix(s, tlc=True)
            pkg.spmv(A, x, y)
            if y0 is None: y0 = y.clone()
            same = bool(t.equal(y, y0))
            ms = B.time_launches(0, args.reps, A=A, x=x, y=y)
            print(json.dumps(dict(matrix=name, n=s.n_rows, nnz=s.nnz, tile_rows=R, plan=list(A.plan_info()), staged=[A.tlc_staged, A.tlc_tiles],
                                  kernel_ms=round(ms, 5), frac=round(byts / (ms * 1e-3) / 8e12, 4), same_bits=same)), flush=True)
            del A
        pkg.set_tuning(tlc_tile_rows=0)
        del coo, s


if __name__ == "__main__":
    main()
