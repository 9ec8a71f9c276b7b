#!/usr/bin/env python3
"""Per-step cost of the distributed step on ONE GPU (loopback: the process plays block `rank` of a P-way seg-rows partition of the
nlpkkt240-class matrix, neighbours = itself, RCCL self send/recv): the single-launch SpMV of the block, the C++ eager step with and
without overlap, and -- through the uspmv CLI, which binds to the system RCCL -- the hipGraph replay.  One JSON line."""
import argparse
import json
import os
import re
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=304)
    ap.add_argument("--P", type=int, default=8)
    ap.add_argument("--rank", type=int, default=3)
    ap.add_argument("--steps", type=int, default=500)
    args = ap.parse_args()
    import torch
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from ultimate_spmv_amd import binding as B
    torch.cuda.set_device(0)
    g, P, rank = args.grid, args.P, args.rank
    counts = pkg.gen_stencil27_row_counts(g, g, g)
    wsa = pkg.seg_from_row_counts(counts, "seg-rows", P)
    loc = pkg.gen_stencil27(g, g, g, row_begin=int(wsa[rank]), row_end=int(wsa[rank + 1]))
    d = pkg.DistNative(loc, wsa, 32, 512, rank, P, pkg.comm_unique_id(), comm_rank=0, comm_size=1)
    x = d.new_x(np.full(d.n_local, 5.0)); y = d.new_y()
    out = dict(workload=f"block {rank} of {P} (seg-rows) of the 27-pt stencil {g}^3, loopback", n_local=d.n_local, n_halo=d.n_halo,
               interior=d.n_interior, boundary=d.n_boundary, tiles=d.use_tiles)

    def timed(fn, n):
        fn(20); d.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(n); d.synchronize(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    out["single_launch_ms"] = round(timed(lambda n: [d.spmv(x, y, comm_halos=False) for _ in range(n)], args.steps), 5)
    out["eager_overlap_ms"] = round(timed(lambda n: d.run(x, y, n, use_graph=False), args.steps), 5)
    # host cost alone: issue the steps without waiting for the device in between, measured per call on an idle device
    d.synchronize()
    t0 = time.perf_counter(); d.run(x, y, 50, use_graph=False); t1 = time.perf_counter(); d.synchronize()
    out["eager_host_issue_us_per_step"] = round((t1 - t0) / 50 * 1e6, 2)
    d.set_overlap(False)
    out["eager_no_overlap_ms"] = round(timed(lambda n: d.run(x, y, n, use_graph=False), args.steps), 5)
    d.close()
    # hipGraph replay: the CLI (system RCCL)
    env = dict(os.environ, USPMV_LOOPBACK=str(P), USPMV_LOOPBACK_RANK=str(rank), USPMV_ID_DIR="/tmp", USPMV_JOB_ID="stepcost")
    for tag, extra in (("cli_graph_ms", ["-graph", "1"]), ("cli_eager_ms", ["-graph", "0"])):
        r = subprocess.run([os.path.join(ROOT, "ultimate-spmv_amd", "uspmv"), f"gen:{g}x{g}x{g}", "scs", "-c", "32", "-s", "512", "-seg_rows", "-comm_halos", "1",
                            "-bench_time", "1"] + extra, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
        m = re.search(r"([0-9.]+) ms per SpMV", r.stdout)
        out[tag] = float(m.group(1)) if m else None
        out[tag.replace("_ms", "_mode")] = "hipGraph replay" if "hipGraph replay" in r.stdout else "eager steps"
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
