#!/usr/bin/env python3
"""Stand-in for the `uspmv` harness in the CPU tests of bench.py's N > 1 launcher (tests/test_bench_launcher.py): no GPU, no library.
Behaviour by $FAKE_USPMV_MODE: "ok" writes the JSON report the real harness writes (made-up numbers, real field names); "hang" sleeps until
it is killed; "fail" exits 1; "capture_crash" gets as far as the step-form stage and dies unless -graph 0; "rccl_down" fails unless
USPMV_EXCHANGE=host."""
import json
import os
import sys
import time

a = sys.argv[1:]
opt = {a[i]: a[i + 1] for i in range(2, len(a) - 1) if a[i].startswith("-")}
mode = os.environ.get("FAKE_USPMV_MODE", "ok")
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
host = os.environ.get("USPMV_EXCHANGE") == "host"


def stage(s):
    sys.stderr.write(f"[uspmv stage] {s} (rank {rank})\n")
    sys.stderr.flush()


stage("device set, waiting for the other ranks")
if mode == "hang":
    time.sleep(10000)
if mode == "fail" or (mode == "rccl_down" and world > 1 and not host):
    sys.stderr.write("ERROR: made to fail\n")
    sys.exit(1)
stage("step object created (communicator up)")
stage("first collective done")
stage("first eager step done")
stage("timing the step forms")
if mode == "capture_crash" and world > 1 and opt.get("-graph") == "1" and not host:
    os.abort()
stage("step form chosen")
stage("timed region done")
steps, warm = int(opt.get("-bench_steps", 5)), int(opt.get("-bench_warmup", 2))
g = [int(v) for v in a[0][4:].split("x")]
n = g[0] * g[1] * g[2]
if rank == 0 and "-json" in opt:
    if world == 1:
        rep = {"gflops": 1000.0, "ms_per_step": 1.6, "kernel_ms": 1.5, "steps": steps, "warmup": warm, "runtime_s": 0.01, "ranks": 1, "n_rows": n, "nnz": 27 * n,
               "n_elements": 27 * n, "n_chunks": n // 32, "n_rows_padded": n, "algorithmic_bytes": 12.0 * 27 * n, "algorithmic_GBs": 6000.0, "plan_kind": 1,
               "plan_tiles": 10, "plan_tiles_planned": 10, "setup_s": 0.1}
    else:
        per = [{"rank": r, "n_local": n // world, "n_halo": 100 + r, "n_send": 100, "interior": 5, "boundary": 3, "n_elements": 27 * n // world, "nnz": 27 * n // world,
                "algorithmic_bytes": 12 * 27 * n // world, "local_kernel_ms": 0.2 + 0.01 * r} for r in range(world)]
        rep = {"gflops": 6000.0, "ms_per_step": 0.25, "steps": steps, "warmup": warm, "runtime_s": 0.01, "ranks": world, "loopback": False,
               "exchange": "host" if host else "rccl", "n_rows": n, "nnz": 27 * n, "protocol": "fixed steps between barriers", "ba_synch": int(opt.get("-ba_synch", 0)),
               "graph_replay": opt.get("-graph") == "1", "graph_launches": steps, "eager_steps": 0, "overlap": True, "step_form": "overlap",
               "step_form_candidates_ms": {"overlap": 0.25, "plain": 0.26}, "other_ba_synch_ms_per_step": 0.27, "y_checked": True, "y_mismatches": 0,
               "y_checksum_rank0": 1.0, "rank0": {k: per[0][k] for k in ("n_local", "n_halo", "n_send", "interior", "boundary", "n_elements", "algorithmic_bytes", "local_kernel_ms")} | {"tiles": True, "n_chunks": 1, "n_rows_padded": n // world},
               "rccl_nranks": 0 if host else world, "versions": {"hip_build": 1, "hip_runtime": 1, "rccl_build": 1, "rccl_runtime": 1}, "per_rank": per}
    json.dump(rep, open(opt["-json"], "w"))
stage("report written")
