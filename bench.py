#!/usr/bin/env python3
"""Headline benchmark: SELL-32-512 double-precision SpMV, GFLOP/s and achieved HBM GB/s.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" is one SpMV y = A x over the whole (distributed) matrix, inputs resident in HBM.
Workload at N = 1 (BASELINE.json configs[1]): nlpkkt200-class matrix -- the SuiteSparse file is not
available offline, so the deterministic stand-in of SURVEY.md 8(d) is used: 27-point stencil on a
253^3 grid (n = 16 194 277, nnz = 4.33e8), `uspmv <mtx> scs -c 32 -s 512 -dp`.  A real .mtx can be
given with --mtx.  At N > 1 the global grid is 253 x 253 x (253*N) (weak scaling: every GPU owns an
nlpkkt200-class row block), partitioned by the reference's -seg_nnz rule, with the halo x-vector
exchange (-comm_halos 1) on RCCL every step.  --scaling strong keeps the N = 1 matrix instead.

GF/s = 2 * nnz_total / t_step / 1e9 (code/main.cpp:521-526).  Algorithmic bytes per SpMV per GPU
= n_elements*(8+4) + 8*n_chunks + 8*(n_local + n_halo) + 8*n_rows_padded (code/main.cpp:655-663).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md); 6.29 TB/s measured copy


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--grid", type=int, default=253, help="stencil grid edge (253 = nlpkkt200-class)")
    ap.add_argument("--mtx", default=None, help="MatrixMarket file instead of the synthetic matrix (N = 1)")
    ap.add_argument("-c", "--chunk", type=int, default=32)
    ap.add_argument("-s", "--sigma", type=int, default=512)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--seg", choices=["seg-nnz", "seg-rows"], default="seg-nnz")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--tune", default="", help="key=value,... forwarded to uspmv_set_tuning")
    ap.add_argument("--no-tlc", action="store_true", help="skip the tile-local-column plan (plain gather kernel)")
    return ap.parse_args()


def stencil_row_counts(nx, ny, nz):
    sx = np.full(nx, 3, np.int64); sx[0] -= 1; sx[-1] -= 1
    sy = np.full(ny, 3, np.int64); sy[0] -= 1; sy[-1] -= 1
    sz = np.full(nz, 3, np.int64); sz[0] -= 1; sz[-1] -= 1
    return (sz[:, None, None] * sy[None, :, None] * sx[None, None, :]).reshape(-1)


def usable_cores():
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def set_omp_threads(n):
    """libgomp is shared by libuspmv.so (host set-up) and the reference/oracle kernels."""
    import ctypes
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except OSError:
        pass


def cpu_baseline(scs_arrays, C, n_chunks, nnz, x, seconds):
    """Reference CPU kernel timed on the host cores (rank 0, N = 1): the genuine scs_impl_cpu<32>
    from oracle/_ref when present (kind "reference"), else the oracle's C port (kind "port")."""
    from oracle import refshim
    a = scs_arrays
    cores = usable_cores()
    set_omp_threads(cores)
    if refshim.available("colwise"):
        kind = "reference"
        L = refshim.lib("colwise")
        y = np.zeros(n_chunks * C)
        xx = np.ascontiguousarray(x)

        def run():
            L.ref_spmv_omp_scs_adv_f64(C, n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], xx, y)
    else:
        kind = "port"
        from oracle import oracle as orc

        def run():
            orc.spmv_scs(C, n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], x)
    run(); run()
    reps, t0 = 0, time.perf_counter()
    while True:
        run(); reps += 1
        el = time.perf_counter() - t0
        if el >= seconds or reps >= 1000:
            break
    return {"value": round(2.0 * nnz * reps / el / 1e9, 3), "unit": "GFLOP/s", "cores": cores, "kind": kind,
            "sample": f"whole matrix, {reps} SpMVs of spmv_omp_scs_adv<C=32,double> in {el:.1f} s, "
                      f"OMP threads = {cores}"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge

    pkg = ge.load_package()
    from ultimate_spmv_amd import binding as B
    from ultimate_spmv_amd.distributed import DistSpmv, seg_from_row_counts

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched through torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if os.environ.get("USPMV_BENCH_ONE_DEVICE"):   # rehearsal only: several ranks share GPU 0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("USPMV_BENCH_ONE_DEVICE"):
            dist.init_process_group("gloo", rank=rank, world_size=world)      # rehearsal: RCCL refuses duplicate GPUs
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    # torchrun exports OMP_NUM_THREADS=1; the host set-up (generation, conversion, planning) is OpenMP code,
    # so give every rank its share of the usable cores instead
    set_omp_threads(max(1, usable_cores() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))))
    for kv in filter(None, args.tune.split(",")):
        k, v = kv.split("=")
        pkg.set_tuning(**{k: int(v)})

    # ------------------------------------------------------------------ matrix (per-rank row block)
    t_setup = time.time()
    if args.mtx:
        assert world == 1, "--mtx is single-GPU in this round"
        loc = pkg.read_mtx(args.mtx)
        wsa = np.array([0, loc.n_rows], np.int32)
        workload = f"{os.path.basename(args.mtx)} scs -c {args.chunk} -s {args.sigma} -dp"
        total_nnz, n_global = loc.nnz, loc.n_rows
    else:
        g = args.grid
        nz = g * world if args.scaling == "weak" else g
        counts = stencil_row_counts(g, g, nz)
        n_global, total_nnz = int(counts.size), int(counts.sum())
        wsa = seg_from_row_counts(counts, args.seg, world) if world > 1 else np.array([0, n_global], np.int32)
        del counts
        loc = pkg.gen_stencil27(g, g, nz, row_begin=int(wsa[rank]), row_end=int(wsa[rank + 1]))
        workload = (f"nlpkkt200-class synthetic (27-pt stencil {g}x{g}x{nz}, n={n_global}, nnz={total_nnz}) "
                    f"scs -c {args.chunk} -s {args.sigma} -dp" + (f" -{args.seg.replace('-', '_')} -comm_halos 1" if world > 1 else ""))
    d = DistSpmv(loc, wsa, args.chunk, args.sigma, B.F64, device=dev, overlap=not args.no_overlap, tlc=not args.no_tlc)
    del loc
    x = d.new_x(np.full(d.n_local, 5.0))          # DefaultValues::x (code/classes_structs.hpp:1799-1800)
    y = d.new_y()
    s = d.scs
    bytes_local = s.n_elements * 12 + 8 * s.n_chunks + 8 * (d.n_local + d.n_halo) + 8 * s.n_rows_padded
    t_setup = time.time() - t_setup

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # ------------------------------------------------------------------ timed region
    for _ in range(args.warmup):
        d.spmv(x, y)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        d.spmv(x, y)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed / args.steps * 1e3
    gflops = 2.0 * total_nnz / (ms_per_step * 1e-3) / 1e9

    # ------------------------------------------------------------------ dominant kernel, HIP events on its stream
    reps = max(10, min(args.steps, 100))
    k_ms = B.time_launches(0, reps, A=d.A, x=x, y=y)
    achieved = bytes_local / (k_ms * 1e-3) / 1e9
    nstream = 1 << 27  # 1 GiB per array
    sa = torch.empty(nstream, dtype=torch.float64, device=dev)
    sb = torch.ones(2 * nstream, dtype=torch.float64, device=dev)
    part = torch.empty(8192, dtype=torch.float64, device=dev)
    B.time_launches(1, 3, x=sb, y=sa, n=nstream)
    copy_gbs = 16.0 * nstream / (B.time_launches(1, 20, x=sb, y=sa, n=nstream) * 1e-3) / 1e9
    triad_gbs = 24.0 * nstream / (B.time_launches(2, 20, x=sb, y=sa, n=nstream) * 1e-3) / 1e9
    read_gbs = 8.0 * nstream / (B.time_launches(3, 20, x=sb, y=part, n=nstream) * 1e-3) / 1e9
    del sa, sb

    traffic = None
    if not args.mtx:
        key = f"stencil27-{args.grid}x{args.grid}x{args.grid * world if args.scaling == 'weak' else args.grid}-c{args.chunk}-s{args.sigma}-f64-n{world}"
        try:  # HBM bytes per launch measured with rocprofv3 PMC passes of this very command (profiles/)
            traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(key, {}).get("traffic_bytes")
        except (OSError, ValueError):
            traffic = None
    out = {
        "metric": "SpMV GFLOP/s (SELL-32-512 dp; achieved HBM GB/s in roofline)",
        "value": round(gflops, 2), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic" if not args.mtx else "file",
        "config": {"workload": workload, "C": args.chunk, "sigma": args.sigma, "n_rows": n_global, "nnz": total_nnz,
                   "beta": round(s.nnz / s.n_elements, 6), "x": "5.0 (DefaultValues)",
                   "partition": args.seg if world > 1 else "none", "halo_overlap": (not args.no_overlap) and world > 1,
                   "kernel": "tile-local-column (LDS-staged x, 16-bit local indices)" if d.use_tiles or (world == 1 and d.A.tlc_staged) else "lane-per-row gather",
                   "tlc_tiles_staged": [d.A.tlc_staged, d.A.tlc_tiles],
                   "tuning": {k: pkg.get_tuning(k) for k in ("unroll", "nontemporal", "xcd_remap", "block", "spmv_variant", "tlc")}},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "kernel": "scs_spmv_tlc<double,32>" if d.A.tlc_staged and pkg.get_tuning("tlc") else "scs_spmv_rows<double,32,8>",
                     "kernel_ms": round(k_ms, 5),
                     "algorithmic_bytes_per_launch": int(bytes_local),
                     "frac_of_stream_copy": round(achieved / copy_gbs, 4),
                     "stream_same_run_GBs": {"copy": round(copy_gbs, 1), "triad": round(triad_gbs, 1), "read": round(read_gbs, 1)}},
        "setup_s": round(t_setup, 1),
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        a = s.arrays()
        out["cpu_baseline"] = cpu_baseline(a, s.C, s.n_chunks, s.nnz, x.cpu().numpy(), args.cpu_seconds)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
