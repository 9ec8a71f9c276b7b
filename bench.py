#!/usr/bin/env python3
"""Headline benchmark: SELL-32-512 double-precision SpMV, GFLOP/s and achieved HBM GB/s.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" is one SpMV y = A x over the whole (distributed) matrix, inputs resident in HBM.
Workload at N = 1 (BASELINE.json configs[1]): nlpkkt200-class matrix -- the SuiteSparse file is not
available offline, so the deterministic stand-in of SURVEY.md 8(d) is used: 27-point stencil on a
253^3 grid (n = 16 194 277, nnz = 4.33e8), `uspmv <mtx> scs -c 32 -s 512 -dp`.  A real .mtx can be
given with --mtx.
Workload at N > 1 (BASELINE.json configs[4]): ONE nlpkkt240-class matrix (27-point stencil on 304^3,
n = 28 094 464, nnz = 7.5e8) split over the N GPUs by the reference's -seg_nnz rule -- STRONG scaling --
with the halo x-vector exchange (-comm_halos 1) on RCCL every step.  The step runs in C++ behind the C ABI
(uspmv_dist_run: pack kernel, grouped ncclSend/ncclRecv on a side stream, interior tiles, boundary tiles, replayed from ONE
captured hipGraph).  Every torchrun rank starts the `uspmv` harness as a CHILD PROCESS before it touches the GPU
(`uspmv gen:... scs -c 32 -s 512 -dp -seg_nnz -comm_halos 1 -bench_steps K -bench_warmup W -check_y 1 -json ...`): the
children bind to the system's RCCL / HIP runtime, where graph capture of the step works (inside a torch process the library
binds to torch's older bundled RCCL), time exactly K steps between barriers, take the slowest rank's clock, and then check
y of every local row bitwise against the rows' entry-ordered FMA chains ("y_checked").  If the children fail, all ranks fall
back to the same C++ step inside this process (eager under torch's RCCL, also y-checked) and "config.step" / "fallback_reason"
say so; if that fails too, the torch.distributed twin of round 1 runs, "value" is null and its number sits in "fallback_value".
A weak-scaling run (every GPU owns an nlpkkt200-class block, grid 253 x 253 x 253*N) is measured after it and reported
inside the same JSON line under "weak_scaling"; --scaling weak makes that the headline instead.

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
    ap.add_argument("--grid", type=int, default=0, help="stencil grid edge (default: 253 = nlpkkt200-class at N = 1 and for weak scaling, 304 = nlpkkt240-class for strong scaling at N > 1)")
    ap.add_argument("--mtx", default=None, help="MatrixMarket file instead of the synthetic matrix (N = 1)")
    ap.add_argument("-c", "--chunk", type=int, default=32)
    ap.add_argument("-s", "--sigma", type=int, default=512)
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None, help="N > 1: strong (default; BASELINE config 5) or weak")
    ap.add_argument("--grid2", type=int, default=0, help="N > 1: grid edge of the second (other scaling mode) measurement (default 253 weak / 304 strong)")
    ap.add_argument("--graph", action="store_true", help="N > 1: replay the C++ step from a hipGraph (crashes in hipStreamEndCapture under torch's bundled RCCL on this image; the uspmv CLI, on the system RCCL, replays graphs)")
    ap.add_argument("--no-second-line", action="store_true", help="N > 1: skip the other scaling mode's measurement")
    ap.add_argument("--in-process", action="store_true", help="N > 1: skip the uspmv child processes, run the C++ step inside this (torch) process")
    ap.add_argument("--ba-synch", type=int, default=0, choices=[0, 1], help="N > 1: per-step barrier of the headline protocol (the reference's default is 1; the other setting is timed too and reported)")
    ap.add_argument("--budget-s", type=float, default=480.0, help="N > 1: wall-clock budget of the WHOLE run (single-GPU reference, the tiers of uspmv children, the second scaling mode); when it ends the line is printed with value null and the reason")
    ap.add_argument("--no-single-gpu-reference", action="store_true", help="N > 1, strong scaling: skip the single-rank child that times the same matrix on one GPU")
    ap.add_argument("--step-form", default="auto", choices=["auto", "auto_all", "overlap", "plain", "pad", "fused"], help="N > 1: -step_form of the uspmv children (auto: overlap | plain timed on the machine, the faster kept)")
    ap.add_argument("--python-step", action="store_true", help="N > 1: round-1 path (torch.distributed all_to_all per step) instead of the C++ step")
    ap.add_argument("--seg", choices=["seg-nnz", "seg-rows"], default="seg-nnz")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="N > 1: eager C++ steps in the uspmv children")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vendor-baseline", action="store_true", help="N = 1: skip the rocSPARSE child process")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--tune", default="", help="key=value,... forwarded to uspmv_set_tuning")
    ap.add_argument("--no-tlc", action="store_true", help="skip the tile-local-column plan (plain gather kernel)")
    ap.add_argument("--no-traffic", action="store_true", help="N = 1: do not run the two rocprofv3 PMC child passes (roofline.traffic then comes from profiles/traffic.json)")
    ap.add_argument("--other-configs", default="3,4b,2k,3s,5one", help="N = 1: further configurations measured after the headline and reported under \"other_configs\" (3 = Queen_4147-class SpMMV b = 8, both layouts; 4b = HV15R-class ap[dp_sp] and dp; 2k = nlpkkt200-class KKT matrix; 3s = the Queen-class stencil in a numbering scrambled inside blocks of 20 000 nodes, SpMV; 5one = config 5's 304^3 matrix on one GPU); \"\" = none")
    ap.add_argument("--kkt", type=int, default=200, help="config 2k: grid edge N of uspmv_gen_kkt")
    ap.add_argument("--grid5", type=int, default=304, help="config 5one: stencil grid edge")
    ap.add_argument("--grid3", type=int, default=111, help="config 3: nodes per edge (3 dof per node)")
    ap.add_argument("--n4b", type=int, default=2017169, help="config 4b: rows")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
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


def cpu_baseline(scs_arrays, C, n_chunks, nnz, x, seconds, x_check=None, y_gpu_check=None):
    """Reference CPU kernel timed on the host cores (rank 0, N = 1): the genuine scs_impl_cpu<32>
    from oracle/_ref when present (kind "reference"), else the oracle's C port (kind "port").
    x_check / y_gpu_check: a non-uniform x and the GPU's y for it -- the CPU kernel's y for the same x is compared bitwise (the checker's
    role; returned as the second value, None when not asked for)."""
    from oracle import refshim
    a = scs_arrays
    cores = usable_cores()
    set_omp_threads(cores)
    same = None
    if refshim.available("colwise"):
        kind = "reference"
        L = refshim.lib("colwise")
        y = np.zeros(n_chunks * C)
        xx = np.ascontiguousarray(x)

        def run(xv=xx):
            L.ref_spmv_omp_scs_adv_f64(C, n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], xv, y)
            return y
    else:
        kind = "port"
        from oracle import oracle as orc

        def run(xv=x):
            return orc.spmv_scs(C, n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], xv)
    if x_check is not None:
        same = bool(np.array_equal(np.asarray(run(np.ascontiguousarray(x_check))), y_gpu_check))
    run(); run()
    reps, t0 = 0, time.perf_counter()
    while True:
        run(); reps += 1
        el = time.perf_counter() - t0
        if el >= seconds or reps >= 1000:
            break
    return {"value": round(2.0 * nnz * reps / el / 1e9, 3), "unit": "GFLOP/s", "cores": cores, "kind": kind,
            "sample": f"whole matrix, {reps} SpMVs of spmv_omp_scs_adv<C=32,double> in {el:.1f} s, "
                      f"OMP threads = {cores}"}, same


def vendor_baseline(grid):
    """rocSPARSE on the same matrix (CSR default / adaptive / rowsplit / LRB and sliced ELL with slice 32), run by tools/rocsparse_baseline in
    a child process: the ROCm twin of the reference's optional cuSPARSE path (USE_CUSPARSE: cusparseCreateCsr / cusparseCreateSlicedEll +
    cusparseSpMV, code/utilities.hpp:3380-3550, code/classes_structs.hpp:998-1011).  A reported baseline, like cpu_baseline."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "rocsparse_baseline")
    if not os.path.exists(exe):
        return {"error": "tools/rocsparse_baseline is not built (rocSPARSE missing at build time)"}
    try:
        r = subprocess.run([exe, str(grid), "1", "--json"], capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": f"rc={r.returncode}: {(r.stdout + r.stderr)[-300:]}"}
        d = json.loads(line[-1])
    except (OSError, subprocess.TimeoutExpired, ValueError) as e:
        return {"error": f"{type(e).__name__}: {e}"}
    best = min(d["ms"], key=d["ms"].get)
    sell = d["ms"].get("rocsparse sliced-ELL (32)")
    return {"library": d["library"], "value": round(2.0 * d["nnz"] / d["ms"][best] / 1e6, 1), "unit": "GFLOP/s", "best": best, "ms": d["ms"],
            "sliced_ell_GFLOPs": round(2.0 * d["nnz"] / sell / 1e6, 1) if sell else None, "sliced_ell": d.get("sliced_ell"),
            "note": "same matrix and x; CSR from the COO, sliced ELL = the SELL-32-512 arrays (slice 32)"}


def _cpu_leg(flops, fn_ref, fn_port, y_cpu, y_gpu, seconds, name):
    """cpu_baseline leg of one configuration: the reference CPU kernel (oracle/_ref when present, else the oracle's C port) timed
    for about `seconds` on the usable host cores; its y doubles as the checker of the GPU's y (bitwise)."""
    from oracle import refshim
    cores = usable_cores()
    set_omp_threads(cores)
    ref = refshim.available("colwise")
    run = fn_ref if ref else fn_port
    out = run()
    if isinstance(out, np.ndarray):       # (the port returns y, the reference writes into y_cpu)
        y_cpu = out
    same = bool(np.array_equal(np.asarray(y_cpu), y_gpu))
    reps, t0 = 0, time.perf_counter()
    while True:
        run(); reps += 1
        el = time.perf_counter() - t0
        if el >= seconds or reps >= 200:
            break
    return ({"value": round(flops * reps / el / 1e9, 2), "unit": "GFLOP/s", "cores": cores, "kind": "reference" if ref else "port",
             "sample": f"whole matrix, {reps} calls of {name} in {el:.1f} s, OMP threads = {cores}"}, same)


def other_configs(pkg, B, torch, args, which):
    """BASELINE configs 3 (Queen_4147-class, -block_vec_size 8, both block-vector layouts) and 4b (HV15R-class banded-random matrix of
    SURVEY 8(d), ap[dp_sp] and plain dp), plus 2k (the KKT-structured member of the nlpkkt200 class) and 5one (config 5's 304^3 matrix on
    ONE GPU: the strong-scaling denominator) under the driver's clock.  Per configuration: set-up, W warm-up + K wall-clock steps like the
    headline's ("ms_per_step"), the kernel's average over the same number of launches by HIP events on its stream ("kernel_ms"), the
    algorithmic-byte roofline of SURVEY 8(d) and the reference CPU kernel of the same configuration, whose result is compared bitwise
    with the GPU's."""
    from oracle import refshim
    from oracle import oracle as orc       # cpu_baseline leg: the port stands in when oracle/_ref did not travel
    t = torch
    K, W = max(args.steps, 50), 30         # (short loops behind seconds of host work measure the GPU's clock ramp)
    res = []

    def measure(call, kind, **kw):
        """(wall-clock ms per step over K steps after W warm-ups, HIP-event ms per launch over K launches)"""
        for _ in range(W):
            call()
        t.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            call()
        t.cuda.synchronize()
        wall = (time.perf_counter() - t0) / K * 1e3
        return wall, B.time_launches(kind, K, **kw)

    def line(config, workload, kernel, wall_ms, k_ms, byts, flops, same, cpu, setup, extra=None):
        d = {"config": config, "workload": workload, "kernel": kernel, "kernel_ms": round(k_ms, 5), "ms_per_step": round(wall_ms, 5), "steps": K, "warmup": W,
             "value": round(flops / wall_ms / 1e6, 1), "unit": "GFLOP/s",
             "roofline": {"bound": "hbm", "achieved": round(byts / k_ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(byts / k_ms / 1e6 / HBM_PEAK_GBS, 4), "algorithmic_bytes": int(byts)},
             "bitexact_vs_reference_cpu": same, "cpu_baseline": cpu, "setup_s": round(setup, 1)}
        if extra:
            d.update(extra)
        res.append(d)

    def ramp(s, a):
        xp = np.zeros(s.n_rows_padded)
        xp[:s.n_rows] = pkg.apply_permutation(1.0 + 1e-3 * (np.arange(s.n_rows) % 1000), a["new_to_old_idx"])
        return xp

    def spmv_config(config, workload, coo):
        """one struct, SELL-32-512 dp, default plan: uspmv_spmv"""
        t0 = time.time()
        s = pkg.convert_to_scs(coo, 32, 512, B.F64)
        a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"]); a = s.arrays()
        A = pkg.DeviceMatrix(s, tlc=True)
        xp = ramp(s, a)
        x = t.from_numpy(xp).cuda(); y = t.zeros(s.n_rows_padded, dtype=t.float64, device="cuda")
        pkg.spmv(A, x, y)
        t.cuda.synchronize()
        setup = time.time() - t0
        wall, k_ms = measure(lambda: pkg.spmv(A, x, y), 0, A=A, x=x, y=y)
        byts = s.n_elements * 12 + 8 * s.n_chunks + 8 * (s.n_rows + s.n_rows_padded)
        yc = np.zeros(s.n_rows_padded)
        fr = (lambda: refshim.lib("colwise").ref_spmv_omp_scs_adv_f64(32, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], xp, yc)) if refshim.available("colwise") else None
        fp = lambda: orc.spmv_scs(32, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], xp)
        cpu, same = _cpu_leg(2.0 * s.nnz, fr or fp, fp, yc, y.cpu().numpy(), args.cpu_seconds / 3, "spmv_omp_scs_adv<C=32,double>")
        kind, ntile, nplan = A.plan_info()
        line(config, workload + " scs -c 32 -s 512 -dp", {2: "scs_spmv_sweep<double> (column-window sweep)", 1: "scs_spmv_tlc<double,32>", 0: "scs_spmv_rows<double,32,8>"}[kind],
             wall, k_ms, byts, 2.0 * s.nnz, same, cpu, setup, {"plan_kind": kind, "plan_tiles_planned": [nplan, ntile], "rows_per_tile": (s.n_rows_padded // ntile) if ntile else None, "local_index_bits": A.index_bits() if kind == 1 else None,
              "x_elements_per_list_entry": A.plan_granularity() if kind == 1 else None, "rows_dealt_by_the_matrix_graph": A.plan_rows_dealt() if kind == 1 else None})
        del A, s, a, x, y
        t.cuda.empty_cache()

    if "3" in which:
        t0 = time.time()
        g = args.grid3
        coo = pkg.gen_stencil27(g, g, g, dof=3)
        s = pkg.convert_to_scs(coo, 32, 512, B.F64)
        a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"]); a = s.arrays()
        nnz = s.nnz
        del coo
        b, ld = 8, s.n_rows_padded
        A = pkg.DeviceMatrix(s, block_tlc=b)
        xp = ramp(s, a)
        byts = s.n_elements * 12 + 8 * s.n_chunks + b * 8 * s.n_rows + b * 8 * s.n_rows_padded
        setup = time.time() - t0
        for lay, nm in ((B.COLWISE, "colwise"), (B.ROWWISE, "rowwise")):
            X = np.zeros(b * ld)
            for v in range(b):
                col = xp * (1.0 + v / 8.0)
                if lay == B.ROWWISE:
                    X[v::b] = col
                else:
                    X[v * ld:(v + 1) * ld] = col
            dX = t.from_numpy(X).cuda(); dY = t.zeros(b * ld, dtype=t.float64, device="cuda")
            pkg.spmmv(A, dX, dY, b, ld, lay)
            t.cuda.synchronize()
            wall, k_ms = measure(lambda: pkg.spmmv(A, dX, dY, b, ld, lay), 5, A=A, x=dX, y=dY, b=b, ld=ld, layout=lay)
            Yc = np.zeros(b * ld)
            var = "rowwise" if lay == B.ROWWISE else "colwise"
            fr = (lambda: refshim.lib(var).ref_block_spmv_omp_scs_general_f64(32, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], X, Yc, b, ld)) if refshim.available(var) else None
            fp = lambda: orc.spmmv_scs(32, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], X, b, ld, lay == B.ROWWISE)
            cpu, same = _cpu_leg(2.0 * nnz * b, fr or fp, fp, Yc, dY.cpu().numpy(), args.cpu_seconds / 3, "block_spmv_omp_scs_general (" + nm + ")")
            kind = A.plan_info()[0]
            line(f"3 ({nm})", f"Queen_4147-class synthetic (27-pt stencil {g}^3 x 3 dof, n={s.n_rows}, nnz={nnz}) scs -c 32 -s 512 -dp -block_vec_size 8, {nm} X / Y",
                 "uspmv_spmmv: scs_spmmv_quadph<double,8> (phased block plan)" + (" behind block_vector_to_rowmajor" if lay == B.COLWISE else ""),
                 wall, k_ms, byts, 2.0 * nnz * b, same, cpu, setup, {"block_plan_tiles": [A.block_staged, A.block_tiles], "plan_kind": kind})
            if lay == B.COLWISE:
                # the reference's bench loop multiplies the SAME X every iteration (code/main.cpp:458-519): re-laid out once
                pkg.spmmv_x_prepared(A, dX, b, ld)
                dY.zero_()
                wall2, k2 = measure(lambda: pkg.spmmv(A, dX, dY, b, ld, lay), 5, A=A, x=dX, y=dY, b=b, ld=ld, layout=lay)
                same2 = bool(np.array_equal(dY.cpu().numpy(), Yc))
                pkg.spmmv_x_release(A)
                line("3 (colwise, X prepared)", f"the same, X unchanged between the calls (uspmv_spmmv_x_prepared: the column-major X re-laid out once, not per call)",
                     "uspmv_spmmv: scs_spmmv_quadph<double,8> (phased block plan), column-major Y", wall2, k2, byts, 2.0 * nnz * b, same2,
                     {"note": "see 3 (colwise)"}, setup, {"plan_kind": kind})
            del dX, dY
        del A, s, a
        t.cuda.empty_cache()
    if "4b" in which:
        t0 = time.time()
        coo = pkg.gen_banded_random(args.n4b, 140, 50000, magnitude_decades=10.0)
        dp, sp = pkg.partition_precisions(coo, 1e-3)
        ds = pkg.convert_to_scs(dp, 32, 512, B.F64)
        perm = ds.arrays()["old_to_new_idx"].copy()
        ss = pkg.convert_to_scs(sp, 32, 512, B.F32, fixed_permutation=perm)
        pkg.permute_scs_cols(ds, perm); pkg.permute_scs_cols(ss, perm)
        da, sa = ds.arrays(), ss.arrays()
        Ad, As = pkg.DeviceMatrix(ds), pkg.DeviceMatrix(ss)
        nt_, ns_ = pkg.optimize_ap(Ad, As, ds, ss)
        xp = ramp(ds, da)
        x = t.from_numpy(xp).cuda(); y = t.zeros(ds.n_rows_padded, dtype=t.float64, device="cuda")
        pkg.spmv_ap(Ad, As, x, y)
        t.cuda.synchronize()
        setup = time.time() - t0
        wall, k_ms = measure(lambda: pkg.spmv_ap(Ad, As, x, y), 4, A=Ad, B=As, x=x, y=y)
        byts = 12 * ds.n_elements + 8 * ss.n_elements + 16 * ds.n_chunks + 8 * (ds.n_rows + ds.n_rows_padded)
        ycpu = np.zeros(ds.n_rows_padded); yspc = np.zeros(ds.n_rows_padded, np.float32); xspc = xp.astype(np.float32)
        fr = (lambda: refshim.lib("colwise").ref_spmv_omp_scs_ap_adv(32, ds.n_chunks, da["chunk_ptrs"], da["chunk_lengths"], da["col_idxs"], da["values"], xp, ycpu,
                                                                     sa["chunk_ptrs"], sa["chunk_lengths"], sa["col_idxs"], sa["values"], xspc, yspc)) if refshim.available("colwise") else None
        fp = lambda: orc.spmv_scs_ap_adv(32, ds.n_chunks, (da["chunk_ptrs"], da["chunk_lengths"], da["col_idxs"], da["values"]),
                                         (sa["chunk_ptrs"], sa["chunk_lengths"], sa["col_idxs"], sa["values"]), xp)
        cpu, same = _cpu_leg(2.0 * coo.nnz, fr or fp, fp, ycpu, y.cpu().numpy(), args.cpu_seconds / 3, "spmv_omp_scs_ap_adv<C=32>")
        kind, ntile, nplan = Ad.plan_info()
        wl = f"HV15R-class synthetic (banded-random n={coo.n_rows}, 140 entries per row within +-50000, |a_ij| log-uniform over 10 decades, nnz={coo.nnz})"
        line("4b (ap[dp_sp])", wl + " scs -c 32 -s 512 -ap[dp_sp] -ap_threshold_1 1e-3",
             {2: "scs_spmv_sweep<double,AP> (column-window sweep)", 1: "scs_spmv_ap_tlc", 0: "scs_spmv_ap_rows"}[kind], wall, k_ms, byts, 2.0 * coo.nnz, same, cpu, setup,
             {"dp_nnz": dp.nnz, "sp_nnz": sp.nnz, "dp_elements": ds.n_elements, "sp_elements": ss.n_elements, "plan_kind": kind, "plan_tiles_planned": [nplan, ntile]})
        del Ad, As, ds, ss, da, sa, dp, sp, x, y
        t.cuda.empty_cache()
        spmv_config("4b (dp)", wl, coo)
        del coo
    if "2k" in which:
        N = args.kkt
        coo = pkg.gen_kkt(N)
        spmv_config("2k", f"nlpkkt200-class KKT [H A^T; A 0] synthetic (uspmv_gen_kkt N={N}: n={coo.n_rows}, nnz={coo.nnz}, rows of 5-28 entries in two index ranges N^3 apart)", coo)
        del coo
    if "3s" in which:
        # the Queen_4147-class stencil (80^3 nodes x 3 dof) with its NODES renumbered at random inside consecutive blocks of 20 000 nodes, rows and columns
        # alike: a numbering that is only locally coherent.  SpMV: the planner deals the rows to the tiles by the matrix graph and lists
        # single x elements.  A failure here costs this line only.
        try:
            g3s, dof, Kb = 80, 3, 20000
            base = pkg.gen_stencil27(g3s, g3s, g3s, dof=dof)
            I0, J0, V0 = (np.array(v) for v in base.arrays())
            nrow = base.n_rows
            del base
            rng = np.random.default_rng(7)
            pn = np.arange(nrow // dof, dtype=np.int64)
            for s0 in range(0, pn.size, Kb):
                seg = pn[s0:s0 + Kb].copy(); rng.shuffle(seg); pn[s0:s0 + Kb] = seg
            I1 = (pn[I0 // dof] * dof + I0 % dof).astype(np.int32); J1 = (pn[J0 // dof] * dof + J0 % dof).astype(np.int32)
            o = np.lexsort((J1, I1))
            coo = pkg.Coo.from_arrays(nrow, nrow, I1[o], J1[o], V0[o])
            del I0, J0, V0, I1, J1, o
            spmv_config("3s", f"Queen_4147-class stencil ({g3s}^3 nodes x {dof} dof, n={coo.n_rows}, nnz={coo.nnz}) with its nodes renumbered at random inside blocks of {Kb} nodes (rows and columns alike), single vector", coo)
            del coo
        except Exception as e:   # noqa: BLE001 -- an informational line must not take the others with it
            res.append({"config": "3s", "error": f"{type(e).__name__}: {e}"})
    if "5one" in which:
        g = args.grid5
        coo = pkg.gen_stencil27(g, g, g)
        spmv_config("5one", f"nlpkkt240-class synthetic (27-pt stencil {g}^3, n={coo.n_rows}, nnz={coo.nnz}) on ONE GPU (config 5's matrix: the strong-scaling denominator)", coo)
        del coo
    return res


def pmc_child(args):
    """Child of measure_traffic(): the N = 1 workload, a few SpMV launches and a 1-GiB read for the FETCH_SIZE calibration."""
    import torch
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from ultimate_spmv_amd import binding as B
    torch.cuda.set_device(0)
    g = args.grid or 253
    coo = pkg.gen_stencil27(g, g, g)
    s = pkg.convert_to_scs(coo, args.chunk, args.sigma, B.F64)
    a = s.arrays()
    pkg.permute_scs_cols(s, a["old_to_new_idx"])
    A = pkg.DeviceMatrix(s, tlc=not args.no_tlc)
    x = torch.full((s.n_rows_padded,), 5.0, dtype=torch.float64, device="cuda")
    y = torch.zeros(s.n_rows_padded, dtype=torch.float64, device="cuda")
    B.time_launches(0, 4, A=A, x=x, y=y)
    n = 1 << 27
    sb = torch.ones(n, dtype=torch.float64, device="cuda")
    part = torch.empty(8192, dtype=torch.float64, device="cuda")
    B.time_launches(3, 4, x=sb, y=part, n=n)
    torch.cuda.synchronize()


def measure_traffic(args):
    """HBM bytes per launch of the dominant kernel, measured in THIS run: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
    kernel trace only) over a child process that repeats the workload, started before this process touches the GPU.
    FETCH_SIZE is calibrated on a 1-GiB streaming read inside the same pass (gfx950 tallies 128-byte requests as 64 bytes,
    /opt/skills/guides/MI355X_MICROARCH.md "HBM").  Returns (bytes, note) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    out = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", td, "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child", "--grid", str(args.grid or 253), "-c", str(args.chunk), "-s", str(args.sigma)] + (["--no-tlc"] if args.no_tlc else [])
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=420)
            except (OSError, subprocess.TimeoutExpired) as e:
                return None, f"rocprofv3 pass {counter} failed: {e}"
            vals = {}
            for f in glob.glob(os.path.join(td, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") != counter:
                        continue
                    k = row["Kernel_Name"]
                    key = "spmv" if "scs_spmv_" in k else "read" if "stream_read" in k else None
                    if key:
                        vals.setdefault(key, []).append(float(row["Counter_Value"]))
            if "spmv" not in vals:
                return None, f"rocprofv3 pass {counter}: no SpMV dispatch in the counter file (rc={r.returncode})"
            out[counter] = {k: sum(v) / len(v) for k, v in vals.items()}
    fetch_kib, write_kib = out["FETCH_SIZE"]["spmv"], out["WRITE_SIZE"]["spmv"]
    cal = (1 << 20) / out["FETCH_SIZE"]["read"] if out["FETCH_SIZE"].get("read") else 2.0     # 1 GiB read / KiB reported
    traffic = int(fetch_kib * 1024 * cal + write_kib * 1024)
    return traffic, (f"measured in this run: rocprofv3 --pmc FETCH_SIZE {fetch_kib:.0f} KiB x {cal:.3f} (calibrated on a 1-GiB read in the same pass) "
                     f"+ --pmc WRITE_SIZE {write_kib:.0f} KiB, separate passes, kernel trace only")


def gpu_state(card=None):
    """Clocks, power and temperatures of the GPU as the amdgpu driver publishes them in sysfs (no HIP call, nothing is changed): the level
    of every pp_dpm_* clock that is active right now, the hwmon sensors (power, power cap, temperatures incl. the HBM's, frequencies).
    `card`: /sys/class/drm/cardN/device; default: the first card that has pp_dpm_sclk."""
    import glob
    if card is None:
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))
        if not cards:
            return {"error": "no amdgpu sysfs (pp_dpm_sclk) visible"}
        card = os.path.dirname(cards[0])
    out = {"card": card.split("/")[4], "pci": os.path.basename(os.path.realpath(card))}
    for f in sorted(glob.glob(os.path.join(card, "pp_dpm_*"))):
        try:
            lines = [ln.strip() for ln in open(f).read().splitlines() if ln.strip()]
        except OSError:
            continue
        act = [ln for ln in lines if ln.endswith("*")]
        out[os.path.basename(f)[7:]] = {"active": act[0].rstrip("* ").split(": ")[-1] if act else None, "top": lines[-1].rstrip("* ").split(": ")[-1] if lines else None}
    for f in sorted(glob.glob(os.path.join(card, "hwmon", "hwmon*", "*"))):
        name = os.path.basename(f)
        if not (name.endswith("_input") or name.endswith("_average") or name.endswith("_cap") or name.endswith("_label") or name.endswith("_cap_max")):
            continue
        try:
            v = open(f).read().strip()
            out["hwmon_" + name] = int(v) if v.lstrip("-").isdigit() else v
        except (OSError, ValueError):
            pass
    return out


def card_of_device(torch, idx):
    """/sys/class/drm/cardN/device of HIP device `idx` (by PCI address; a box shows the sysfs of every GPU of its host), or None"""
    import glob
    try:
        p = torch.cuda.get_device_properties(idx)
        bus = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        c = sorted(glob.glob(f"/sys/bus/pci/devices/{bus}/drm/card[0-9]*"))
        return os.path.join("/sys/class/drm", os.path.basename(c[0]), "device") if c else None
    except (AttributeError, RuntimeError, OSError):
        return None


class GpuStateSampler:
    """Samples gpu_state() every `period` seconds in a thread while a measurement runs (sysfs reads only)."""

    def __init__(self, card=None, period=0.02):
        import threading
        self.card = card
        self.period, self.samples, self._stop = period, [], threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            self.samples.append(gpu_state(self.card))
            self._stop.wait(self.period)

    def __enter__(self):
        self._t.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._t.join(timeout=2)
        return False

    def summary(self):
        """per key: the distinct values seen (clock levels) or min / max (sensors)"""
        agg = {}
        for smp in self.samples:
            for k, v in smp.items():
                if isinstance(v, dict):
                    agg.setdefault(k, set()).add(v.get("active"))
                elif isinstance(v, int):
                    lo, hi = agg.get(k, (v, v))
                    agg[k] = (min(lo, v), max(hi, v))
        return {"samples": len(self.samples), **{k: (sorted(x for x in v if x) if isinstance(v, set) else list(v)) for k, v in agg.items()}}


def stream_rates(B, torch, dev, x_elems=0, plane=0, line=0, rows=256):
    n = 1 << 27  # 1 GiB per array
    sa = torch.empty(n, dtype=torch.float64, device=dev)
    sb = torch.ones(2 * n, dtype=torch.float64, device=dev)
    part = torch.empty(8192, dtype=torch.float64, device=dev)
    B.time_launches(1, 3, x=sb, y=sa, n=n)
    copy = 16.0 * n / (B.time_launches(1, 20, x=sb, y=sa, n=n) * 1e-3) / 1e9
    triad = 24.0 * n / (B.time_launches(2, 20, x=sb, y=sa, n=n) * 1e-3) / 1e9
    read = 8.0 * n / (B.time_launches(3, 20, x=sb, y=part, n=n) * 1e-3) / 1e9
    out = {"copy": round(copy, 1), "triad": round(triad, 1), "read": round(read, 1)}
    if x_elems >= 4096:
        # second yardstick: the x traffic of the tile-local-column kernel without its matrix stream -- one workgroup per 256-row tile, mapped
        # to the XCDs like the kernel's tiles, reads the nine runs of x lines its rows touch (uspmv_stream_gather_lines): a vector of
        # x_elems doubles, every line wanted by ~9 workgroups, ~3 of them far apart in launch order.  L2 / fabric, not HBM streaming.
        tiles = (x_elems + rows - 1) // rows
        part2 = torch.empty(4 * tiles, dtype=torch.float64, device=dev)
        gbytes = tiles * 9 * (((rows + 2 + 15) // 16 + 1) * 128)
        B.time_launches(6, 3, x=sb, y=part2, n=x_elems, ld=plane, b=line, layout=rows)
        ms = B.time_launches(6, 50, x=sb, y=part2, n=x_elems, ld=plane, b=line, layout=rows)
        out["x_line_gather"] = round(gbytes / (ms * 1e-3) / 1e9, 1)
        out["x_line_gather_note"] = (f"{tiles} workgroups x 9 runs of x lines ({gbytes / 1e9:.2f} GB asked for, vector of {8 * x_elems / 1e6:.0f} MB, plane stride {plane}, "
                                     f"line stride {line}) in {ms:.4f} ms")
    return out


class stdout_to_stderr:
    """RCCL prints a version banner on STDOUT when a communicator is created; the contract is ONE JSON line there, so file descriptor 1
    points at stderr while communicators come up."""
    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


class Parents:
    """Agreement between the bench.py processes of one N > 1 run.  Self-launched (`python bench.py --gpus N`, no outer launcher): ONE
    process, which starts all N `uspmv` rank children itself.  Under torch.distributed.run: N processes, each starting the child of its
    own rank; they agree on every verdict through the library's host communicator (no torch, no GPU)."""

    def __init__(self, pkg, rank, world, self_launch, timeout_s):
        self.rank, self.world, self.self_launch = rank, world, self_launch
        self.hc = None if self_launch else pkg.HostComm(f"benchparents{os.environ.get('MASTER_PORT', '0')}", rank, world, timeout_s=timeout_s)

    def my_ranks(self):
        return list(range(self.world)) if self.self_launch else [self.rank]

    def allgather(self, v):
        return [int(v)] if self.hc is None else self.hc.allgather(np.array([int(v)], np.int64)).ravel().tolist()

    def close(self):
        if self.hc is not None:
            self.hc.barrier()
            self.hc.close()
            self.hc = None


def _last_stage(text):
    st = [ln.split("] ", 1)[1] for ln in text.splitlines() if ln.startswith("[uspmv stage] ")]
    return st[-1] if st else "no stage reached"


def run_children(cmd_of_rank, env_of_rank, ranks, timeout_s, tmp):
    """Start one `uspmv` process per rank in `ranks`, wait until all have ended or `timeout_s` has passed, then terminate exactly the
    processes started here that are still alive (their own sessions: SIGTERM, then SIGKILL).  Returns {rank: (rc, output tail, last stage)};
    rc -9 = killed at the time limit."""
    import signal
    import subprocess
    procs, logs = {}, {}
    for r in ranks:
        logs[r] = open(os.path.join(tmp, f"rank{r}.log"), "w+")
        try:
            procs[r] = subprocess.Popen(cmd_of_rank(r), cwd=tmp, env=env_of_rank(r), stdout=logs[r], stderr=subprocess.STDOUT, start_new_session=True)
        except OSError as e:
            logs[r].write(f"cannot start: {e}\n")
            procs[r] = None
    t_end = time.time() + timeout_s
    failed_at = None
    while True:
        alive = [r for r, p in procs.items() if p is not None and p.poll() is None]
        if not alive:
            break
        bad = [r for r, p in procs.items() if p is None or (p.poll() is not None and p.returncode != 0)]
        if bad and failed_at is None:
            failed_at = time.time()        # a rank has failed: its peers notice through the communicator's failure flag; give them 20 s
        if time.time() >= t_end or (failed_at is not None and time.time() - failed_at > 20):
            for r in alive:
                try:
                    os.killpg(procs[r].pid, signal.SIGTERM)
                except OSError:
                    pass
            time.sleep(3)
            for r in alive:
                if procs[r].poll() is None:
                    try:
                        os.killpg(procs[r].pid, signal.SIGKILL)
                    except OSError:
                        pass
                    procs[r].wait()
                    procs[r].returncode = -9
            break
        time.sleep(0.2)
    out = {}
    for r in ranks:
        logs[r].flush(); logs[r].seek(0)
        text = logs[r].read()
        logs[r].close()
        rc = -2 if procs[r] is None else procs[r].returncode
        if rc is not None and rc < 0 and rc != -2:
            rc = -9
        out[r] = (rc, text[-1500:], _last_stage(text))
    return out


def cli_measure(args, parents, scaling, grid, timeout_s, host_exchange=False, eager=False):
    """One N > 1 measurement by the `uspmv` harness: one child process per rank (the bench.py processes never touch the GPU), rank 0's child
    writes the JSON report.  Returns (report dict | None on ranks other than the one holding rank 0, reason, stage the slowest child had
    reached).  host_exchange: the halo exchange staged through host memory and the ranks' shared segment instead of RCCL
    (USPMV_EXCHANGE=host) -- the tier that still yields a native, checked number when RCCL does not come up on the machine."""
    import tempfile
    world, rank = parents.world, parents.rank
    exe = os.environ.get("USPMV_BENCH_EXE") or os.path.join(ROOT, "ultimate-spmv_amd", "uspmv")     # (USPMV_BENCH_EXE: the launcher's CPU tests)
    g = grid
    nz = g * world if scaling == "weak" else g
    cores = max(1, usable_cores() // max(1, world if parents.self_launch else int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))))
    tmp = tempfile.mkdtemp(prefix=f"uspmv_bench_r{rank}_")
    js = os.path.join(tmp, "report.json")
    cmd = [exe, f"gen:{g}x{g}x{nz}", "scs", "-c", str(args.chunk), "-s", str(args.sigma), "-dp", "-" + args.seg.replace("-", "_"), "-comm_halos", "1",
           "-ba_synch", str(args.ba_synch), "-bench_steps", str(args.steps), "-bench_warmup", str(args.warmup), "-check_y", "1", "-json", js,
           "-tlc", "0" if args.no_tlc else "1", "-graph", "0" if (args.no_graph or eager) else "1", "-step_form", args.step_form]
    job = f"bench{os.environ.get('MASTER_PORT', str(os.getppid() if not parents.self_launch else os.getpid()))}_{scaling}_{g}{'_hx' if host_exchange else ''}{'_e' if eager else ''}"

    def env_of(r):
        env = dict(os.environ, OMP_NUM_THREADS=str(cores), USPMV_JOB_ID=job, USPMV_STAGES="1", RANK=str(r), WORLD_SIZE=str(world),
                   LOCAL_RANK=str(r) if parents.self_launch else os.environ.get("LOCAL_RANK", str(r)))
        env.setdefault("USPMV_HC_TIMEOUT", str(int(timeout_s)))     # (the children's rendezvous gives up with them, not an hour later)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # (dmabuf IPC: without it RCCL's hipIpcGetMemHandle fails on this driver)
        if host_exchange:
            env["USPMV_EXCHANGE"] = "host"
        if args.no_overlap:
            env["USPMV_NO_OVERLAP"] = "1"
        return env

    sim = os.environ.get("USPMV_BENCH_SIMULATE_RCCL_FAILURE")      # rehearsal of the later tiers: these children fail at once
    fail_now = bool(sim) and not host_exchange and (not eager or sim == "2")
    res = run_children(lambda r: ["/bin/false"] if fail_now else cmd, env_of, parents.my_ranks(), timeout_s, tmp)
    mine_bad = max((1 if rc else 0) for rc, _, _ in res.values())
    rcs = parents.allgather(mine_bad)
    stages = "; ".join(f"rank {r}: rc {rc}, {stg}" for r, (rc, _, stg) in sorted(res.items()))
    if any(rcs):
        for r, (rc, tail, _) in sorted(res.items()):
            if rc:
                sys.stderr.write(f"[bench] uspmv child of rank {r} ended with rc {rc}:\n{tail}\n")
        hung = any(rc == -9 for rc, _, _ in res.values())
        return None, ("uspmv child processes " + ("killed at the time limit of %.0f s" % timeout_s if hung else "failed") + f" ({stages})"), stages
    rep = None
    if 0 in res:
        try:
            rep = json.load(open(js))
            rep["cmd"] = " ".join(cmd[1:])
        except (OSError, ValueError) as e:
            return None, f"rank 0's report is unreadable: {e}", stages
    return rep, "", stages


def single_gpu_reference(args, grid, timeout_s):
    """The SAME matrix on ONE GPU (strong scaling's denominator): `uspmv gen:... scs -c 32 -s 512 -dp -bench_steps K -json` as a single-rank
    child on GPU 0, all usable cores for its set-up, before any communicator exists.  Returns a dict (ms_per_step, ...) or {"error": ...}."""
    import tempfile
    exe = os.environ.get("USPMV_BENCH_EXE") or os.path.join(ROOT, "ultimate-spmv_amd", "uspmv")
    tmp = tempfile.mkdtemp(prefix="uspmv_bench_single_")
    js = os.path.join(tmp, "single.json")
    cmd = [exe, f"gen:{grid}x{grid}x{grid}", "scs", "-c", str(args.chunk), "-s", str(args.sigma), "-dp", "-bench_steps", str(args.steps),
           "-bench_warmup", str(args.warmup), "-json", js, "-tlc", "0" if args.no_tlc else "1"]

    def env_of(_):
        env = dict(os.environ, OMP_NUM_THREADS=str(usable_cores()))
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "USPMV_LOOPBACK", "USPMV_FORCE_DIST"):
            env.pop(k, None)
        return env

    t0 = time.time()
    rc, tail, _ = run_children(lambda r: cmd, env_of, [0], timeout_s, tmp)[0]
    if rc:
        return {"error": f"single-GPU child rc {rc}" + (" (killed at its time limit of %.0f s)" % timeout_s if rc == -9 else "") + f": {tail[-300:]}"}
    try:
        d = json.load(open(js))
    except (OSError, ValueError) as e:
        return {"error": f"single-GPU report unreadable: {e}"}
    return {"ms_per_step": d["ms_per_step"], "kernel_ms": d["kernel_ms"], "gflops": d["gflops"], "n_rows": d["n_rows"], "nnz": d["nnz"],
            "algorithmic_GBs": d["algorithmic_GBs"], "frac_of_8TBs": round(d["algorithmic_GBs"] / HBM_PEAK_GBS, 4),
            "plan_tiles_planned": [d["plan_tiles_planned"], d["plan_tiles"]], "setup_s": d["setup_s"], "wall_s": round(time.time() - t0, 1),
            "cmd": "uspmv " + " ".join(cmd[1:])}


def cli_result(args, rep, scaling, grid, world):
    g = grid
    nz = g * world if scaling == "weak" else g
    klass = "nlpkkt240-class" if g == 304 else "nlpkkt200-class" if g == 253 else "stencil"
    r0 = rep["rank0"]
    k_ms = r0["local_kernel_ms"]
    per_rank = []
    for r in rep.get("per_rank", []):
        r = dict(r)
        r["local_kernel_GBs"] = round(r["algorithmic_bytes"] / (r["local_kernel_ms"] * 1e-3) / 1e9, 1) if r.get("local_kernel_ms") else None
        per_rank.append(r)
    return {
        "value": round(rep["gflops"], 2), "ms_per_step": round(rep["ms_per_step"], 5), "scaling": scaling,
        "workload": (f"{klass} synthetic (27-pt stencil {g}x{g}x{nz}, n={rep['n_rows']}, nnz={rep['nnz']}) scs -c {args.chunk} -s {args.sigma} -dp "
                     f"-{args.seg.replace('-', '_')} -comm_halos 1"),
        "n_rows": rep["n_rows"], "nnz": rep["nnz"], "beta": None,
        "exchange": rep.get("exchange"), "rccl_nranks": rep.get("rccl_nranks"), "ranks": rep.get("ranks"),
        "step": "uspmv child processes on the system RCCL: C++ uspmv_dist_run, " + ("hipGraph replay" if rep["graph_replay"] else "eager C++ steps"),
        "protocol": f"exactly {rep['steps']} steps between barriers after {rep['warmup']} warm-ups, slowest rank's clock; -ba_synch {rep['ba_synch']} "
                    f"(with -ba_synch {1 - rep['ba_synch']}: {rep['other_ba_synch_ms_per_step']:.5f} ms per step; 1 = the reference's default, a barrier behind every step, code/main.cpp:467)",
        "y_checked": rep["y_checked"], "y_mismatches": rep["y_mismatches"],
        "step_form": rep.get("step_form"), "step_form_candidates_ms": rep.get("step_form_candidates_ms"),
        "rank0": {"n_local": r0["n_local"], "n_halo": r0["n_halo"], "n_send": r0["n_send"], "interior": r0["interior"], "boundary": r0["boundary"],
                  "tiles": r0["tiles"], "plan_kind": 1 if r0["tiles"] else 0, "local_kernel_ms": round(k_ms, 5), "algorithmic_bytes": int(r0["algorithmic_bytes"]),
                  "local_kernel_GBs": round(r0["algorithmic_bytes"] / (k_ms * 1e-3) / 1e9, 1) if k_ms > 0 else None},
        "per_rank": per_rank,
        "versions": rep["versions"], "cmd": "uspmv " + rep["cmd"],
    }


def dist_bench(args, pkg, world, rank, self_launch, tuning):
    """N > 1 (BASELINE config 5): the `uspmv` rank processes do everything on the GPUs; this process only starts them, keeps ONE wall-clock
    budget over the whole run and prints ONE JSON line -- with "value": null and the reason when no tier produced a number in time.
    Order: [strong scaling] the same matrix on one GPU (single-rank child) -> tier 1 graph replay on RCCL -> tier 2 eager steps on RCCL
    (only when one eager step of tier 1 completed on every rank: a capture gone wrong takes the child down; an exchange that does not
    work eagerly would not work in tier 2 either) -> tier 3 the exchange staged through host memory (no RCCL) -> the other scaling mode when at
    least a third of the budget is left.  Every child is a fresh process; nothing that touched a GPU is ever re-used or re-exec'ed."""
    t_start = time.time()
    left = lambda: args.budget_s - (time.time() - t_start)
    first = args.scaling or "strong"
    second = "weak" if first == "strong" else "strong"
    grid1 = args.grid or (304 if first == "strong" else 253)
    grid2 = args.grid2 or (253 if second == "weak" else 304)
    parents = Parents(pkg, rank, world, self_launch, timeout_s=args.budget_s + 60)
    tiers, res, other, reason, single = [], None, None, "", None
    lead = rank == 0
    worth = lambda: left() > min(40.0, 0.15 * args.budget_s)                # (is there time for another tier?)
    lead_says = lambda flag: bool(parents.allgather(int(bool(flag)))[0])    # (clock-dependent decisions are rank 0's: every parent takes the same road)

    def tier(name, scaling, grid, share, **kw):
        """(ok on every rank, report on the lead, reason, did every child get as far as its steps) -- the same verdicts on all parents"""
        t0 = time.time()
        tmo = max(1.0, min(max(20.0, left() * share), left() - 4.0))     # (never past the budget: 4 s are kept for ending the children)
        rep, why, stages = cli_measure(args, parents, scaling, grid, tmo, **kw)
        ok = bool(min(parents.allgather(0 if (why or (lead and rep is None)) else 1)))
        # (did one EAGER step with its exchange complete on every rank?  Then a failure later on -- graph capture inside the step-form timing
        #  or the timed steps -- is what the eager tier is for; if not, eager steps would fail the same way and the tier is skipped)
        steps_seen = bool(min(parents.allgather(int(any(k in stages for k in ("first eager step done", "timing the step forms", "step form chosen", "timed region", "report written"))))))
        tiers.append({"tier": name, "scaling": scaling, "ok": ok, "wall_s": round(time.time() - t0, 1), "time_limit_s": round(tmo, 1),
                      **({} if ok else {"why": why or "failed on another rank"})})
        return ok, (rep if ok else None), why or ("" if ok else "failed on another rank"), steps_seen

    if first == "strong" and not args.no_single_gpu_reference:
        if lead:
            single = single_gpu_reference(args, grid1, max(1.0, min(0.3 * left(), 200.0)))
        parents.allgather(0)                                         # (the other parents wait for rank 0's single-GPU child here)
    ok, rep, why, steps_seen = tier("1: hipGraph replay, RCCL" if not args.no_graph else "1: eager steps, RCCL", first, grid1, 0.55)
    if not ok:
        reason = why
        if not args.no_graph and steps_seen and lead_says(worth()):
            ok, rep, why2, _ = tier("2: eager steps, RCCL", first, grid1, 0.5, eager=True)
            reason += "; measured with eager steps instead of graph replay" if ok else f"; eager retry: {why2}"
        if not ok and lead_says(worth()):
            ok, rep, why3, _ = tier("3: exchange staged through host memory (no RCCL)", first, grid1, 0.85, host_exchange=True)
            reason += "; measured with the halo exchange staged through host memory instead of RCCL" if ok else f"; host-staged retry: {why3}"
        if not ok and not worth():
            reason += "; the budget ended"
    host_tier = False
    if ok and lead:
        res = cli_result(args, rep, first, grid1, world)
        host_tier = rep.get("exchange") == "host"
        if host_tier:
            res["step"] = res["step"].replace("on the system RCCL", "with the host-staged exchange (USPMV_EXCHANGE=host)")
    host_tier = bool(max(parents.allgather(int(host_tier))))
    if ok and not args.no_second_line:
        if lead_says(left() >= args.budget_s / 3.0):
            ok2, rep2, why2, _ = tier("second line (" + second + " scaling)", second, grid2, 0.9, host_exchange=host_tier, eager=bool(args.no_graph))
            if lead:
                other = cli_result(args, rep2, second, grid2, world) if ok2 else {"value": None, "scaling": second, "error": why2}
                if ok2 and rep2.get("exchange") == "host":
                    other["step"] = other["step"].replace("on the system RCCL", "with the host-staged exchange (USPMV_EXCHANGE=host)")
        elif lead:
            other = {"value": None, "scaling": second, "error": f"skipped: {left():.0f} s of the {args.budget_s:.0f} s budget left (< 1/3)"}
    parents.close()
    if not lead:
        return 0
    out = {
        "metric": "SpMV GFLOP/s (SELL-32-512 dp; achieved HBM GB/s in roofline)",
        "value": res["value"] if res else None, "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": res["ms_per_step"] if res else None, "higher_is_better": True, "scaling": first, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
    }
    if res:
        r0 = res["rank0"]
        slow = max(res["per_rank"], key=lambda r: r.get("local_kernel_ms") or 0.0) if res["per_rank"] else None
        out.update({
            "y_checked": res.get("y_checked"), "y_mismatches": res.get("y_mismatches"),
            "config": {"workload": res["workload"], "C": args.chunk, "sigma": args.sigma, "n_rows": res["n_rows"], "nnz": res["nnz"],
                       "x": "5.0 (DefaultValues) in the timed steps; x_global[j] = 1 + 1e-3 (j mod 1000) in the checked step", "partition": args.seg,
                       "halo_overlap": (res.get("step_form") or ("plain" if args.no_overlap else "overlap")) != "plain", "step": res["step"], "protocol": res.get("protocol"),
                       "step_form": res.get("step_form"), "step_form_candidates_ms": res.get("step_form_candidates_ms"), "exchange": res.get("exchange"),
                       "rccl_nranks": res.get("rccl_nranks"), "ranks": res.get("ranks"),
                       "launcher": "bench.py started the uspmv rank processes itself" if self_launch else "torch.distributed.run ranks, each started its uspmv rank process",
                       "rank0": r0, "per_rank": res["per_rank"], "tuning": tuning, "versions": res.get("versions"), "cmd": res.get("cmd")},
            "roofline": {"bound": "hbm", "achieved": r0["local_kernel_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(r0["local_kernel_GBs"] / HBM_PEAK_GBS, 4) if r0["local_kernel_GBs"] else None, "traffic": None,
                         "kernel": "rank 0's local SpMV (interior + boundary tiles, no exchange): " + ("scs_spmv_tlc<double,32>" if r0["plan_kind"] == 1 else "scs_spmv_rows<double,32,8>"),
                         "kernel_ms": r0["local_kernel_ms"], "algorithmic_bytes_per_launch": r0["algorithmic_bytes"],
                         "rank0_step_frac": round(r0["algorithmic_bytes"] / (res["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "slowest_rank": None if not slow else {"rank": slow["rank"], "kernel_ms": slow["local_kernel_ms"], "achieved": slow["local_kernel_GBs"],
                                                                "frac": round(slow["local_kernel_GBs"] / HBM_PEAK_GBS, 4) if slow["local_kernel_GBs"] else None}},
        })
        if first == "strong":
            out["single_gpu_same_matrix"] = single
            if single and "ms_per_step" in single:
                out["speedup_vs_single_gpu"] = round(single["ms_per_step"] / res["ms_per_step"], 3)
                out["strong_scaling_efficiency"] = round(single["ms_per_step"] / (world * res["ms_per_step"]), 4)
    else:
        out["error"] = reason or "no tier produced a number"
        out["config"] = {"workload": f"27-pt stencil {grid1}^3 scs -c {args.chunk} -s {args.sigma} -dp -{args.seg.replace('-', '_')} -comm_halos 1", "tuning": tuning}
        if single is not None:
            out["single_gpu_same_matrix"] = single
    if reason:
        out["fallback_reason"] = reason
    if other is not None:
        out[second + "_scaling"] = {k: other.get(k) for k in ("value", "ms_per_step", "scaling", "workload", "step", "exchange", "protocol", "y_checked", "y_mismatches", "rank0", "per_rank", "rccl_nranks", "error") if k in other}
        out[second + "_scaling"]["unit"] = "GFLOP/s"
    out["budget"] = {"budget_s": args.budget_s, "used_s": round(time.time() - t_start, 1), "tiers": tiers}
    print(json.dumps(out), flush=True)
    return 0 if res else 1


def run_distributed(args, pkg, B, torch, dist, dev, world, rank, scaling, grid):
    """One N > 1 measurement: the C++ step (uspmv_dist_run) on one RCCL communicator, K timed steps between barriers."""
    from ultimate_spmv_amd.distributed import DistSpmv
    g = grid
    nz = g * world if scaling == "weak" else g
    t_setup = time.time()
    counts = pkg.gen_stencil27_row_counts(g, g, nz)
    n_global, total_nnz = int(counts.size), int(counts.sum(dtype=np.int64))
    wsa = pkg.seg_from_row_counts(counts, args.seg, world)
    del counts
    loc = pkg.gen_stencil27(g, g, nz, row_begin=int(wsa[rank]), row_end=int(wsa[rank + 1]))
    native = not args.python_step and not os.environ.get("USPMV_BENCH_ONE_DEVICE")
    native_error = None
    if native:
        idt = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(pkg.comm_unique_id()), dtype=torch.uint8))
        with stdout_to_stderr():
            dist.broadcast(idt, 0)                # (the first collective creates torch's communicator)
        try:
            with stdout_to_stderr():
                d = pkg.DistNative(loc, wsa, args.chunk, args.sigma, rank, world, bytes(idt.cpu().numpy().tobytes()), tlc=not args.no_tlc)
            if args.no_overlap:
                d.set_overlap(False)
        except Exception as e:          # e.g. ncclCommInitRank refused: every rank must take the same road, and the line must say so
            native_error = f"{type(e).__name__}: {e}"
            d = None
        bad = torch.tensor([1 if d is None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()):
            native_error = native_error or "another rank could not create the C++ step object"
            if d is not None:
                d.close()
            native = False
    if native:                                    # one step of the C++ object before anything is timed: if it fails on ANY rank, every rank takes the python step
        try:
            x = d.new_x(np.full(d.n_local, 5.0))
            y = d.new_y()
            d.run(x, y, 1, use_graph=False)
            torch.cuda.synchronize(dev)
        except Exception as e:
            native_error = f"first C++ step failed: {type(e).__name__}: {e}"
        bad = torch.tensor([1 if native_error else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()):
            native_error = native_error or "the first C++ step failed on another rank"
            try:
                d.close()
            except Exception:
                pass
            native = False
    y_ref_twin = None
    if not native:
        d = DistSpmv(loc, wsa, args.chunk, args.sigma, B.F64, device=dev, overlap=not args.no_overlap, tlc=not args.no_tlc)
        y_ref_twin = pkg.dist_check_reference(loc, wsa, rank, world)   # host side of the checked step below (entry-ordered FMA chains)
        del loc
        loc = None
    x = d.new_x(np.full(d.n_local, 5.0))          # DefaultValues::x (code/classes_structs.hpp:1799-1800)
    y = d.new_y()
    s = d.scs
    bytes_local = s.n_elements * 12 + 8 * s.n_chunks + 8 * (d.n_local + d.n_halo) + 8 * s.n_rows_padded
    t_setup = time.time() - t_setup

    def sync():
        torch.cuda.synchronize(dev)
        dist.barrier()
        torch.cuda.synchronize(dev)

    def steps(n):
        if native:
            d.run(x, y, n, use_graph=args.graph)
        else:
            for _ in range(n):
                d.spmv(x, y)

    step_form = step_form_ms = None
    if native and not args.no_overlap:            # the arrangement of the step that is fastest on THIS machine (collective; all forms give the same bits)
        step_form, step_form_ms = d.autotune(x, y, use_graph=args.graph, local=loc, wsa=wsa)
    steps(args.warmup)
    sync()
    t0 = time.perf_counter()
    steps(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
    ms = elapsed / args.steps * 1e3
    y_checked, y_bad = None, None
    if native:
        d._refresh()
        # one checked step: x_global[j] = 1 + 1e-3 (j mod 1000), y of the local rows bitwise against the entry-ordered FMA chains
        bad, _ = d.check(loc, x, y, use_graph=args.graph)
        del loc
        tb = torch.tensor([bad], dtype=torch.int64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tb)
        y_bad = int(tb.item())
        y_checked = y_bad == 0
        x = d.new_x(np.full(d.n_local, 5.0))
        # the local kernel alone (interior + boundary tiles without the exchange), HIP events on the object's stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(d.stream):
            d.spmv(x, y, comm_halos=False)
            e0.record(d.stream)
            for _ in range(20):
                d.spmv(x, y, comm_halos=False)
            e1.record(d.stream)
        d.synchronize()
        k_ms = e0.elapsed_time(e1) / 20
        kind = d.plan_info()[0]
    else:
        # the twin's checked step: the same ramp, y of the local rows bitwise against the host chains
        xr = d.new_x(1.0 + 1e-3 * (np.arange(int(wsa[rank]), int(wsa[rank + 1])) % 1000))
        yr = d.new_y()
        d.spmv(xr, yr)
        torch.cuda.synchronize(dev)
        got = d.y_to_original_order(yr)[:d.n_local]
        tb = torch.tensor([int(np.count_nonzero(got.view(np.uint64) != y_ref_twin.view(np.uint64)))], dtype=torch.int64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tb)
        y_bad = int(tb.item())
        y_checked = y_bad == 0
        del xr, yr, y_ref_twin
        k_ms = B.time_launches(0, 20, A=d.A, x=x, y=y)
        kind = 1 if d.use_tiles else 0
    klass = "nlpkkt240-class" if g == 304 else "nlpkkt200-class" if g == 253 else "stencil"
    res = {
        "value": round(2.0 * total_nnz / (ms * 1e-3) / 1e9, 2), "ms_per_step": round(ms, 5), "scaling": scaling,
        "workload": (f"{klass} synthetic (27-pt stencil {g}x{g}x{nz}, n={n_global}, nnz={total_nnz}) scs -c {args.chunk} -s {args.sigma} -dp "
                     f"-{args.seg.replace('-', '_')} -comm_halos 1"),
        "n_rows": n_global, "nnz": total_nnz, "beta": round(s.nnz / s.n_elements, 6),
        "step": ("C++ uspmv_dist_run, " + ("hipGraph replay" if d.graph_captured else "eager (graph capture refused)" if args.graph else "eager C++ steps")) if native
                else "python: torch.distributed all_to_all_single per step" + (f" (C++ step object unavailable: {native_error})" if native_error else ""),
        "rank0": {"n_local": d.n_local, "n_halo": d.n_halo, "n_send": d.n_send, "interior": int(d.n_interior) if native else int(len(d.interior_ids)),
                  "boundary": int(d.n_boundary) if native else int(len(d.boundary_ids)), "tiles": bool(d.use_tiles), "plan_kind": kind,
                  "local_kernel_ms": round(k_ms, 5), "algorithmic_bytes": int(bytes_local),
                  "local_kernel_GBs": round(bytes_local / (k_ms * 1e-3) / 1e9, 1)},
        "setup_s": round(t_setup, 1), "y_checked": y_checked, "y_mismatches": y_bad, "native": bool(native),
        "step_form": step_form, "step_form_candidates_ms": step_form_ms,
        "protocol": f"exactly {args.steps} steps between barriers after {args.warmup} warm-ups, slowest rank's clock; -ba_synch 0",
    }
    if native:
        d.close()
    del d, x, y
    torch.cuda.empty_cache()
    return res


def in_process_distributed(args, pkg, world, rank, local_rank, tuning):
    """--in-process / --python-step (explicit options, under torch.distributed.run only; NOT part of the default chain): the C++ step object
    inside this torch process on torch's bundled RCCL, or the round-1 torch.distributed twin."""
    import torch
    import torch.distributed as dist
    from ultimate_spmv_amd import binding as B
    first = args.scaling or "strong"
    second = "weak" if first == "strong" else "strong"
    grid1 = args.grid or (304 if first == "strong" else 253)
    grid2 = args.grid2 or (253 if second == "weak" else 304)
    if os.environ.get("USPMV_BENCH_ONE_DEVICE"):   # rehearsal only: several ranks share GPU 0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if os.environ.get("USPMV_BENCH_ONE_DEVICE"):
        dist.init_process_group("gloo", rank=rank, world_size=world)      # rehearsal: RCCL refuses duplicate GPUs
    else:
        with stdout_to_stderr():
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    res = run_distributed(args, pkg, B, torch, dist, dev, world, rank, first, grid1)
    other = None if args.no_second_line else run_distributed(args, pkg, B, torch, dist, dev, world, rank, second, grid2)
    if rank == 0:
        r0 = res["rank0"]
        twin = not res.get("native", True)
        out = {
            "metric": "SpMV GFLOP/s (SELL-32-512 dp; achieved HBM GB/s in roofline)",
            "value": None if twin else res["value"], "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": first, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "y_checked": res.get("y_checked"), "y_mismatches": res.get("y_mismatches"),
            "config": {"workload": res["workload"], "C": args.chunk, "sigma": args.sigma, "n_rows": res["n_rows"], "nnz": res["nnz"], "beta": res["beta"],
                       "x": "5.0 (DefaultValues) in the timed steps; x_global[j] = 1 + 1e-3 (j mod 1000) in the checked step", "partition": args.seg,
                       "halo_overlap": (res.get("step_form") or ("plain" if args.no_overlap else "overlap")) != "plain", "step": res["step"], "protocol": res.get("protocol"),
                       "step_form": res.get("step_form"), "step_form_candidates_ms": res.get("step_form_candidates_ms"),
                       "launcher": "torch.distributed.run ranks, the step inside the torch process (--in-process / --python-step)",
                       "rank0": r0, "tuning": tuning},
            "roofline": {"bound": "hbm", "achieved": r0["local_kernel_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(r0["local_kernel_GBs"] / HBM_PEAK_GBS, 4) if r0["local_kernel_GBs"] else None, "traffic": None,
                         "kernel": "rank 0's local SpMV (interior + boundary tiles, no exchange): " + ("scs_spmv_tlc<double,32>" if r0["plan_kind"] == 1 else "scs_spmv_rows<double,32,8>"),
                         "kernel_ms": r0["local_kernel_ms"], "algorithmic_bytes_per_launch": r0["algorithmic_bytes"],
                         "rank0_step_frac": round(r0["algorithmic_bytes"] / (res["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "setup_s": res.get("setup_s"),
        }
        if twin:
            out["fallback_value"] = res["value"]
        if other is not None:
            out[second + "_scaling"] = {k: other.get(k) for k in ("value", "ms_per_step", "scaling", "workload", "step", "protocol", "y_checked", "y_mismatches", "rank0", "setup_s", "error") if k in other}
            out[second + "_scaling"]["unit"] = "GFLOP/s"
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0


def main():
    args = parse()
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.pmc_child:
        return pmc_child(args)

    # ================================================================== N > 1: BASELINE config 5 (strong) + the weak line
    if args.gpus > 1 or world_env > 1 or os.environ.get("USPMV_BENCH_WORLD1"):
        in_proc = args.in_process or args.python_step or args.graph or os.environ.get("USPMV_BENCH_WORLD1") or os.environ.get("USPMV_BENCH_ONE_DEVICE")
        self_launch = world_env == 1 and not in_proc       # no outer launcher: this process starts the N rank processes itself
        world = args.gpus if self_launch else world_env
        if not self_launch and world != args.gpus:
            args.gpus = world
        set_omp_threads(max(1, usable_cores() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))))
        import __graft_entry__ as ge
        if not in_proc:
            os.environ["USPMV_NO_TORCH"] = "1"             # this process never touches a GPU and never needs torch: the library binds to the system runtime
        pkg = ge.load_package()
        tuning = None
        if in_proc:
            if world_env == 1 and not os.environ.get("USPMV_BENCH_WORLD1"):
                raise SystemExit("--in-process / --python-step / --graph need torch.distributed.run (one rank per GPU); the default N > 1 path does not")
            for kv in filter(None, args.tune.split(",")):
                k, v = kv.split("=")
                pkg.set_tuning(**{k: int(v)})
            tuning = {k: pkg.get_tuning(k) for k in ("unroll", "nontemporal", "xcd_remap", "block", "spmv_variant", "tlc")}
            return in_process_distributed(args, pkg, world_env, rank, local_rank, tuning)
        if args.tune:
            sys.stderr.write("bench.py: --tune applies to in-process measurements only (the uspmv children run the library defaults)\n")
        return dist_bench(args, pkg, world, 0 if self_launch else rank, self_launch, "library defaults (the uspmv rank processes)")

    traffic, traffic_note = None, "not measured"
    if not args.mtx and not args.no_traffic:
        traffic, traffic_note = measure_traffic(args)      # child processes under rocprofv3, BEFORE this process touches the GPU
    import torch
    import __graft_entry__ as ge

    pkg = ge.load_package()
    from ultimate_spmv_amd import binding as B

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    set_omp_threads(usable_cores())
    for kv in filter(None, args.tune.split(",")):
        k, v = kv.split("=")
        pkg.set_tuning(**{k: int(v)})
    tuning = {k: pkg.get_tuning(k) for k in ("unroll", "nontemporal", "xcd_remap", "block", "spmv_variant", "tlc")}

    # ================================================================== N = 1: BASELINE config 2
    t_setup = time.time()
    if args.mtx:
        loc = pkg.read_mtx(args.mtx)
        workload = f"{os.path.basename(args.mtx)} scs -c {args.chunk} -s {args.sigma} -dp"
    else:
        g = args.grid or 253
        loc = pkg.gen_stencil27(g, g, g)
        workload = (f"nlpkkt200-class synthetic (27-pt stencil {g}x{g}x{g}, n={loc.n_rows}, nnz={loc.nnz}) "
                    f"scs -c {args.chunk} -s {args.sigma} -dp")
    total_nnz, n_global = loc.nnz, loc.n_rows
    s = pkg.convert_to_scs(loc, args.chunk, args.sigma, B.F64)
    a = s.arrays()
    pkg.permute_scs_cols(s, a["old_to_new_idx"])
    del loc
    A = pkg.DeviceMatrix(s, dev, tlc=not args.no_tlc)
    x = torch.full((s.n_rows_padded,), 0.0, dtype=torch.float64, device=dev)
    x[:s.n_rows] = 5.0                            # DefaultValues::x (code/classes_structs.hpp:1799-1800); a constant is its own permutation
    y = torch.zeros(s.n_rows_padded, dtype=torch.float64, device=dev)
    bytes_local = s.n_elements * 12 + 8 * s.n_chunks + 8 * s.n_rows + 8 * s.n_rows_padded
    t_setup = time.time() - t_setup

    # ------------------------------------------------------------------ one checked launch on a NON-uniform x (a constant would hide a permutation
    #                                                                    error): its y is compared bitwise with the CPU kernel's further down
    x_check = np.zeros(s.n_rows_padded)
    x_check[:s.n_rows] = pkg.apply_permutation(1.0 + 1e-3 * (np.arange(s.n_rows) % 1000), a["new_to_old_idx"])
    xc = torch.from_numpy(x_check).to(dev)
    B.spmv(A, xc, y)
    torch.cuda.synchronize(dev)
    y_gpu_check = y.cpu().numpy().copy()
    del xc
    card = card_of_device(torch, local_rank)
    state_idle = gpu_state(card)

    # ------------------------------------------------------------------ dominant kernel, HIP events on its stream (first: 100 launches also bring
    #                                                                    the GPU out of the power state the host-side set-up left it in)
    reps = 100
    with GpuStateSampler(card) as smp:
        k_ms = B.time_launches(0, reps, A=A, x=x, y=y)
    state_load = smp.summary()

    # ------------------------------------------------------------------ timed region: W warm-up steps, then EXACTLY K steps
    for _ in range(args.warmup):
        B.spmv(A, x, y)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        B.spmv(A, x, y)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    ms_per_step = elapsed / args.steps * 1e3
    gflops = 2.0 * total_nnz / (ms_per_step * 1e-3) / 1e9

    # ------------------------------------------------------------------ what y's placement alone is worth in THIS process: the same
    #                                                                    kernel, matrix and x, four more output vectors (fresh allocations, all held)
    placement = None
    try:
        ys = [torch.zeros(s.n_rows_padded, dtype=torch.float64, device=dev) for _ in range(3)]
        raw = torch.zeros(s.n_rows_padded * 8 + (2 << 20), dtype=torch.uint8, device=dev)
        ys.append(raw[1 << 20:(1 << 20) + s.n_rows_padded * 8].view(torch.float64))                 # one that starts 1 MiB behind a 2 MiB boundary
        placement = {"kernel_ms_per_output_vector": [round(B.time_launches(0, 30, A=A, x=x, y=yy), 5) for yy in [y] + ys],
                     "note": "same kernel, matrix and x; y = the line's own vector, three more fresh allocations, one placed 1 MiB behind a 2 MiB boundary "
                             "(informational: value / roofline use the first)"}
        del ys, raw
    except Exception as e:      # noqa: BLE001 -- an informational extra must not cost the line
        placement = {"error": f"{type(e).__name__}: {e}"}

    achieved = bytes_local / (k_ms * 1e-3) / 1e9
    g_ = args.grid or 253
    stream = stream_rates(B, torch, dev, x_elems=0 if args.mtx else s.n_rows, plane=g_ * g_, line=g_)
    kind, n_tiles, n_planned = A.plan_info()
    kname = {0: "scs_spmv_rows<double,32,8>", 1: "scs_spmv_tlc<double,32>", 2: "scs_spmv_sweep<double>"}[kind if pkg.get_tuning("tlc") else 0]
    if traffic is None and not args.mtx:
        key = f"stencil27-{args.grid or 253}x{args.grid or 253}x{args.grid or 253}-c{args.chunk}-s{args.sigma}-f64-n1"
        try:
            traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(key, {}).get("traffic_bytes")
            traffic_note = f"NOT measured in this run ({traffic_note}); value from profiles/traffic.json (rocprofv3 PMC passes of an earlier run of this command)"
        except (OSError, ValueError):
            traffic = None
    out = {
        "metric": "SpMV GFLOP/s (SELL-32-512 dp; achieved HBM GB/s in roofline)",
        "value": round(gflops, 2), "unit": "GFLOP/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": None, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic" if not args.mtx else "file",
        "config": {"workload": workload, "C": args.chunk, "sigma": args.sigma, "n_rows": n_global, "nnz": total_nnz,
                   "beta": round(s.nnz / s.n_elements, 6), "x": "5.0 (DefaultValues)", "partition": "none", "halo_overlap": False,
                   "kernel": {0: "lane-per-row gather", 1: f"tile-local-column (LDS-staged x lines, {A.index_bits()}-bit local indices)",
                              2: "column-window sweep (LDS-staged x windows, compacted stream)"}[kind],
                   "plan_tiles_planned": [n_planned, n_tiles], "tuning": tuning},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": traffic, "traffic_source": traffic_note,
                     "achieved_physical": None if not traffic else round(traffic / (k_ms * 1e-3) / 1e9, 1),
                     "frac_physical": None if not traffic else round(traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "kernel": kname, "kernel_ms": round(k_ms, 5),
                     "algorithmic_bytes_per_launch": int(bytes_local),
                     "note": "achieved/frac = algorithmic bytes (SURVEY 8d formula, 12 B per non-zero) / kernel time; achieved_physical/frac_physical = "
                             "HBM bytes really moved (PMC) / kernel time -- the plan kernels stream 10 B per non-zero",
                     "stream_same_run_GBs": stream,
                     "frac_physical_of_stream_read": None if not traffic else round(traffic / (k_ms * 1e-3) / 1e9 / stream["read"], 4)},
        "setup_s": round(t_setup, 1),
    }
    out["gpu_state"] = {"idle_before_timing": state_idle, "during_the_kernel_timing": state_load,
                        "note": "amdgpu sysfs (pp_dpm_* active levels, hwmon sensors in their native units: microwatts, millidegrees, Hz), read only"}
    out["y_placement"] = placement
    out["order"] = "set-up, one checked launch, HIP-event timing of the kernel (100 launches), W warm-up steps, K timed steps, stream yardsticks, CPU baseline, other configs"
    if not args.no_cpu_baseline:
        a = s.arrays()
        out["cpu_baseline"], same = cpu_baseline(a, s.C, s.n_chunks, s.nnz, x.cpu().numpy(), args.cpu_seconds, x_check=x_check, y_gpu_check=y_gpu_check)
        out["bitexact_vs_reference_cpu"] = same
        out["bitexact_note"] = ("one launch on x_i = 1 + 1e-3 (i mod 1000) (permuted), y of all n_rows_padded rows compared bitwise with the y of " +
                                ("the reference's spmv_omp_scs_adv<32> (oracle/_ref)" if out["cpu_baseline"]["kind"] == "reference" else "the oracle's C port") + " for the same x")
    if not args.no_vendor_baseline and not args.mtx:
        out["vendor_baseline"] = vendor_baseline(args.grid or 253)
    which = [w for w in args.other_configs.split(",") if w]
    if which and not args.mtx:
        del A, x, y, s
        torch.cuda.empty_cache()
        try:
            out["other_configs"] = other_configs(pkg, B, torch, args, which)
        except Exception as e:      # the headline line must survive a failure down here
            out["other_configs"] = {"error": f"{type(e).__name__}: {e}"}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    sys.exit(main() or 0)
