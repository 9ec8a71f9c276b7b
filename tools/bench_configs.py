#!/usr/bin/env python3
"""Single-GPU BASELINE configurations 2-4 (synthetic stand-ins): kernel time (HIP events),
GF/s, algorithmic GB/s, and a full-size parity check against the oracle.  One JSON line per config."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="2,3,4")
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--scale", type=float, default=1.0, help="shrink grids for quick runs")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--tune", default="", help="key=value,... passed to uspmv_set_tuning")
    ap.add_argument("--cpu-seconds", type=float, default=0.0, help="> 0: also time the genuine reference CPU kernel (oracle/_ref) for about this long")
    ap.add_argument("--sigma", type=int, default=512, help="sorting scope (the BASELINE configurations use 512)")
    ap.add_argument("--sp", action="store_true", help="config 3 in single precision (block plan kernel)")
    ap.add_argument("--no-block-plan", action="store_true", help="config 3 --sp without uspmv_dmat_optimize_block")
    args = ap.parse_args()
    import torch
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from ultimate_spmv_amd import binding as B
    from oracle import oracle as orc
    torch.cuda.set_device(0)
    t = torch
    if args.tune:
        pkg.set_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in args.tune.split(","))})

    def prep(coo, dtype, fixed=None):
        s = pkg.convert_to_scs(coo, 32, args.sigma, dtype, fixed_permutation=fixed)
        return s

    for cfg in args.configs.split(","):
        t0 = time.time()
        if cfg == "2":
            g = int(253 * args.scale)
            coo = pkg.gen_stencil27(g, g, g)
            s = prep(coo, pkg.F64)
            a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"]); a = s.arrays()
            A = pkg.DeviceMatrix(s, tlc=True)
            xp = np.zeros(s.n_rows_padded); xp[:s.n_rows] = pkg.apply_permutation(1.0 + 1e-3 * (np.arange(s.n_rows) % 1000), a["new_to_old_idx"])
            x = t.from_numpy(xp).cuda(); y = t.zeros(s.n_rows_padded, dtype=t.float64, device="cuda")
            pkg.spmv(A, x, y)
            ok = None if args.no_check else bool(np.array_equal(y.cpu().numpy(), orc.spmv_scs(32, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], xp)))
            ms = B.time_launches(0, args.reps, A=A, x=x, y=y)
            byts = s.n_elements * 12 + 8 * s.n_chunks + 8 * s.n_rows + 8 * s.n_rows_padded
            out = dict(config=2, workload=f"nlpkkt200-class stencil27 {g}^3 scs -c 32 -s 512 -dp", n=s.n_rows, nnz=s.nnz, b=1)
            flops = 2.0 * s.nnz
            ycpu = np.zeros(s.n_rows_padded)
            cpu_fn = lambda R: R.lib("colwise").ref_spmv_omp_scs_adv_f64(32, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], xp, ycpu)
            cpu_name = "spmv_omp_scs_adv<C=32,double>"
        elif cfg == "2k":   # the KKT-structured member of the nlpkkt class (uspmv_gen_kkt): rows of 5-28 entries, two index ranges N^3 apart
            N = int(200 * args.scale)
            coo = pkg.gen_kkt(N)
            s = prep(coo, pkg.F64)
            a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"]); a = s.arrays()
            A = pkg.DeviceMatrix(s, tlc=True)
            xp = np.zeros(s.n_rows_padded); xp[:s.n_rows] = pkg.apply_permutation(1.0 + 1e-3 * (np.arange(s.n_rows) % 1000), a["new_to_old_idx"])
            x = t.from_numpy(xp).cuda(); y = t.zeros(s.n_rows_padded, dtype=t.float64, device="cuda")
            pkg.spmv(A, x, y)
            ok = None if args.no_check else bool(np.array_equal(y.cpu().numpy(), orc.spmv_scs(32, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], xp)))
            ms = B.time_launches(0, args.reps, A=A, x=x, y=y)
            byts = s.n_elements * 12 + 8 * s.n_chunks + 8 * s.n_rows + 8 * s.n_rows_padded
            A0 = pkg.DeviceMatrix(s)
            out = dict(config="2k", workload=f"nlpkkt200-class KKT [H A^T; A 0] N={N} (n = 2N^3 + 6N^2) scs -c 32 -s 512 -dp", n=s.n_rows, nnz=s.nnz, b=1,
                       beta=round(s.nnz / s.n_elements, 5), plan_kind_tiles_planned=list(A.plan_info()), tlc_tiles_staged=[A.tlc_staged, A.tlc_tiles],
                       gather_kernel_ms=round(B.time_launches(0, args.reps, A=A0, x=x, y=y), 5))
            flops = 2.0 * s.nnz
            ycpu = np.zeros(s.n_rows_padded)
            cpu_fn = lambda R: R.lib("colwise").ref_spmv_omp_scs_adv_f64(32, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], xp, ycpu)
            cpu_name = "spmv_omp_scs_adv<C=32,double>"
        elif cfg == "3":
            g = int(111 * args.scale)
            coo = pkg.gen_stencil27(g, g, g, dof=3)
            vdt, ndt, tdt, vs = (pkg.F32, np.float32, t.float32, 4) if args.sp else (pkg.F64, np.float64, t.float64, 8)
            s = prep(coo, vdt)
            a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"]); a = s.arrays()
            b, ld = 8, s.n_rows_padded
            A = pkg.DeviceMatrix(s, block_tlc=b if not args.no_block_plan else 0)
            xp = np.zeros(ld, ndt); xp[:s.n_rows] = pkg.apply_permutation((1.0 + 1e-3 * (np.arange(s.n_rows) % 1000)).astype(ndt), a["new_to_old_idx"])
            res = {}
            for lay, nm in ((pkg.COLWISE, "colwise"), (pkg.ROWWISE, "rowwise")):
                X = np.zeros(b * ld, ndt)
                for v in range(b):
                    col = (xp * ndt(1.0 + v / 8.0)).astype(ndt)
                    if lay == pkg.ROWWISE: X[np.arange(ld) * b + v] = col
                    else: X[v * ld:(v + 1) * ld] = col
                dX = t.from_numpy(X).cuda(); dY = t.zeros(b * ld, dtype=tdt, device="cuda")
                pkg.spmmv(A, dX, dY, b, ld, lay)
                okl = None if args.no_check else bool(np.array_equal(dY.cpu().numpy(), orc.spmmv_scs(32, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], X, b, ld, lay == pkg.ROWWISE)))
                res[nm] = (B.time_launches(5, args.reps, A=A, x=dX, y=dY, b=b, ld=ld, layout=lay), okl)
            ms, ok = res["colwise"]
            byts = s.n_elements * (vs + 4) + 8 * s.n_chunks + b * vs * s.n_rows + b * vs * s.n_rows_padded
            out = dict(config=3, workload=f"Queen_4147-class stencil27 {g}^3 x 3 dof scs -c 32 -s 512 {'-sp' if args.sp else '-dp'} -block_vec_size 8", n=s.n_rows, nnz=s.nnz, b=b,
                       block_plan_tiles=[A.block_staged, A.block_tiles],
                       rowwise_ms=round(res["rowwise"][0], 5), rowwise_bitexact=res["rowwise"][1])
            flops = 2.0 * s.nnz * b
            Xc = np.zeros(b * ld, ndt)
            for v in range(b):
                Xc[v * ld:(v + 1) * ld] = (xp * ndt(1.0 + v / 8.0)).astype(ndt)
            Ycpu = np.zeros(b * ld, ndt)
            cpu_fn = lambda R: getattr(R.lib("colwise"), "ref_block_spmv_omp_scs_general_" + ("f32" if args.sp else "f64"))(
                32, s.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"], a["values"], Xc, Ycpu, b, ld)
            cpu_name = "block_spmv_omp_scs_general (colwise)"
        else:   # "4": stencil stand-in; "4b": the banded-random HV15R-class matrix of SURVEY.md 8(d)
            g = int(74 * args.scale)
            if cfg == "4b":
                nb = int(2017169 * args.scale ** 3)
                coo = pkg.gen_banded_random(nb, 140, 50000, magnitude_decades=10.0)
            else:
                coo = pkg.gen_stencil27(g, g, g, dof=5, magnitude_decades=10.0)
            dp, sp = pkg.partition_precisions(coo, 1e-3)
            ds = prep(dp, pkg.F64)
            perm = ds.arrays()["old_to_new_idx"].copy()
            ss = prep(sp, pkg.F32, fixed=perm)
            pkg.permute_scs_cols(ds, perm); pkg.permute_scs_cols(ss, perm)
            da, sa = ds.arrays(), ss.arrays()
            Ad, As = pkg.DeviceMatrix(ds), pkg.DeviceMatrix(ss)
            xp = np.zeros(ds.n_rows_padded); xp[:ds.n_rows] = pkg.apply_permutation(1.0 + 1e-3 * (np.arange(ds.n_rows) % 1000), da["new_to_old_idx"])
            x = t.from_numpy(xp).cuda(); y = t.zeros(ds.n_rows_padded, dtype=t.float64, device="cuda")
            pkg.spmv_ap(Ad, As, x, y)
            ok = None if args.no_check else bool(np.array_equal(y.cpu().numpy(), orc.spmv_scs_ap_adv(
                32, ds.n_chunks, (da["chunk_ptrs"], da["chunk_lengths"], da["col_idxs"], da["values"]),
                (sa["chunk_ptrs"], sa["chunk_lengths"], sa["col_idxs"], sa["values"]), xp)))
            ms_gather = B.time_launches(4, args.reps, A=Ad, B=As, x=x, y=y)
            nt_, ns_ = pkg.optimize_ap(Ad, As, ds, ss)
            y2 = t.zeros_like(y)
            pkg.spmv_ap(Ad, As, x, y2)
            ok = ok and bool(t.equal(y, y2)) if ok is not None else None
            ms = B.time_launches(4, args.reps, A=Ad, B=As, x=x, y=y)
            byts = 12 * ds.n_elements + 8 * ss.n_elements + 16 * ds.n_chunks + 8 * (ds.n_rows + ds.n_rows_padded)
            out = dict(config=cfg, workload=(f"HV15R-class banded-random n={coo.n_rows} 140/row band 50000" if cfg == "4b" else f"HV15R-class stencil27 {g}^3 x 5 dof") +
                       ", |a_ij| log-uniform over 10 decades, scs -c 32 -s 512 -ap[dp_sp] -ap_threshold_1 1e-3",
                       n=ds.n_rows, nnz=coo.nnz, dp_nnz=dp.nnz, sp_nnz=sp.nnz, dp_elements=ds.n_elements, sp_elements=ss.n_elements, b=1)
            # same matrix in plain dp for comparison
            s = prep(coo, pkg.F64); a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"])
            A = pkg.DeviceMatrix(s)
            out["gather_kernel_ms"] = round(ms_gather, 5); out["tlc_tiles_staged"] = [ns_, nt_]; out["ap_plan_kind_tiles_planned"] = list(Ad.plan_info())
            out["plain_dp_gather_ms"] = round(B.time_launches(0, args.reps, A=A, x=x, y=y), 5)
            y3 = t.zeros_like(y); pkg.spmv(A, x, y3)
            A.optimize(s)
            y4 = t.zeros_like(y); pkg.spmv(A, x, y4)
            out["plain_dp_plan_bitexact_vs_gather"] = bool(t.equal(y3, y4)); out["plain_dp_plan_kind_tiles_planned"] = list(A.plan_info())
            out["plain_dp_tlc_ms"] = round(B.time_launches(0, args.reps, A=A, x=x, y=y), 5)
            out["plain_dp_plan_frac_of_8TBs"] = round((s.n_elements * 12 + 8 * s.n_chunks + 8 * (s.n_rows + s.n_rows_padded)) / out["plain_dp_tlc_ms"] / 1e6 / 8000, 4)
            flops = 2.0 * coo.nnz
            ycpu = np.zeros(ds.n_rows_padded); yspc = np.zeros(ds.n_rows_padded, np.float32); xspc = xp.astype(np.float32)
            cpu_fn = lambda R: R.lib("colwise").ref_spmv_omp_scs_ap_adv(32, ds.n_chunks, da["chunk_ptrs"], da["chunk_lengths"], da["col_idxs"], da["values"], xp, ycpu,
                                                                        sa["chunk_ptrs"], sa["chunk_lengths"], sa["col_idxs"], sa["values"], xspc, yspc)
            cpu_name = "spmv_omp_scs_ap_adv<C=32>"
        if args.cpu_seconds > 0:
            from oracle import refshim
            import bench as _bench
            if refshim.available("colwise"):
                cores = _bench.usable_cores(); _bench.set_omp_threads(cores)
                cpu_fn(refshim)
                reps, tc = 0, time.perf_counter()
                while reps < 2 or time.perf_counter() - tc < args.cpu_seconds:
                    cpu_fn(refshim); reps += 1
                el = time.perf_counter() - tc
                out["cpu_baseline"] = dict(value=round(flops * reps / el / 1e9, 2), unit="GFLOP/s", cores=cores, kind="reference",
                                           sample=f"whole matrix, {reps} calls of {cpu_name} in {el:.1f} s, OMP threads = {cores}")
        out.update(kernel_ms=round(ms, 5), gflops=round(flops / ms / 1e6, 1), algorithmic_GBs=round(byts / ms / 1e6, 1),
                   frac_of_8TBs=round(byts / ms / 1e6 / 8000, 4), bitexact_vs_oracle=ok, setup_s=round(time.time() - t0, 1))
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
