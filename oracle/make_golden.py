#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- generates the golden vectors under tests/golden/ by running the
GENUINE reference (oracle/_ref, built from /root/reference/code by oracle/Makefile).

Run in the build container (needs /root/reference):   python oracle/make_golden.py
The outputs (small .npz / .json fixtures: inputs and expected outputs only) are committed; the
reference itself never is.  Matrices under tests/golden/matrices/ are the reference's own test
inputs (data files from /root/reference/matrices).

Flow per case = the reference's own non-MPI flow (code/main.cpp:1075-1334, :51-102):
  read_mtx -> convert_to_scs -> permute_scs_cols(old_to_new) -> x_perm = apply_permutation(x,
  new_to_old) -> kernel -> y_orig = apply_permutation(y, old_to_new).
"""
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import refshim as R  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
ADV_CS = (2, 4, 8, 16, 32, 64, 128)  # code/classes_structs.hpp:510-517


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_x(n, dtype=np.float64):
    return (1.0 + 1e-3 * (np.arange(n) % 1000).astype(np.float64)).astype(dtype)


def scs_case(mtx, Cc, sigma, dtype, x):
    """Full reference flow for one (C, sigma, dtype).  Returns dict of arrays."""
    scs = R.convert_to_scs(mtx, Cc, sigma, dtype)
    pre = scs.arrays()
    scs.permute_cols(pre["old_to_new_idx"])
    post = scs.arrays()
    xv = x.astype(post["values"].dtype)
    xp = np.zeros(scs.n_rows_padded, xv.dtype)
    xp[:scs.n_rows] = R.apply_permutation(xv, post["new_to_old_idx"])
    kind = "adv" if Cc in ADV_CS else "gen"
    y = R.spmv_scs(kind, Cc, scs.n_chunks, post["chunk_ptrs"], post["chunk_lengths"], post["col_idxs"],
                   post["values"], xp)
    y_orig = R.apply_permutation(y, post["old_to_new_idx"])
    return scs, pre, post, xp, y, y_orig


def gen_scs_goldens(mats):
    full = {  # matrix -> (C, sigma) stored with every array
        "FDM-2d-16": (16, 512), "impcol_e": (32, 512), "matrix1": (4, 8), "myBigMat": (2, 4),
        "mySymmMat": (2, 2), "matrix_band_klein": (8, 16), "bcsstk13": (32, 512),
    }
    grid_C = (1, 2, 4, 8, 10, 16, 32, 64, 128)
    grid_S = (1, 2, 3, 8, 10, 64, 512)
    grid = {}
    for name, path in mats.items():
        mtx = R.RefMtx.read(path)
        I, J, V = mtx.arrays()
        x = make_x(mtx.n_rows)
        if name in full:
            Cc, sg = full[name]
            out = dict(n_rows=mtx.n_rows, n_cols=mtx.n_cols, nnz=mtx.nnz, C=Cc, sigma=sg, I=I, J=J, vals=V, x=x)
            for dt in ("f64", "f32"):
                scs, pre, post, xp, y, yo = scs_case(mtx, Cc, sg, dt, x)
                out.update({f"{dt}_chunk_ptrs": post["chunk_ptrs"], f"{dt}_chunk_lengths": post["chunk_lengths"],
                            f"{dt}_col_idxs_pre": pre["col_idxs"], f"{dt}_col_idxs": post["col_idxs"],
                            f"{dt}_values": post["values"], f"{dt}_old_to_new": post["old_to_new_idx"],
                            f"{dt}_new_to_old": post["new_to_old_idx"], f"{dt}_x_perm": xp, f"{dt}_y_perm": y,
                            f"{dt}_y_orig": yo})
                out["n_elements"] = scs.n_elements
                out["n_chunks"] = scs.n_chunks
            np.savez_compressed(os.path.join(OUT, f"scs_{name}.npz"), **out)
            print(f"scs_{name}: C={Cc} sigma={sg} n_el={out['n_elements']} beta={mtx.nnz / out['n_elements']:.8f}")
        if name in ("FDM-2d-16", "impcol_e", "matrix1", "bcsstk13"):
            for Cc in grid_C:
                for sg in grid_S:
                    key = f"{name}|{Cc}|{sg}"
                    ent = {}
                    for dt in ("f64", "f32"):
                        scs, pre, post, xp, y, yo = scs_case(mtx, Cc, sg, dt, x)
                        ent[dt] = dict(n_elements=scs.n_elements, n_chunks=scs.n_chunks,
                                       chunk_lengths=sha(post["chunk_lengths"]),
                                       chunk_ptrs=sha(post["chunk_ptrs"]), col_idxs=sha(post["col_idxs"]),
                                       values=sha(post["values"]), old_to_new=sha(post["old_to_new_idx"]),
                                       y_orig=sha(yo), y_perm=sha(y))
                    grid[key] = ent
    with open(os.path.join(OUT, "scs_grid_sha1.json"), "w") as f:
        json.dump(grid, f, indent=0, sort_keys=True)
    print(f"grid: {len(grid)} (matrix, C, sigma) cases")


def gen_csr_goldens(mats):
    out = {}
    for name in ("FDM-2d-16", "impcol_e", "matrix1", "bcsstk13"):
        mtx = R.RefMtx.read(mats[name])
        x = make_x(mtx.n_rows)
        for dt in ("f64", "f32"):
            scs = R.convert_to_scs(mtx, 1, 1, dt)
            a = scs.arrays()
            scs.permute_cols(a["old_to_new_idx"])
            a = scs.arrays()
            xv = x.astype(a["values"].dtype)
            y = R.spmv_csr(scs.n_rows, a["chunk_ptrs"], a["col_idxs"], a["values"], xv)
            out[f"{name}_{dt}_y"] = y
            out[f"{name}_{dt}_row_ptrs"] = a["chunk_ptrs"]
            # SpMMV CRS, b=4, both layouts
            b = 4
            n = scs.n_rows
            for rowwise in (0, 1):
                X = block_x(xv, n, b, n, rowwise)
                Y = R.spmmv_csr(n, a["chunk_ptrs"], a["col_idxs"], a["values"], X, b, n, rowwise)
                out[f"{name}_{dt}_Yb4_{'row' if rowwise else 'col'}"] = Y
    np.savez_compressed(os.path.join(OUT, "csr.npz"), **out)
    print("csr goldens:", len(out), "arrays")


def block_x(xp, n_used, b, ld, rowwise):
    """Block vector with column v = xp * (1 + v/8) in the requested layout (length b*ld)."""
    X = np.zeros(b * ld, xp.dtype)
    for v in range(b):
        col = (xp[:n_used] * xp.dtype.type(1.0 + v / 8.0)).astype(xp.dtype)
        if rowwise:
            X[np.arange(n_used) * b + v] = col
        else:
            X[v * ld: v * ld + n_used] = col
    return X


def gen_spmmv_goldens(mats):
    out = {}
    for name, (Cc, sg) in {"FDM-2d-16": (16, 512), "impcol_e": (32, 512), "bcsstk13": (32, 512),
                           "matrix1": (10, 3)}.items():
        mtx = R.RefMtx.read(mats[name])
        x = make_x(mtx.n_rows)
        for dt in ("f64", "f32"):
            scs, pre, post, xp, y, yo = scs_case(mtx, Cc, sg, dt, x)
            ld = scs.n_rows_padded  # non-MPI: padded_vec_size = n_rows + scs_padding (main.cpp:1406-1412)
            for b in (2, 8):
                for rowwise in (0, 1):
                    X = block_x(xp, scs.n_rows_padded, b, ld, rowwise)
                    Y = R.spmmv_scs_general(Cc, scs.n_chunks, post["chunk_ptrs"], post["chunk_lengths"],
                                            post["col_idxs"], post["values"], X, b, ld, rowwise)
                    out[f"{name}_{dt}_b{b}_{'row' if rowwise else 'col'}_Y"] = Y
        out[f"{name}_C"] = Cc
        out[f"{name}_sigma"] = sg
    np.savez_compressed(os.path.join(OUT, "spmmv.npz"), **out)
    print("spmmv goldens:", len(out), "arrays")


def gen_ap_goldens(mats):
    out = {}
    for name, (Cc, sg, th) in {"bcsstk13": (32, 512, 1e3), "impcol_e": (32, 512, 1.0),
                               "FDM-2d-16": (16, 512, 2.0), "matrix1": (10, 3, 1.0)}.items():
        mtx = R.RefMtx.read(mats[name])
        x = make_x(mtx.n_rows)
        dp_m, sp_h, (sI, sJ, sV) = R.partition_precisions_dpsp(mtx, th)
        dI, dJ, dV = dp_m.arrays()
        # reference flow: dp sorted by its own row lengths, sp forced to dp's permutation
        # (code/main.cpp:1170-1175)
        dps = R.convert_to_scs(dp_m, Cc, sg, "f64")
        da = dps.arrays()
        sps = R.convert_to_scs(sp_h, Cc, sg, "f32", fixed_perm=da["old_to_new_idx"])
        sa = sps.arrays()
        pfx = f"{name}_"
        out.update({pfx + "C": Cc, pfx + "sigma": sg, pfx + "th": th, pfx + "dp_I": dI, pfx + "dp_J": dJ,
                    pfx + "dp_V": dV, pfx + "sp_I": sI, pfx + "sp_J": sJ, pfx + "sp_V": sV,
                    pfx + "dp_chunk_ptrs": da["chunk_ptrs"], pfx + "dp_chunk_lengths": da["chunk_lengths"],
                    pfx + "dp_col_idxs_pre": da["col_idxs"], pfx + "dp_values": da["values"],
                    pfx + "sp_chunk_ptrs": sa["chunk_ptrs"], pfx + "sp_chunk_lengths": sa["chunk_lengths"],
                    pfx + "sp_col_idxs_pre": sa["col_idxs"], pfx + "sp_values": sa["values"],
                    pfx + "old_to_new": da["old_to_new_idx"], pfx + "new_to_old": da["new_to_old_idx"],
                    pfx + "sp_old_to_new": sa["old_to_new_idx"], pfx + "x": x})
        kind = "adv" if Cc in ADV_CS else "gen"
        npad = dps.n_rows_padded
        # (1) the reference flow as shipped: columns of dp/sp structs are NOT permuted
        #     (code/main.cpp:1308-1332).  Only meaningful for uniform x; recorded with x = 5.0.
        x5 = np.full(npad, 5.0); x5[dps.n_rows:] = 0.0
        dpt = (da["chunk_ptrs"], da["chunk_lengths"], da["col_idxs"], da["values"])
        spt = (sa["chunk_ptrs"], sa["chunk_lengths"], sa["col_idxs"], sa["values"])
        y5 = R.spmv_scs_ap(kind, Cc, dps.n_chunks, dpt, spt, x5)
        out[pfx + "y5_orig"] = R.apply_permutation(y5, da["old_to_new_idx"])
        # (2) corrected flow: permute_scs_cols applied to both structs with dp's permutation
        dps.permute_cols(da["old_to_new_idx"]); sps.permute_cols(da["old_to_new_idx"])
        da2, sa2 = dps.arrays(), sps.arrays()
        out[pfx + "dp_col_idxs"] = da2["col_idxs"]; out[pfx + "sp_col_idxs"] = sa2["col_idxs"]
        xp = np.zeros(npad); xp[:dps.n_rows] = R.apply_permutation(x, da["new_to_old_idx"])
        dpt = (da2["chunk_ptrs"], da2["chunk_lengths"], da2["col_idxs"], da2["values"])
        spt = (sa2["chunk_ptrs"], sa2["chunk_lengths"], sa2["col_idxs"], sa2["values"])
        for k in ("adv", "gen"):
            if k == "adv" and Cc not in ADV_CS + (1, 256):
                continue
            y = R.spmv_scs_ap(k, Cc, dps.n_chunks, dpt, spt, xp)
            out[pfx + f"y_perm_{k}"] = y
            out[pfx + f"y_orig_{k}"] = R.apply_permutation(y, da["old_to_new_idx"])
        out[pfx + "x_perm"] = xp
        # CRS dp+sp (C=1, sigma=1)
        d1 = R.convert_to_scs(dp_m, 1, 1, "f64"); a1 = d1.arrays()
        s1 = R.convert_to_scs(sp_h, 1, 1, "f32", fixed_perm=a1["old_to_new_idx"]); b1 = s1.arrays()
        ycsr = R.spmv_csr_apdpsp(mtx.n_rows, (a1["chunk_ptrs"], a1["col_idxs"], a1["values"]),
                                 (b1["chunk_ptrs"], b1["col_idxs"], b1["values"]), x)
        out[pfx + "csr_y"] = ycsr
        out[pfx + "csr_dp_row_ptrs"] = a1["chunk_ptrs"]; out[pfx + "csr_sp_row_ptrs"] = b1["chunk_ptrs"]
        print(f"ap {name}: th={th} dp nnz={dp_m.nnz} ({dps.n_elements} elts) sp nnz={len(sV)} ({sps.n_elements} elts)")
    np.savez_compressed(os.path.join(OUT, "ap.npz"), **out)


def gen_halo_goldens(mats):
    """Fake-rank emulation of the MPI set-up (code/main.cpp:1104-1108, :1128, :1271-1308) plus an
    in-process halo exchange, kernel and gather; y_global must equal the 1-rank y_orig."""
    out = {}
    meta = {}
    cases = [("bcsstk13", 32, 512, "seg-nnz", 4), ("bcsstk13", 32, 512, "seg-rows", 2),
             ("bcsstk13", 32, 512, "seg-nnz", 8), ("FDM-2d-16", 16, 512, "seg-nnz", 3),
             ("FDM-2d-16", 4, 8, "seg-rows", 4), ("impcol_e", 8, 16, "seg-nnz", 2),
             ("matrix1", 10, 3, "seg-rows", 2)]
    for name, Cc, sg, method, P in cases:
        mtx = R.RefMtx.read(mats[name], "mpi")
        xg = make_x(mtx.n_rows)
        wsa = R.seg_work_sharing_arr(mtx, method, P)
        key = f"{name}_C{Cc}_s{sg}_{method}_P{P}"
        out[key + "_wsa"] = wsa
        ranks = []
        for r in range(P):
            loc = R.seg_local_mtx(mtx, wsa, r)
            scs = R.convert_to_scs(loc, Cc, sg, "f64")
            nh, recv_idxs, cumsum = R.collect_local_needed_heri(scs, wsa, r, P)
            a = scs.arrays()
            scs.permute_cols(a["old_to_new_idx"])
            a = scs.arrays()
            ranks.append((loc, scs, a, nh, recv_idxs, cumsum))
        ys = []
        m = dict(n_local=[], nnz=[], n_elements=[], n_halo=[], recv_counts=[])
        for r in range(P):
            loc, scs, a, nh, recv_idxs, cumsum = ranks[r]
            n_local = int(wsa[r + 1] - wsa[r])
            pad = max(scs.n_rows_padded - scs.n_rows, nh)
            xl = np.zeros(n_local + pad)
            xl[:n_local] = R.apply_permutation(xg[wsa[r]:wsa[r + 1]].copy(), a["new_to_old_idx"], "mpi")
            for s in range(P):  # what the owner packs: x_s_perm[perm_s[idx]] == x_global[wsa[s] + idx]
                for k, idx in enumerate(recv_idxs[s]):
                    xl[n_local + cumsum[s] + k] = xg[wsa[s] + idx]
            kind = "adv" if Cc in ADV_CS else "gen"
            y = R.spmv_scs(kind, Cc, scs.n_chunks, a["chunk_ptrs"], a["chunk_lengths"], a["col_idxs"],
                           a["values"], xl, "mpi")
            ys.append(R.apply_permutation(y, a["old_to_new_idx"], "mpi"))
            out[f"{key}_r{r}_col_idxs"] = a["col_idxs"]
            out[f"{key}_r{r}_old_to_new"] = a["old_to_new_idx"]
            out[f"{key}_r{r}_chunk_lengths"] = a["chunk_lengths"]
            out[f"{key}_r{r}_recv_cumsum"] = cumsum
            out[f"{key}_r{r}_recv_idxs"] = np.concatenate(recv_idxs) if nh else np.zeros(0, np.int32)
            out[f"{key}_r{r}_x_local"] = xl
            m["n_local"].append(n_local); m["nnz"].append(loc.nnz); m["n_elements"].append(scs.n_elements)
            m["n_halo"].append(nh); m["recv_counts"].append([len(v) for v in recv_idxs])
        out[key + "_y_global"] = np.concatenate(ys)
        meta[key] = m
        print(key, "wsa", wsa.tolist(), "halo", m["n_halo"])
    np.savez_compressed(os.path.join(OUT, "halo.npz"), **out)
    with open(os.path.join(OUT, "halo_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


SOLVE_MATS = ("FDM-2d-16", "matrix1", "impcol_e")          # scripts/validate_master.sh:16-23
SOLVE_CS = (4, 8, 10, 16, 32, 64)
SOLVE_SIGMAS = (1, 2, 3, 4, 8, 10, 16, 32, 64)


def gen_solve_goldens(mats):
    """Solve mode (-mode s -rev 3, code/main.cpp:528-607: three times {y = A x; swap x <-> y}, result copied back in original
    row order) over the grid of scripts/validate_master.sh:16-23 -- matrices, C, sigma, crs + scs, dp / sp, -rand_x {0, 1} -- with
    the reference's own kernels: spmv_omp_scs_adv for C in {2,...,128}, spmv_omp_scs otherwise, spmv_omp_csr for crs.  x0 as the
    harness makes it: 5.0 (DefaultValues) or random_init's default-seeded mt19937 draw in [matrix_min, matrix_max]."""
    out = {}
    rev = 3
    for name in SOLVE_MATS:
        mtx = R.RefMtx.read(mats[name])
        I, J, V = mtx.arrays()
        n = mtx.n_rows
        vmin, vmax = float(np.abs(V).min()), float(np.abs(V).max())      # extract_matrix_min_mean_max (code/utilities.hpp:2502-2540)
        for dt, npdt in (("f64", np.float64), ("f32", np.float32)):
            for rx in (0, 1):
                cases = [("crs", 1, 1)] + [("scs", Cc, sg) for Cc in SOLVE_CS for sg in SOLVE_SIGMAS]
                for fmt, Cc, sg in cases:
                    scs = R.convert_to_scs(mtx, Cc, sg, dt)
                    scs.permute_cols(scs.arrays()["old_to_new_idx"])
                    a = scs.arrays()
                    npad = scs.n_rows_padded
                    # random_init draws one value per element of the padded vector; the padding is zeroed afterwards
                    x0 = R.random_init(vmin, vmax, npad, npdt)[:n] if rx else np.full(n, 5.0, npdt)
                    x = np.zeros(npad, npdt)
                    x[:n] = R.apply_permutation(x0, a["new_to_old_idx"])
                    for _ in range(rev):
                        if fmt == "crs":
                            y = np.zeros(npad, npdt)
                            y[:n] = R.spmv_csr(n, a["chunk_ptrs"], a["col_idxs"], a["values"], x)
                        else:
                            y = R.spmv_scs("adv" if Cc in ADV_CS else "gen", Cc, scs.n_chunks, a["chunk_ptrs"], a["chunk_lengths"],
                                           a["col_idxs"], a["values"], x)
                        x = y
                    out[f"{name}_{fmt}_C{Cc}_s{sg}_{dt}_r{rx}"] = R.apply_permutation(x, a["old_to_new_idx"])
        out[name + "_minmax"] = np.array([vmin, vmax])
    np.savez_compressed(os.path.join(OUT, "solve.npz"), **out)
    print("solve goldens:", len(out), "vectors")


def gen_reference_unit_fixtures():
    """The reference's own hand-written unit-test expectations (code/test_suite/test_data/M1.cpp,
    M_big.cpp, M0.cpp): numbers only, re-encoded as JSON {object name: {field: [numbers]}}."""
    res = {}
    for fn in ("M0.cpp", "M1.cpp", "M_big.cpp"):
        txt = open(os.path.join(REF, "code/test_suite/test_data", fn)).read()
        for m in re.finditer(r"(MtxData|ScsExplicitData)<(\w+),\s*int>\s+(\w+)\s*\{(.*?)\n\};", txt, re.S):
            kind, vt, name, body = m.groups()
            body = re.sub(r"//[^\n]*", "", body)
            vecs = [[float(t) for t in v.split(",") if t.strip()]
                    for v in re.findall(r"std::vector<\w+>\s*\{([^}]*)\}", body)]
            if kind == "MtxData":
                scal = re.findall(r"^\s*(\w+),", body, re.M)[:5]
                res[name] = dict(kind="coo", vt=vt, n_rows=int(scal[0]), n_cols=int(scal[1]), nnz=int(scal[2]),
                                 I=[int(v) for v in vecs[0]], J=[int(v) for v in vecs[1]], values=vecs[2])
            else:
                res[name] = dict(kind="scs_explicit", vt=vt, chunk_ptrs=[int(v) for v in vecs[0]],
                                 chunk_lengths=[int(v) for v in vecs[1]], col_idxs=[int(v) for v in vecs[2]],
                                 values=vecs[3], old_to_new_idx=[int(v) for v in vecs[4]],
                                 new_to_old_idx=[int(v) for v in vecs[5]])
        # (C, sigma) of every expected ScsData object
        for m in re.finditer(r"ScsData<(\w+),\s*int>\s+(\w+)\s*\{\s*(\d+),[^\n]*\n\s*(\d+),", txt):
            res.setdefault("_c_sigma", {})[m.group(2)] = [int(m.group(3)), int(m.group(4))]
    with open(os.path.join(OUT, "reference_unit_fixtures.json"), "w") as f:
        json.dump(res, f, indent=0, sort_keys=True)
    print("reference unit fixtures:", len(res) - 1, "objects")


def main():
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    os.makedirs(os.path.join(OUT, "matrices"), exist_ok=True)
    mats = {}
    for fn in sorted(os.listdir(os.path.join(REF, "matrices"))):
        if fn.endswith(".mtx"):
            dst = os.path.join(OUT, "matrices", fn)
            shutil.copyfile(os.path.join(REF, "matrices", fn), dst)
            mats[fn[:-4]] = dst
    gen_reference_unit_fixtures()
    gen_scs_goldens(mats)
    gen_csr_goldens(mats)
    gen_spmmv_goldens(mats)
    gen_ap_goldens(mats)
    gen_solve_goldens(mats)
    if R.available("mpi"):
        gen_halo_goldens(mats)
    else:
        print("WARNING: oracle/_ref/libuspmv_ref_mpi.so missing -> halo goldens not regenerated")


if __name__ == "__main__":
    main()
