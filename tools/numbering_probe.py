#!/usr/bin/env python3
"""How much do the plans depend on a friendly row numbering?  The 27-point x 3 dof stencil (Queen_4147-class) with its NODES renumbered at random inside
consecutive blocks of K nodes (symmetric permutation; K = 0: the generator's numbering; larger K ~ an unstructured mesh whose numbering is only locally
coherent), SELL-32-512 dp: the tile-local-column SpMV plan (rows per tile, staged tiles, x lines, kernel time) and the phased SpMMV plan for b = 8
(phases, staged X rows per row, kernel time), fractions of the 8 TB/s roofline in algorithmic bytes.  Every result bit-identical to the lane-per-row kernels."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
from ultimate_spmv_amd import binding as B
torch.cuda.set_device(0)
if os.environ.get("TLC_ELEM") is not None: pkg.set_tuning(tlc_elem=int(os.environ["TLC_ELEM"]))
if os.environ.get("TLC_ELEM_SEG") is not None: pkg.set_tuning(tlc_elem_seg_rows=int(os.environ["TLC_ELEM_SEG"]))
if os.environ.get("TLC_ELEM_ROWS") is not None: pkg.set_tuning(tlc_elem_rows=int(os.environ["TLC_ELEM_ROWS"]))      # 1: balls for 256-row tiles, 4: the block plan's 64-row patches, 0: off
g = int(sys.argv[1]) if len(sys.argv) > 1 else 80
Ks = [int(k) for k in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,64,1000,20000".split(","))]
dof = int(os.environ.get("NUMBERING_DOF", "3"))
cols_only = os.environ.get("NUMBERING_COLS_ONLY") == "1"     # renumber the COLUMNS only (x in another numbering than the rows)
base = pkg.gen_stencil27(g, g, g, dof=dof)
I0, J0, V0 = (np.array(a) for a in base.arrays())
n = base.n_rows
del base
rng = np.random.default_rng(7)
for K in Ks:
    t0 = time.time()
    if K:
        nn = n // dof
        p = np.arange(nn, dtype=np.int64)
        for s0 in range(0, nn, K):
            seg = p[s0:s0 + K].copy(); rng.shuffle(seg); p[s0:s0 + K] = seg
        prow = I0.astype(np.int32) if cols_only else (p[I0 // dof] * dof + I0 % dof).astype(np.int32)
        pcol = (p[J0 // dof] * dof + J0 % dof).astype(np.int32)
        o = np.lexsort((pcol, prow))
        I, J, V = prow[o], pcol[o], V0[o]
        del prow, pcol, o
    else:
        I, J, V = I0, J0, V0
    m = pkg.Coo.from_arrays(n, n, I, J, V)
    s = pkg.convert_to_scs(m, 32, 512, pkg.F64); a = s.arrays()
    if not cols_only: pkg.permute_scs_cols(s, a["old_to_new_idx"])
    prep_s = time.time() - t0
    npad, nel, nch = s.n_rows_padded, int(a["chunk_ptrs"][-1]), s.n_chunks
    # ---- SpMV
    x = torch.rand(npad, dtype=torch.float64, device="cuda"); y = torch.zeros_like(x); yr = torch.zeros_like(x)
    pkg.set_tuning(tlc=0); A0 = pkg.DeviceMatrix(s); pkg.spmv(A0, x, yr); pkg.set_tuning(tlc=1)
    A = pkg.DeviceMatrix(s, tlc=True)
    pkg.spmv(A, x, y)
    same = bool(torch.equal(y[: s.n_rows], yr[: s.n_rows]))
    B.time_launches(0, 20, A=A, x=x, y=y)
    ms = min(B.time_launches(0, 40, A=A, x=x, y=y) for _ in range(3))
    byts = nel * 12 + 8 * nch + 8 * n + 8 * npad
    kind = dict(tiles=A.tlc_tiles, staged=A.tlc_staged, index_bits=A.index_bits(), elements_per_list_entry=A.plan_granularity())
    print(json.dumps(dict(K=K, cols_only=cols_only, tlc_elem=pkg.get_tuning("tlc_elem"), op="spmv", n=n, nnz=int(len(I)), prep_s=round(prep_s, 1), plan=kind, tile_rows=getattr(A, "tile_rows", None), bitexact=same, ms=round(ms, 4),
                          frac=round(byts / ms / 1e6 / 8000, 3))), flush=True)
    del A, A0
    if os.environ.get("NUMBERING_SPMV_ONLY") == "1":
        del m, s
        continue
    # ---- SpMMV b = 8, row-wise
    b, ld = 8, npad
    X = torch.rand(b * ld, dtype=torch.float64, device="cuda"); Y = torch.zeros_like(X); Yr = torch.zeros_like(X)
    pkg.set_tuning(spmmv_variant=3); A0 = pkg.DeviceMatrix(s); pkg.spmmv(A0, X, Yr, b, ld, pkg.ROWWISE); pkg.set_tuning(spmmv_variant=0); del A0
    Ab = pkg.DeviceMatrix(s, block_tlc=b)
    pkg.spmmv(Ab, X, Y, b, ld, pkg.ROWWISE)
    same = bool(torch.equal(Y, Yr))
    B.time_launches(5, 20, A=Ab, x=X, y=Y, b=b, ld=ld, layout=pkg.ROWWISE)
    ms = min(B.time_launches(5, 40, A=Ab, x=X, y=Y, b=b, ld=ld, layout=pkg.ROWWISE) for _ in range(3))
    info = Ab.block_plan_info()
    byts = nel * 12 + 8 * nch + 8 * b * n + 8 * b * npad
    print(json.dumps(dict(K=K, op="spmmv b=8 rowwise", phased=info["phased_plan"], phases_per_tile=round(info["phases"] / max(info["tiles"], 1), 2),
                          staged_rows_per_row=round(info["rows_staged"] / n, 2), bitexact=same, ms=round(ms, 4), frac=round(byts / ms / 1e6 / 8000, 3))), flush=True)
    del Ab, m, s
