#!/usr/bin/env python3
"""The tile-local-column kernel with 16-bit against 12-bit local indices (`tlc_idx12`), alternating on one box, per matrix class:
kernel ms, fraction of 8 TB/s in algorithmic bytes, bits compared with the 16-bit result and with the gather kernel."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=60)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    import torch as t
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from ultimate_spmv_amd import binding as B
    t.cuda.set_device(0)
    classes = {
        "stencil27_253": lambda: pkg.gen_stencil27(253, 253, 253),
        "stencil27_304": lambda: pkg.gen_stencil27(304, 304, 304),
        "kkt_200": lambda: pkg.gen_kkt(200),
        "stencil27_111_dof3": lambda: pkg.gen_stencil27(111, 111, 111, dof=3),
        "stencil9_2d_4000": lambda: pkg.gen_stencil27(4000, 4000, 1),
        "banded_8M_12_300": lambda: pkg.gen_banded_random(8000000, 12, 300),
    }
    for name, gen in classes.items():
        if args.only and name not in args.only.split(","):
            continue
        coo = gen()
        s = pkg.convert_to_scs(coo, 32, 512, pkg.F64)
        a = s.arrays(); pkg.permute_scs_cols(s, a["old_to_new_idx"])
        x = t.rand(s.n_rows_padded, dtype=t.float64, device="cuda"); y = t.zeros_like(x)
        byts = s.n_elements * 12 + 8 * s.n_chunks + 8 * s.n_rows + 8 * s.n_rows_padded
        H = {}
        pkg.set_tuning(tlc_measure_tile=0)                       # (no on-the-spot verdict: both forms are wanted here)
        for v in (0, 1):
            pkg.set_tuning(tlc_idx12=2 if v else 0)
            H[v] = pkg.DeviceMatrix(s, tlc=True)
        pkg.set_tuning(tlc_idx12=1, tlc_measure_tile=1)
        Hm = pkg.DeviceMatrix(s, tlc=True)                        # what the library decides on its own (measured on the spot for >= 2^20 rows)
        auto = dict(bits=Hm.index_bits(), tile_rows=Hm.tile_rows)
        del Hm
        A0 = pkg.DeviceMatrix(s)                                  # gather kernel
        yg = t.zeros_like(x); pkg.spmv(A0, x, yg); del A0
        ys = {}
        for v in (0, 1):
            y.fill_(-1.0); pkg.spmv(H[v], x, y); ys[v] = y.clone()
        for v in (0, 1): B.time_launches(0, 20, A=H[v], x=x, y=y)
        ms = {0: [], 1: []}
        for _ in range(5):
            for v in (0, 1): ms[v].append(B.time_launches(0, args.reps, A=H[v], x=x, y=y))
        for v in (0, 1):
            m = sorted(ms[v])[2]
            print(json.dumps(dict(matrix=name, n=s.n_rows, nnz=s.nnz, idx12=v, tile_rows=H[v].tile_rows, kernel_ms_median=round(m, 5), kernel_ms_min=round(min(ms[v]), 5),
                                  frac=round(byts / (m * 1e-3) / 8e12, 4), same_as_16bit=bool(t.equal(ys[v], ys[0])), same_as_gather=bool(t.equal(ys[v], yg)), bits=H[v].index_bits(), library_choice=auto)), flush=True)
        del H, coo, s


if __name__ == "__main__":
    main()
