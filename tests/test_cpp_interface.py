"""The C++ header include/uspmv_interface.hpp (names of the reference's interface.hpp) compiles with
g++ against libuspmv.so and its host half reproduces the golden SELL-C-sigma structure."""
import os
import subprocess

import numpy as np

from conftest import ROOT, golden, mtx_path

SRC = r'''
#include <cstdio>
#include "uspmv_interface.hpp"
int main(int argc, char **argv) {
    MtxData<double, int> m;
    read_mtx(argv[1], &m);
    ScsData<double, int> s;
    convert_to_scs<double, double, int>(&m, 32, 512, &s);
    permute_scs_cols<double, int>(&s, s.old_to_new_idx.data());
    std::vector<double> x(m.n_rows), xp(m.n_rows);
    for (long i = 0; i < m.n_rows; ++i) x[i] = 1.0 + 1e-3 * (i % 1000);
    apply_permutation<double, int>(xp.data(), x.data(), s.new_to_old_idx.data(), (int)m.n_rows);
    long h = 0;
    for (int c : s.col_idxs) h = h * 31 + c;
    double cs = 0; for (double v : xp) cs += v;
    printf("%ld %ld %ld %ld %.17g\n", s.n_chunks, s.n_elements, (long)s.old_to_new_idx[7], h, cs);
    MtxData<double, int> dp; MtxData<float, int> sp;
    partition_precisions(1e3, &m, &dp, &sp);
    printf("%ld %ld\n", dp.nnz, sp.nnz);
    try { ScsData<double, int> bad; convert_to_scs<double, double, int>(&m, 0, 1, &bad); }
    catch (const std::runtime_error &e) { printf("caught: %s\n", e.what()); }
    return 0;
}
'''


def test_interface_header_compiles_and_matches_golden(tmp_path, pkg):
    src = tmp_path / "t.cpp"
    src.write_text(SRC)
    exe = tmp_path / "t"
    libdir = os.path.join(ROOT, "ultimate-spmv_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-luspmv", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([str(exe), mtx_path("bcsstk13")], text=True).splitlines()
    g = golden("scs_bcsstk13.npz")
    n_chunks, n_el, o7, h, cs = out[0].split()
    assert (int(n_chunks), int(n_el)) == (int(g["n_chunks"]), int(g["n_elements"]))
    assert int(o7) == int(g["f64_old_to_new"][7])
    hh = 0
    for c in g["f64_col_idxs"].tolist():
        hh = (hh * 31 + c) & 0xFFFFFFFFFFFFFFFF
    assert int(h) & 0xFFFFFFFFFFFFFFFF == hh
    acc = 0.0
    for v in g["f64_x_perm"][:int(g["n_rows"])].tolist():
        acc += v
    assert float(cs) == acc
    a = golden("ap.npz")
    assert out[1].split() == [str(len(a["bcsstk13_dp_V"])), str(len(a["bcsstk13_sp_V"]))]
    assert out[2].startswith("caught: convert_to_scs")
