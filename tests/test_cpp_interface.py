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


GPU_SRC = r'''
#include <cstdio>
#include <hip/hip_runtime_api.h>
#include "uspmv_interface.hpp"
int main(int argc, char **argv) {
    MtxData<double, int> m;
    read_mtx(argv[1], &m);
    std::vector<int> perm, inv;
    DeviceScs A = DeviceScs::from_mtx(m, 32, 512, false, &perm, &inv);          // GPU-side conversion + device-built plan
    const long np = A.n_rows_padded();
    std::vector<double> x(m.n_rows), xp(np, 0.0), y(np);
    for (long i = 0; i < m.n_rows; ++i) x[i] = 1.0 + 1e-3 * (i % 1000);
    apply_permutation<double, int>(xp.data(), x.data(), inv.data(), (int)m.n_rows);
    double *dx, *dy;
    if (hipMalloc((void **)&dx, 8 * np) || hipMalloc((void **)&dy, 8 * np)) return 2;
    hipMemcpy(dx, xp.data(), 8 * np, hipMemcpyHostToDevice);
    A.spmv(dx, dy);
    hipDeviceSynchronize();
    hipMemcpy(y.data(), dy, 8 * np, hipMemcpyDeviceToHost);
    FILE *f = fopen(argv[2], "wb"); fwrite(y.data(), 8, np, f); fclose(f);
    // the same arrays through the raw-array launcher of interface.hpp on a second, host-converted struct
    ScsData<double, int> s;
    convert_to_scs<double, double, int>(&m, 32, 512, &s);
    permute_scs_cols<double, int>(&s, s.old_to_new_idx.data());
    int *cp, *cl, *ci; double *va;
    hipMalloc((void **)&cp, 4 * (s.n_chunks + 1)); hipMalloc((void **)&cl, 4 * s.n_chunks); hipMalloc((void **)&ci, 4 * s.n_elements); hipMalloc((void **)&va, 8 * s.n_elements);
    hipMemcpy(cp, s.chunk_ptrs.data(), 4 * (s.n_chunks + 1), hipMemcpyHostToDevice); hipMemcpy(cl, s.chunk_lengths.data(), 4 * s.n_chunks, hipMemcpyHostToDevice);
    hipMemcpy(ci, s.col_idxs.data(), 4 * s.n_elements, hipMemcpyHostToDevice); hipMemcpy(va, s.values.data(), 8 * s.n_elements, hipMemcpyHostToDevice);
    DeviceScs W = DeviceScs::wrap<double, int>(32, s.n_chunks, s.n_elements, cp, cl, ci, va);
    hipMemset(dy, 0, 8 * np);
    W.spmv(dx, dy);
    hipDeviceSynchronize();
    std::vector<double> y2(np);
    hipMemcpy(y2.data(), dy, 8 * np, hipMemcpyDeviceToHost);
    // ... and from COO arrays that already live in HBM: the reference's tie order gives the same y_permuted; the stable device ordering the
    // same y in ORIGINAL row order
    int *dI, *dJ, *dperm, *dinv; double *dV;
    hipMalloc((void **)&dI, 4 * m.nnz); hipMalloc((void **)&dJ, 4 * m.nnz); hipMalloc((void **)&dV, 8 * m.nnz);
    hipMalloc((void **)&dperm, 4 * m.n_rows); hipMalloc((void **)&dinv, 4 * m.n_rows);
    hipMemcpy(dI, m.I.data(), 4 * m.nnz, hipMemcpyHostToDevice); hipMemcpy(dJ, m.J.data(), 4 * m.nnz, hipMemcpyHostToDevice);
    hipMemcpy(dV, m.values.data(), 8 * m.nnz, hipMemcpyHostToDevice);
    int ok3 = 1;
    for (int stable = 0; stable < 2; ++stable) {
        DeviceScs D = DeviceScs::from_device_coo(dI, dJ, dV, m.n_rows, m.n_cols, m.nnz, 32, 512, false, dperm, dinv, stable != 0);
        std::vector<int> p3(m.n_rows), i3(m.n_rows);
        hipMemcpy(p3.data(), dperm, 4 * m.n_rows, hipMemcpyDeviceToHost); hipMemcpy(i3.data(), dinv, 4 * m.n_rows, hipMemcpyDeviceToHost);
        std::vector<double> xq(np, 0.0), y3(np);
        for (long i = 0; i < m.n_rows; ++i) xq[p3[i]] = x[i];
        hipMemcpy(dx, xq.data(), 8 * np, hipMemcpyHostToDevice);
        hipMemset(dy, 0, 8 * np);
        D.spmv(dx, dy);
        hipDeviceSynchronize();
        hipMemcpy(y3.data(), dy, 8 * np, hipMemcpyDeviceToHost);
        if (!stable) ok3 = ok3 && (p3 == perm) && (y3 == y);
        for (long i = 0; i < m.n_rows; ++i) ok3 = ok3 && (y3[p3[i]] == y[perm[i]]);
    }
    printf("%d %d\n", (int)(y2 == y), ok3);
    return 0;
}
'''


import pytest  # noqa: E402


@pytest.mark.gpu
def test_device_class_of_the_cpp_header(tmp_path, pkg):
    """DeviceScs (RAII over the C ABI): GPU-side conversion + device-built plan, and wrap() around caller-owned device
    arrays; y equals the reference's golden y bit for bit."""
    src = tmp_path / "g.cpp"
    src.write_text(GPU_SRC)
    exe = tmp_path / "g"
    libdir = os.path.join(ROOT, "ultimate-spmv_amd")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-luspmv", f"-Wl,-rpath,{libdir}"])
    yfile = tmp_path / "y.bin"
    out = subprocess.check_output([str(exe), mtx_path("bcsstk13"), str(yfile)], text=True).split()
    g = golden("scs_bcsstk13.npz")
    y = np.fromfile(yfile, np.float64)
    assert np.array_equal(y, g["f64_y_perm"])
    assert out[-2:] == ["1", "1"]
