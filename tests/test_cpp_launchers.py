"""include/uspmv_launchers.hpp: launchers with exactly the reference's OnePrecFuncPtr / MultiPrecFuncPtr signatures
(code/classes_structs.hpp:283-333, the GPU-build form with n_thread_blocks), assigned to std::function objects of those very
types and called the way SpmvKernel::execute_one_prec / execute_two_prec call them (code/classes_structs.hpp:997-1075): every
array in device memory, C and n_chunks as DEVICE scalars (code/utilities.hpp:3803-3811).  y is compared with the reference's
golden y, bit for bit: scs dp / sp, crs, block vectors (b = 8, column-wise), ap[dp_sp] with the compile-time-C and the generic-C
numerics.  Calls are repeated: the second one runs on the cached handle and its device-built plan."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ADV_CS, ROOT, block_x, golden, mtx_path

SRC = r'''
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
#include <hip/hip_runtime_api.h>
#include "uspmv_launchers.hpp"

typedef long ST;
// the reference's types, verbatim shape (code/classes_structs.hpp:283-333, __CUDACC__ branch, no HAVE_HALF_MATH)
template <typename VT, typename IT>
using OnePrecFuncPtr = std::function<void(bool, const ST *, const ST *, const IT *, const IT *, const IT *, const VT *, VT *, VT *, int *, int *, const ST, const int *)>;
template <typename IT>
using MultiPrecFuncPtr = std::function<void(bool, const ST *, const ST *, const IT *, const IT *, const IT *, const double *, double *, double *,
                                            const ST *, const ST *, const IT *, const IT *, const IT *, const float *, float *, float *, const ST, const int *)>;

static std::vector<char> slurp(const std::string &p) {
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", p.c_str()); exit(2); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<char> b((size_t)n);
    if (n && fread(b.data(), 1, (size_t)n, f) != (size_t)n) exit(2);
    fclose(f);
    return b;
}
static void *to_dev(const std::vector<char> &b) {
    void *d = nullptr;
    if (hipMalloc(&d, b.size() ? b.size() : 4) != hipSuccess) exit(3);
    if (b.size() && hipMemcpy(d, b.data(), b.size(), hipMemcpyHostToDevice) != hipSuccess) exit(3);
    return d;
}
template <typename T> static T *dev_scalar(T v) { std::vector<char> b(sizeof(T)); memcpy(b.data(), &v, sizeof(T)); return (T *)to_dev(b); }
static void dump(const std::string &p, const void *d, size_t bytes) {
    std::vector<char> h(bytes);
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost) != hipSuccess) exit(4);
    FILE *f = fopen(p.c_str(), "wb"); fwrite(h.data(), 1, bytes, f); fclose(f);
}

template <typename VT>
static void one_prec(const std::string &dir, const std::string &pre, bool crs, long C, long n_chunks, long n_out, int b, int ld) {
    const int *cp = (const int *)to_dev(slurp(dir + "/" + pre + "cp.bin")), *cl = (const int *)to_dev(slurp(dir + "/" + pre + "cl.bin"));
    const int *ci = (const int *)to_dev(slurp(dir + "/" + pre + "ci.bin"));
    const VT *va = (const VT *)to_dev(slurp(dir + "/" + pre + "va.bin"));
    VT *x = (VT *)to_dev(slurp(dir + "/" + pre + (b > 1 ? "X.bin" : "x.bin")));
    VT *y = nullptr;
    if (hipMalloc((void **)&y, sizeof(VT) * (size_t)n_out) != hipSuccess) exit(3);
    const ST *dC = dev_scalar<ST>(C), *dN = dev_scalar<ST>(n_chunks);   // device scalars, as assign_spmv_kernel_gpu_data leaves them
    OnePrecFuncPtr<VT, int> f;
    if (crs) f = uspmv_launchers::spmv_hip_csr_launcher<VT, int>;
    else f = uspmv_launchers::spmv_hip_scs_launcher<VT, int>;
    int rank = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(y, 0xff, sizeof(VT) * (size_t)n_out);
        f(rep == 0, dC, dN, cp, cl, ci, va, x, y, &b, &ld, (ST)((n_chunks * C + 255) / 256), &rank);
        if (uspmv_stream_synchronize(nullptr) != USPMV_OK) exit(5);
        dump(dir + "/" + pre + "y" + std::to_string(rep) + ".bin", y, sizeof(VT) * (size_t)n_out);
    }
    printf("PLAN_KIND %d\n", uspmv_launchers::plan_kind(cp));
}

static void two_prec(const std::string &dir, long C, long n_chunks) {
    const int *dcp = (const int *)to_dev(slurp(dir + "/dp_cp.bin")), *dcl = (const int *)to_dev(slurp(dir + "/dp_cl.bin")), *dci = (const int *)to_dev(slurp(dir + "/dp_ci.bin"));
    const double *dva = (const double *)to_dev(slurp(dir + "/dp_va.bin"));
    const int *scp = (const int *)to_dev(slurp(dir + "/sp_cp.bin")), *scl = (const int *)to_dev(slurp(dir + "/sp_cl.bin")), *sci = (const int *)to_dev(slurp(dir + "/sp_ci.bin"));
    const float *sva = (const float *)to_dev(slurp(dir + "/sp_va.bin"));
    double *dx = (double *)to_dev(slurp(dir + "/ap_x.bin"));
    float *sx = (float *)to_dev(slurp(dir + "/ap_xs.bin"));
    const size_t n = (size_t)(C * n_chunks);
    double *dy; float *sy;
    if (hipMalloc((void **)&dy, 8 * n) != hipSuccess || hipMalloc((void **)&sy, 4 * n) != hipSuccess) exit(3);
    const ST *dC = dev_scalar<ST>(C), *dN = dev_scalar<ST>(n_chunks);
    MultiPrecFuncPtr<int> f = uspmv_launchers::spmv_hip_ap_scs_launcher<int>;
    int rank = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(dy, 0xff, 8 * n);
        f(false, dC, dN, dcp, dcl, dci, dva, dx, dy, dC, dN, scp, scl, sci, sva, sx, sy, (ST)((n + 255) / 256), &rank);
        if (uspmv_stream_synchronize(nullptr) != USPMV_OK) exit(5);
        dump(dir + "/ap_y" + std::to_string(rep) + ".bin", dy, 8 * n);
    }
}

int main(int argc, char **argv) {
    const std::string dir = argv[1], mode = argv[2];
    const long C = atol(argv[3]), n_chunks = atol(argv[4]);
    if (mode == "scs64") one_prec<double>(dir, "f64_", false, C, n_chunks, C * n_chunks, 1, (int)(C * n_chunks));
    else if (mode == "scs32") one_prec<float>(dir, "f32_", false, C, n_chunks, C * n_chunks, 1, (int)(C * n_chunks));
    else if (mode == "irr64") one_prec<double>(dir, "irr_", false, C, n_chunks, C * n_chunks, 1, (int)(C * n_chunks));
    else if (mode == "crs64") one_prec<double>(dir, "crs_", true, 1, n_chunks, n_chunks, 1, (int)n_chunks);
    else if (mode == "mm64") one_prec<double>(dir, "f64_", false, C, n_chunks, 8 * C * n_chunks, 8, (int)(C * n_chunks));
    else if (mode == "ap") two_prec(dir, C, n_chunks);
    else return 9;
    uspmv_launchers::release();
    printf("OK\n");
    return 0;
}
'''


def _dump(d, name, arr):
    np.ascontiguousarray(arr).tofile(os.path.join(d, name))


@pytest.mark.gpu
def test_function_pointer_launchers_match_golden(tmp_path, pkg):
    d = str(tmp_path)
    src = tmp_path / "l.cpp"
    src.write_text(SRC)
    exe = str(tmp_path / "l")
    libdir = os.path.join(ROOT, "ultimate-spmv_amd")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe,
                           "-L", libdir, "-luspmv", f"-Wl,-rpath,{libdir}"])
    g = golden("scs_bcsstk13.npz")
    C, nc = int(g["C"]), int(g["n_chunks"])
    n = C * nc

    def run(mode, Cc=C, ncc=nc):
        r = subprocess.run([exe, d, mode, str(Cc), str(ncc)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (mode, r.stdout, r.stderr)
        return r.stdout

    for dt in ("f64", "f32"):
        _dump(d, f"{dt}_cp.bin", g[f"{dt}_chunk_ptrs"]); _dump(d, f"{dt}_cl.bin", g[f"{dt}_chunk_lengths"])
        _dump(d, f"{dt}_ci.bin", g[f"{dt}_col_idxs"]); _dump(d, f"{dt}_va.bin", g[f"{dt}_values"]); _dump(d, f"{dt}_x.bin", g[f"{dt}_x_perm"])
    # ---- OnePrecFuncPtr, scs dp / sp
    run("scs64"); run("scs32")
    for rep in range(3):
        assert np.array_equal(np.fromfile(os.path.join(d, f"f64_y{rep}.bin"), np.float64), g["f64_y_perm"]), rep
        assert np.array_equal(np.fromfile(os.path.join(d, f"f32_y{rep}.bin"), np.float32), g["f32_y_perm"]), rep
    assert "PLAN_KIND 1" in run("scs64")               # bcsstk13: the tile-local-column plan, built on the device
    # ---- wide irregular rows (HV15R-class, SURVEY 8(d) at reduced size): the launcher's device-built plan is the column-window SWEEP
    #      (plan kind 2) -- no host struct, no copy of the matrix to the host -- and y is the oracle's, bit for bit
    from oracle import oracle as orc
    coo = pkg.gen_banded_random(120000, 140, 50000)
    si = pkg.convert_to_scs(coo, 32, 512, pkg.F64)
    ai = si.arrays(); pkg.permute_scs_cols(si, ai["old_to_new_idx"]); ai = si.arrays()
    xi = np.zeros(si.n_rows_padded); xi[:si.n_rows] = pkg.apply_permutation(1.0 + 1e-3 * (np.arange(si.n_rows) % 1000), ai["new_to_old_idx"])
    _dump(d, "irr_cp.bin", ai["chunk_ptrs"]); _dump(d, "irr_cl.bin", ai["chunk_lengths"]); _dump(d, "irr_ci.bin", ai["col_idxs"])
    _dump(d, "irr_va.bin", ai["values"]); _dump(d, "irr_x.bin", xi)
    out = run("irr64", 32, si.n_chunks)
    assert "PLAN_KIND 2" in out, out
    yo = orc.spmv_scs(32, si.n_chunks, ai["chunk_ptrs"], ai["chunk_lengths"], ai["col_idxs"], ai["values"], xi)
    for rep in range(3):
        assert np.array_equal(np.fromfile(os.path.join(d, f"irr_y{rep}.bin"), np.float64), yo), rep
    # ---- block vectors through the same launcher (b = 8, column-wise, vec_length = n_rows_padded)
    sp = golden("spmmv.npz")
    _dump(d, "f64_X.bin", block_x(g["f64_x_perm"], n, 8, n, 0))
    run("mm64")
    for rep in range(3):
        assert np.array_equal(np.fromfile(os.path.join(d, f"f64_y{rep}.bin"), np.float64), sp["bcsstk13_f64_b8_col_Y"]), rep
    # ---- crs (C = 1): structure from the host converter.  bcsstk13 as SELL-32-1 would carry 64 % padding, so the handle keeps the
    #      several-lanes-per-row CRS kernel (twin of the omp-simd spmv_omp_csr): the reference's own tolerance 1e-13 * sum|a x|
    m = pkg.read_mtx(mtx_path("bcsstk13"))
    s1 = pkg.convert_to_scs(m, 1, 1, pkg.F64)
    a1 = s1.arrays()
    _dump(d, "crs_cp.bin", a1["chunk_ptrs"]); _dump(d, "crs_cl.bin", a1["chunk_lengths"]); _dump(d, "crs_ci.bin", a1["col_idxs"])
    _dump(d, "crs_va.bin", a1["values"]); _dump(d, "crs_x.bin", g["x"].astype(np.float64))
    run("crs64", 1, s1.n_chunks)
    I, J, V = m.arrays()
    scale = np.bincount(I, weights=np.abs(V * g["x"][J]), minlength=m.n_rows)
    for rep in range(3):
        got = np.fromfile(os.path.join(d, f"crs_y{rep}.bin"), np.float64)
        assert np.all(np.abs(got - g["f64_y_orig"]) <= 1e-13 * scale), rep
    # ---- MultiPrecFuncPtr: ap[dp_sp], compile-time-C numerics (bcsstk13, C = 32) and generic-C numerics (matrix1, C = 10: float x)
    a = golden("ap.npz")
    for name in ("bcsstk13", "matrix1"):
        p = name + "_"
        Cc = int(a[p + "C"])
        _dump(d, "dp_cp.bin", a[p + "dp_chunk_ptrs"]); _dump(d, "dp_cl.bin", a[p + "dp_chunk_lengths"]); _dump(d, "dp_ci.bin", a[p + "dp_col_idxs"])
        _dump(d, "dp_va.bin", a[p + "dp_values"]); _dump(d, "sp_cp.bin", a[p + "sp_chunk_ptrs"]); _dump(d, "sp_cl.bin", a[p + "sp_chunk_lengths"])
        _dump(d, "sp_ci.bin", a[p + "sp_col_idxs"]); _dump(d, "sp_va.bin", a[p + "sp_values"])
        _dump(d, "ap_x.bin", a[p + "x_perm"]); _dump(d, "ap_xs.bin", a[p + "x_perm"].astype(np.float32))
        run("ap", Cc, len(a[p + "dp_chunk_lengths"]))
        want = a[p + ("y_perm_adv" if Cc in ADV_CS else "y_perm_gen")]
        for rep in range(2):
            assert np.array_equal(np.fromfile(os.path.join(d, f"ap_y{rep}.bin"), np.float64), want), (name, rep)


def test_launcher_header_compiles_without_hip(tmp_path, pkg):
    """The header needs nothing but uspmv.h: it compiles with plain g++ and its launchers convert to the reference's function types."""
    src = tmp_path / "c.cpp"
    src.write_text(r'''
#include <functional>
#include "uspmv_launchers.hpp"
typedef long ST;
int main() {
    std::function<void(bool, const ST *, const ST *, const int *, const int *, const int *, const double *, double *, double *, int *, int *, const ST, const int *)>
        f = uspmv_launchers::spmv_hip_scs_launcher<double, int>, g = uspmv_launchers::spmv_hip_csr_launcher<double, int>;
    std::function<void(bool, const ST *, const ST *, const int *, const int *, const int *, const float *, float *, float *, int *, int *, const ST, const int *)>
        h = uspmv_launchers::spmv_hip_scs_launcher<float, int>;
    std::function<void(bool, const ST *, const ST *, const int *, const int *, const int *, const double *, double *, double *, const ST *, const ST *,
                       const int *, const int *, const int *, const float *, float *, float *, const ST, const int *)> m = uspmv_launchers::spmv_hip_ap_scs_launcher<int>;
    return (f && g && h && m) ? 0 : 1;
}
''')
    libdir = os.path.join(ROOT, "ultimate-spmv_amd")
    exe = str(tmp_path / "c")
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe, "-L", libdir, "-luspmv",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    assert subprocess.run([exe]).returncode == 0
