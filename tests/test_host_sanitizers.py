"""AddressSanitizer + UBSan pass over the host data layer (reader, conversion, planners incl. the sweep and phased plans, halo set-up,
chunk classes, the self-check's reference rows, graph partition, generators, host communicator, exchange plan, cache): the
driver tools/sanitize_host.cpp is compiled together with host/*.cpp (no HIP) and run on the golden matrices.
GPU sanitizers are not available on the test pool, so this is the sanitizer coverage of the product."""
import os
import subprocess

import pytest

from conftest import ROOT, mtx_path

PKG = os.path.join(ROOT, "ultimate-spmv_amd")


def test_host_layer_is_clean_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "sanitize_host")
    srcs = [os.path.join(ROOT, "tools", "sanitize_host.cpp")] + [os.path.join(PKG, "host", f) for f in
            ("mtx_io.cpp", "scs_convert.cpp", "halo_plan.cpp", "gen_matrix.cpp", "tlc_plan.cpp", "sweep_plan.cpp", "graph_partition.cpp",
             "dist_check.cpp", "hostcomm.cpp", "comm_plan.cpp")]
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fopenmp", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "host")] + srcs + ["-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and ("asan" in r.stderr.lower() or "ubsan" in r.stderr.lower()):
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, OMP_NUM_THREADS="4", ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1")
    mats = [mtx_path(n) for n in ("FDM-2d-16", "bcsstk13", "impcol_e", "matrix1", "myBigMat", "mySymmMat")]
    r = subprocess.run([exe] + mats, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert r.stdout.count("ok ") == len(mats) and "ERROR" not in r.stderr and "runtime error" not in r.stderr
