"""The block (SpMMV) plan's host planner without a GPU: row orders (ties undone, balls, flat patches) are chunk-length-preserving
permutations whose copies keep every row's slots in order; phases tile a tile's slot groups, respect the 8-group / 256-row limits and
their one-list-per-phase indices decode to the entries' columns; flat patches + dynamic-programming cuts stage at most 0.8 of the
rows of the ties-undone / greedy plan on a 3-dof mesh and never more than the ties-undone plan elsewhere (tests/helpers/block_plan_check.cpp)."""
import os
import subprocess

from conftest import ROOT, mtx_path

PKG = os.path.join(ROOT, "ultimate-spmv_amd")


def test_block_plan_row_orders_and_phase_cuts(tmp_path):
    exe = str(tmp_path / "block_plan_check")
    srcs = [os.path.join(ROOT, "tests", "helpers", "block_plan_check.cpp")] + [os.path.join(PKG, "host", f) for f in
            ("mtx_io.cpp", "scs_convert.cpp", "gen_matrix.cpp", "tlc_plan.cpp", "debug.cpp")]
    cmd = ["g++", "-O2", "-std=c++17", "-fopenmp", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "host")] + srcs + ["-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, OMP_NUM_THREADS="4")
    r = subprocess.run([exe, mtx_path("bcsstk13"), mtx_path("impcol_e"), mtx_path("FDM-2d-16")], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1500:] + r.stderr[-3000:])
    assert r.stdout.rstrip().endswith("all ok") and r.stdout.count("ok ") >= 9 * 9
