"""The `uspmv` command-line harness: reference flag names and checks (code/utilities.hpp:1047-1545) and the
spmv_bench.txt block of write_bench_to_file (code/write_results.hpp:42-157)."""
import os
import re
import subprocess

import pytest

from conftest import ROOT, mtx_path

EXE = os.path.join(ROOT, "ultimate-spmv_amd", "uspmv")


def run(args, cwd):
    return subprocess.run([EXE] + args, cwd=cwd, capture_output=True, text=True, timeout=300)


def test_cli_argument_checks(pkg, tmp_path):
    assert os.path.exists(EXE), "build() must produce ultimate-spmv_amd/uspmv"
    r = run([], tmp_path)
    assert r.returncode == 1 and "Usage: uspmv" in r.stderr
    m = mtx_path("FDM-2d-16")
    for args, msg in (([m, "scs", "-c", "0"], "chunk size must be >= 1"),
                      ([m, "scs", "-bogus"], "unknown argument"),
                      ([m, "ell"], "kernel format not recognized"),
                      ([m, "scs", "-hp"], "Half precision selected"),
                      ([m, "scs", "-seg_metis"], "seg-metis selected, but this is a single-rank run"),
                      ([m, "scs", "-ap[dp_sp]", "-block_vec_size", "2"], "SpMMV is not yet implemented for AP kernels"),
                      ([m, "scs", "-mode", "x"], "Only bench (b) and solve (s) modes"),
                      ([m, "scs", "-equilibrate", "2"], "You can only choose to equilibrate data"),
                      ([m, "scs", "-ap[dp_sp]", "-equilibrate", "1"], "undefined behaviour in the reference"),
                      ([m, "scs", "-block_vec_layout", "rowwise"], "Row-wise block vector layout selected")):
        r = run(args, tmp_path)
        assert r.returncode == 1 and msg in r.stderr, (args, r.stderr)


@pytest.mark.gpu
def test_cli_bench_and_solve(pkg, tmp_path):
    m = mtx_path("FDM-2d-16")
    r = run([m, "scs", "-c", "16", "-s", "512", "-mode", "b", "-dp", "-bench_time", "0.2"], tmp_path)   # BASELINE config 1
    assert r.returncode == 0, r.stderr
    assert "beta = 0.98701299" in r.stdout and "n_elements = 1232" in r.stdout
    txt = open(tmp_path / "spmv_bench.txt").read()
    assert "kernel: scs, block_vec_size: 1, C: 16 sigma: 512, beta: 0.98701299, block_vec_layout: colwise, data_type: double, revisions:" in txt
    assert re.search(r"Total Gflops:\s+Total Walltime:\s*\n-+\s+-+\s*\n[0-9.e+-]+\s+[0-9.e+-]+", txt)
    assert "Achieved GB/s:" in txt
    r = run([mtx_path("bcsstk13"), "scs", "-c", "32", "-s", "512", "-mode", "s", "-rev", "3", "-rand_x", "1"], tmp_path)
    assert r.returncode == 0 and "-> OK" in r.stdout, r.stdout + r.stderr
    r = run([mtx_path("bcsstk13"), "crs", "-mode", "b", "-sp", "-block_vec_size", "4", "-bench_time", "0.1"], tmp_path)
    assert r.returncode == 0 and "CRS SpMMV kernel selected" in r.stdout
    r = run([mtx_path("impcol_e"), "scs", "-c", "32", "-s", "512", "-ap[dp_sp]", "-ap_threshold_1", "1.0", "-bench_time", "0.1"], tmp_path)
    assert r.returncode == 0 and "ap[dp_sp]" in r.stdout and "data_type: ap[dp_sp], threshold: 1.00" in open(tmp_path / "spmv_bench.txt").read()
    r = run([mtx_path("bcsstk13"), "scs", "-c", "32", "-s", "512", "-mode", "s", "-equilibrate", "1", "-dropout", "1", "-dropout_threshold", "0.5"], tmp_path)
    assert r.returncode == 0 and "-> OK" in r.stdout, r.stdout + r.stderr
    r = run([str(tmp_path / "nope.mtx"), "scs", "-c", "4", "-s", "4"], tmp_path)
    assert r.returncode == 1 and "cannot open" in r.stderr
