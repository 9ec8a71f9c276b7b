"""Solve mode of the `uspmv` harness (-mode s: rev times {y = A x; swap x <-> y}, result in original row order; reference
code/main.cpp:528-607) against goldens made with the GENUINE reference kernels (oracle/make_golden.py::gen_solve_goldens ->
tests/golden/solve.npz) over the grid of the reference's validation script (scripts/validate_master.sh:16-23): matrices
FDM-2d-16 / matrix1 / impcol_e, C in {4,8,10,16,32,64}, sigma in {1,2,3,4,8,10,16,32,64}, crs + scs, dp / sp, -rev 3,
-rand_x {0,1}.  The reference validates this mode against MKL (code/write_results.hpp:442-556); here the independent result
is the reference's own CPU kernel.  SELL kernels: bit for bit (same FMA chain, three times over).  crs: the reference's loop is
omp-simd re-associated, so there is no canonical bit pattern: relative to the largest |y|, 1e-12 (dp) / 1e-5 (sp)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, golden, mtx_path

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "ultimate-spmv_amd", "uspmv")
MATS = ("FDM-2d-16", "matrix1", "impcol_e")
CS = (4, 8, 10, 16, 32, 64)
SIGMAS = (1, 2, 3, 4, 8, 10, 16, 32, 64)


def _cases():
    """Every C with four sigmas (rotating through the nine, so that all nine occur), both precisions, -rand_x 1; the default x
    for every C at one sigma; crs in all four (precision, x) combinations: 150 harness runs."""
    out = []
    for mi, name in enumerate(MATS):
        for ci, C in enumerate(CS):
            for k in range(4):
                sg = SIGMAS[(2 * ci + 3 * k + mi) % len(SIGMAS)]
                for dt in ("f64", "f32"):
                    out.append((name, "scs", C, sg, dt, 1))
            out.append((name, "scs", C, SIGMAS[(ci + mi) % len(SIGMAS)], "f64", 0))
        for dt in ("f64", "f32"):
            for rx in (0, 1):
                out.append((name, "crs", 1, 1, dt, rx))
    return sorted(set(out))


def _harness_runs(tmp_path, jobs, workers=4):
    """jobs: [(key, args without the -dump_y file)] -> {key: (CompletedProcess, y file)}; `workers` harness processes at a time (each is a
    process of its own on the GPU: start-up dominates its cost; the pool stays below the box's limit of concurrent GPU processes)."""
    from concurrent.futures import ThreadPoolExecutor

    def one(job):
        k, (key, args) = job
        d = tmp_path / f"run{k}"
        d.mkdir()
        yf = str(d / "y.bin")
        env = dict(os.environ, OMP_NUM_THREADS="2", OMP_WAIT_POLICY="passive")       # (tiny matrices: keep the pool from oversubscribing the cores)
        return key, (subprocess.run(args + ["-dump_y", yf], cwd=d, env=env, capture_output=True, text=True, timeout=180), yf)

    with ThreadPoolExecutor(workers) as ex:
        return dict(ex.map(one, enumerate(jobs)))


def test_solve_mode_grid_against_reference_goldens(pkg, tmp_path):
    g = golden("solve.npz")
    seen_sigma = set()
    n_exact = n_tol = 0
    jobs = []
    for case in _cases():
        name, fmt, C, sg, dt, rx = case
        jobs.append((case, [EXE, mtx_path(name), fmt] + (["-c", str(C), "-s", str(sg)] if fmt == "scs" else []) +
                     ["-mode", "s", "-rev", "3", "-rand_x", str(rx), "-dp" if dt == "f64" else "-sp", "-validate", "0"]))
    done = _harness_runs(tmp_path, jobs)
    for case in _cases():
        name, fmt, C, sg, dt, rx = case
        r, yf = done[case]
        assert r.returncode == 0 and "3 revision(s) done" in r.stdout, (case, r.stdout, r.stderr)
        want = g[f"{name}_{fmt}_C{C}_s{sg}_{dt}_r{rx}"]
        got = np.fromfile(yf, want.dtype)
        if fmt == "scs":
            assert np.array_equal(got, want), (name, C, sg, dt, rx, np.abs(got - want).max())
            seen_sigma.add(sg); n_exact += 1
        else:
            tol = 1e-12 if dt == "f64" else 1e-5
            assert np.all(np.abs(got - want) <= tol * np.abs(want).max()), (name, dt, rx)
            n_tol += 1
    assert seen_sigma == set(SIGMAS) and n_exact >= 120 and n_tol == 12


def test_solve_mode_with_the_conversion_on_the_device(pkg, tmp_path):
    """-convert device | device_stable: convert_to_scs + permute_scs_cols run on the device from the COO arrays in HBM
    (uspmv_convert_to_scs_device_from_arrays), the plan is built there too.  Same goldens, bit for bit -- the stable tie order changes the
    permuted numbering, not y in original row order; bench mode and block vectors run as well."""
    g = golden("solve.npz")
    jobs = []
    for name, fmt, C, sg, dt, rx in _cases():
        if fmt != "scs" or (C, sg) not in ((32, 64), (16, 32), (4, 3), (64, 2), (8, 8), (10, 10), (32, 16), (64, 64), (16, 4), (4, 1)):
            continue
        for conv in ("device", "device_stable"):
            jobs.append(((name, fmt, C, sg, dt, rx, conv), [EXE, mtx_path(name), fmt, "-c", str(C), "-s", str(sg), "-mode", "s", "-rev", "3", "-rand_x", str(rx),
                                                            "-dp" if dt == "f64" else "-sp", "-validate", "0", "-convert", conv]))
    assert len(jobs) >= 12
    for key, (r, yf) in _harness_runs(tmp_path, jobs).items():
        name, fmt, C, sg, dt, rx, conv = key
        assert r.returncode == 0 and "convert_to_scs on the device" in r.stdout, (key, r.stdout, r.stderr)
        want = g[f"{name}_{fmt}_C{C}_s{sg}_{dt}_r{rx}"]
        assert np.array_equal(np.fromfile(yf, want.dtype), want), key
    for extra in ([], ["-block_vec_size", "4"], ["-sp", "-block_vec_size", "8"]):
        r = subprocess.run([EXE, mtx_path("bcsstk13"), "scs", "-c", "32", "-s", "512", "-mode", "b", "-bench_time", "0.05", "-convert", "device_stable"] + extra,
                           cwd=tmp_path, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "Total Gflops" in r.stdout, (extra, r.stdout, r.stderr)
    r = subprocess.run([EXE, mtx_path("bcsstk13"), "scs", "-c", "32", "-s", "512", "-ap[dp_sp]", "-ap_threshold_1", "1", "-convert", "device"], cwd=tmp_path, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "one-precision" in (r.stdout + r.stderr)


def test_tune_flag_passes_library_tuning_keys(pkg, tmp_path):
    """-tune key=value (repeatable): the 12-bit / 16-bit index stream of the SpMV plan and the block plan's row order / phase cuts give the
    goldens' bits in solve mode, single and block vectors; an unknown key or a malformed pair ends the run with the library's message."""
    g = golden("solve.npz")
    name, C, sg = "impcol_e", 32, 64          # (the solve goldens hold FDM-2d-16 / matrix1 / impcol_e)
    jobs = []
    for tag, tune, extra in (("i12off", ["-tune", "tlc_idx12=0"], []), ("i12on", ["-tune", "tlc_idx12=2"], []),
                             ("ties", ["-tune", "spmmv_reorder=1", "-tune", "spmmv_phase_dp=0"], ["-block_vec_size", "8"]),
                             ("patches", ["-tune", "spmmv_reorder=4", "-tune", "spmmv_phase_dp=24"], ["-block_vec_size", "8"])):
        jobs.append((tag, [EXE, mtx_path(name), "scs", "-c", str(C), "-s", str(sg), "-mode", "s", "-rev", "3", "-rand_x", "1", "-dp", "-validate", "0"] + tune + extra))
    done = _harness_runs(tmp_path, jobs)
    want = g[f"{name}_scs_C{C}_s{sg}_f64_r1"]
    for tag in ("i12off", "i12on"):
        r, yf = done[tag]
        assert r.returncode == 0, (tag, r.stdout, r.stderr)
        assert np.array_equal(np.fromfile(yf, want.dtype), want), tag
    ys = {}
    for tag in ("ties", "patches"):
        r, yf = done[tag]
        assert r.returncode == 0, (tag, r.stdout, r.stderr)
        ys[tag] = np.fromfile(yf, np.float64)
    assert ys["ties"].size == 8 * want.size and np.array_equal(ys["ties"], ys["patches"])
    assert np.array_equal(ys["ties"][:want.size], want)      # (column 0 of the block vector draws the x of the single-vector run)
    for bad in (["-tune", "no_such_key=1"], ["-tune", "tlc_idx12"]):
        r = subprocess.run([EXE, mtx_path(name), "scs"] + bad, cwd=tmp_path, capture_output=True, text=True, timeout=60)
        assert r.returncode != 0 and ("unknown key" in (r.stdout + r.stderr) or "key=value" in (r.stdout + r.stderr)), (bad, r.stdout, r.stderr)
