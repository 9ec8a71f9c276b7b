"""Column-window sweep plan (ultimate-spmv_amd/host/sweep_plan.cpp) checked on the host: tests/cpp/sweep_plan_emulate.cpp
replays the plan the way the HIP kernel consumes it and compares with the reference's slot-ordered FMA chain, bit for bit
(dp and ap[dp_sp], several C / sigma / window / tile shapes, special values, rows with > 255 entries per window)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sweep_plan_replay_matches_fma_chain(pkg, tmp_path):
    libdir = os.path.dirname(pkg.library_path())
    exe = str(tmp_path / "sweep_plan_emulate")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(libdir, "host"),
                           os.path.join(ROOT, "tests", "cpp", "sweep_plan_emulate.cpp"), "-o", exe, "-L" + libdir, "-luspmv",
                           "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout + out.stderr
