"""The tracked tree builds: the in-tree artefacts are up to date with their sources (`make -q`), and a HIP source really goes through
hipcc for gfx950 from scratch (one small kernel file into a temporary object) with the Makefile's flags."""
import os
import subprocess

from conftest import ROOT

PKG = os.path.join(ROOT, "ultimate-spmv_amd")


def test_in_tree_build_is_current(pkg):
    r = subprocess.run(["make", "-q", "-C", PKG, "libuspmv.so", "uspmv"], capture_output=True, text=True)
    assert r.returncode == 0, "libuspmv.so / uspmv are older than their sources: run __graft_entry__.build()\n" + r.stdout + r.stderr


def test_hip_source_compiles_for_gfx950(tmp_path):
    obj = tmp_path / "plan_kernels.o"
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-I", os.path.join(ROOT, "include"),
                        "-I", os.path.join(PKG, "host"), "-c", os.path.join(PKG, "csrc", "plan_kernels.hip"), "-o", str(obj)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and obj.stat().st_size > 10000, r.stderr[-2000:]
    sym = subprocess.run(["nm", "-C", str(obj)], capture_output=True, text=True).stdout
    assert "launch_plan_count" in sym and "launch_block_values_gather" in sym
