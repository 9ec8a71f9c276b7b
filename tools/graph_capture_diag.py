#!/usr/bin/env python3
"""Diagnosis of the hipStreamEndCapture crash of round 2 (profiles/r02/dist_graph_capture.txt): the captured distributed step
(uspmv_dist_run, csrc/uspmv_dist_api.hip) in RCCL loopback, once per VARIANT, every variant in its own child process (a crash
ends only that child), native call stack printed by uspmv_debug_backtrace_on_crash.

    python tools/graph_capture_diag.py            # all variants -> stdout
Variants:
  torch                 libuspmv.so inside a Python process that imported torch first (binds to torch/lib: HIP 7.0, RCCL 2.26)
  torch_no_overlap      same, the step on ONE stream (no fork / join through events in the capture)
  torch_skip_rccl       same, two streams, but the RCCL group left out of the step (kernels + events only)
  torch_global / torch_threadlocal   other hipStreamCaptureModes
  torch_ba_synch        with the per-step all-reduce in the capture
  notorch               no torch in the process: libuspmv.so binds to /opt/rocm (HIP 7.2, RCCL 2.27), device memory via hipMalloc
  preload               torch imported, but /opt/rocm's libamdhip64 + librccl pre-loaded (LD_PRELOAD) so that the WHOLE process,
                        torch included, runs on HIP 7.2 / RCCL 2.27
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPE, P, RANK = (24, 24, 24), 2, 1


def child(variant):
    os.environ["USPMV_BACKTRACE"] = "1"
    os.environ["USPMV_VERBOSE"] = "1"
    import numpy as np
    if variant == "notorch":
        os.environ["USPMV_NO_TORCH"] = "1"
    import __graft_entry__ as ge
    pkg = ge.load_package()
    L = pkg.lib()
    print("versions (hip build, hip runtime, rccl build, rccl runtime):", pkg.runtime_versions(), flush=True)
    with open("/proc/self/maps") as f:
        libs = sorted({ln.split()[-1] for ln in f if ("libamdhip64" in ln or "librccl" in ln or "libhsa-runtime" in ln)})
    print("loaded:", libs, flush=True)
    counts = pkg.gen_stencil27_row_counts(*SHAPE)
    wsa = pkg.seg_from_row_counts(counts, "seg-rows", P)
    loc = pkg.gen_stencil27(*SHAPE, row_begin=int(wsa[RANK]), row_end=int(wsa[RANK + 1]))
    if variant == "notorch":
        hip = C.CDLL("libamdhip64.so.7")
        vp = C.c_void_p
        idb = (C.c_ubyte * 128)()
        assert L.uspmv_comm_unique_id(idb) == 0
        h = vp()
        w = np.ascontiguousarray(wsa, np.int32)
        rc = L.uspmv_dist_create_from_coo_ex(idb, 0, 1, RANK, P, loc.h, w.ctypes.data, 32, 512, 0, 1, None, C.byref(h))
        assert rc == 0, L.uspmv_last_error()
        m = (C.c_int64 * 12)()
        L.uspmv_dist_info(h, m)
        ld = int(m[2])
        dx, dy, st = vp(), vp(), vp()
        assert hip.hipMalloc(C.byref(dx), 8 * ld) == 0 and hip.hipMalloc(C.byref(dy), 8 * ld) == 0
        assert hip.hipStreamCreate(C.byref(st)) == 0
        bad, cs = C.c_int64(), C.c_double()
        rc = L.uspmv_dist_check(h, loc.h, w.ctypes.data, dx, dy, 1, st, C.byref(bad), C.byref(cs))
        assert rc == 0, L.uspmv_last_error()
        rc = L.uspmv_dist_run(h, dx, dy, 10, 1, st)
        assert rc == 0, L.uspmv_last_error()
        hip.hipStreamSynchronize(st)
        L.uspmv_dist_info(h, m)
        print(f"RESULT {variant}: graph captured = {bool(m[9])}, graph launches = {int(m[10])}, check mismatches = {bad.value}", flush=True)
        return
    import torch
    torch.cuda.set_device(0)
    d = pkg.DistNative(loc, wsa, 32, 512, RANK, P, pkg.comm_unique_id(), comm_rank=0, comm_size=1)
    if variant == "torch_no_overlap":
        d.set_option("overlap", 0)
    if variant == "torch_skip_rccl":
        d.set_option("diag_skip_exchange", 1)
    if variant == "torch_global":
        d.set_option("capture_mode", 0)
    if variant == "torch_threadlocal":
        d.set_option("capture_mode", 1)
    if variant == "torch_ba_synch":
        d.set_option("ba_synch", 1)
    x, y = d.new_x(np.zeros(d.n_local)), d.new_y()
    bad, _ = d.check(loc, x, y, use_graph=True)
    d.run(x, y, 10, use_graph=True)
    d.synchronize()
    d._refresh()
    print(f"RESULT {variant}: graph captured = {d.graph_captured}, graph launches = {d.graph_launches}, check mismatches = {bad}", flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    variants = sys.argv[1:] or ["notorch", "torch_skip_rccl", "torch_no_overlap", "torch", "torch_global", "torch_threadlocal", "torch_ba_synch", "preload"]
    for v in variants:
        env = dict(os.environ)
        if v == "preload":
            env["LD_PRELOAD"] = "/opt/rocm/lib/libamdhip64.so.7:/opt/rocm/lib/librccl.so.1"
        print(f"\n================ variant {v}", flush=True)
        try:
            # (no `timeout` wrapper: with LD_PRELOAD it would carry the HIP / RCCL runtime and then exec python -- an exec hop from a process
            #  with the GPU runtime loaded, which this pool forbids; subprocess.run's own timeout ends a hung child)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "preload_child" if v == "preload" else v],
                               env=env, capture_output=True, text=True, timeout=200)
            out = (r.stdout + r.stderr)
            keep = [ln for ln in out.splitlines() if not ln.startswith("  File ") and "site-packages" not in ln]
            print("\n".join(keep[-60:]))
            print(f"---- exit code {r.returncode}" + (" (killed by signal)" if r.returncode < 0 or r.returncode > 128 else ""), flush=True)
        except subprocess.TimeoutExpired:
            print("---- TIMEOUT", flush=True)


if __name__ == "__main__":
    main()
