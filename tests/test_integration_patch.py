"""INTEGRATION.md section 1 as a PATCH: the three assignments (and the function-pointer types' GPU-build arguments) applied to a copy
of the reference's code/classes_structs.hpp under /tmp, and the SpmvKernel class compiled with -DUSE_USPMV_HIP (host-only syntax
check: the launchers of include/uspmv_launchers.hpp must have exactly the std::function types the patched class declares).
Build-container only: needs /root/reference (never read on the GPU box) and g++."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

REF = "/root/reference/code"
GUARD = "#if defined(__CUDACC__) || defined(USE_USPMV_HIP)"


def _blocks(lines):
    """(start, end) line index pairs of every `#ifdef __CUDACC__` ... matching `#endif` block (nested conditionals respected)"""
    out, stack = [], []
    for i, ln in enumerate(lines):
        t = ln.strip()
        if t.startswith("#if"):
            stack.append((i, t.startswith("#ifdef __CUDACC__")))
        elif t.startswith("#endif") and stack:
            s, is_cuda = stack.pop()
            if is_cuda:
                out.append((s, i))
    return out


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")
def test_three_assignment_patch_compiles_into_spmvkernel(tmp_path):
    src = open(os.path.join(REF, "classes_structs.hpp")).read().split("\n")
    cuda_api = re.compile(r"cudaMemcpy|cudaMalloc|<<<|CUDA_CHECK|cusparse")
    n_switched = 0
    for s, e in _blocks(src):
        body = "\n".join(src[s:e + 1])
        if cuda_api.search(body):
            continue                      # device staging / debug dumps: the part of a HIP port INTEGRATION.md leaves to hipMalloc & co.
        src[s] = GUARD
        n_switched += 1
    text = "\n".join(src)
    # the three assignments of INTEGRATION.md section 1 (every scs / crs / ap[dp_sp] launcher of the GPU build -> a uspmv launcher)
    text, n1 = re.subn(r"\b(?:block_)?spmv_gpu_scs(?:_adv|_general)?_launcher<VT, IT>", "uspmv_launchers::spmv_hip_scs_launcher<VT, IT>", text)
    text, n2 = re.subn(r"\b(?:block_)?spmv_gpu_csr_launcher<VT, IT>", "uspmv_launchers::spmv_hip_csr_launcher<VT, IT>", text)
    text, n3 = re.subn(r"\bspmv_gpu_(?:ap_scs|scs_ap_adv|ap_scs_adv|ap_csr)\w*_launcher<IT>", "uspmv_launchers::spmv_hip_ap_scs_launcher<IT>", text)
    text, n4 = re.subn(r"cudaDeviceSynchronize\(\)", "uspmv_stream_synchronize(nullptr)", text)
    assert n_switched >= 15 and n1 >= 2 and n2 >= 1 and n3 >= 1 and n4 >= 3, (n_switched, n1, n2, n3, n4)
    assert not re.search(r"spmv_gpu_\w+_launcher", "\n".join(ln for ln in text.split("\n") if "//" not in ln.split("spmv_gpu")[0])), "a GPU launcher of the reference is still referenced"
    text = text.replace('#include "kernels.hpp"', '#include "uspmv_launchers.hpp"\n#include "kernels.hpp"', 1)
    d = tmp_path / "patched"
    d.mkdir()
    (d / "classes_structs.hpp").write_text(text)
    (d / "tu.cpp").write_text('#include "classes_structs.hpp"\n'
                              "template class SpmvKernel<double, int>;\ntemplate class SpmvKernel<float, int>;\nint main() { return 0; }\n")
    cmd = ["g++", "-std=c++14", "-fsyntax-only", "-fopenmp", "-DUSE_USPMV_HIP", "-DCOLWISE_BLOCK_VECTOR_LAYOUT", "-DSIMD_LENGTH=4", "-w",
           f"-I{d}", f"-I{REF}", f"-I{os.path.join(ROOT, 'include')}", str(d / "tu.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]
