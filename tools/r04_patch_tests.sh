#!/bin/bash
# round 4: the block plan's new defaults (flat row patches + dynamic-programming cuts) under the GPU tests that touch SpMMV, then the probe
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_spmmv_sweep.py tests/test_gpu_leaks.py -x -q -m gpu > gpurun_out/r04/patch_tests_1.txt 2>&1
echo "rc=$?" >> gpurun_out/r04/patch_tests_1.txt
tail -5 gpurun_out/r04/patch_tests_1.txt
python -m pytest tests/test_dist_native_gpu.py tests/test_cpp_interface.py tests/test_cpp_launchers.py tests/test_cli_solve_gpu.py -x -q -m gpu > gpurun_out/r04/patch_tests_2.txt 2>&1
echo "rc=$?" >> gpurun_out/r04/patch_tests_2.txt
tail -5 gpurun_out/r04/patch_tests_2.txt
