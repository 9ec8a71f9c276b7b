set -x
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "block_plan or spmmv" > gpurun_out/r02_g6_block_tests.log 2>&1; tail -5 gpurun_out/r02_g6_block_tests.log
for tune in "spmmv_variant=7" "spmmv_variant=7,spmmv_persist_w=3" "spmmv_variant=7,spmmv_persist_x=2" "spmmv_variant=7,spmmv_persist_x=4" "spmmv_variant=6"; do
  echo "== cfg3 $tune" >> gpurun_out/r02_g6_cfg3.log
  timeout -k 10 300 python tools/bench_configs.py --configs 3 --tune $tune >> gpurun_out/r02_g6_cfg3.log 2>&1
done
grep -E "^==|kernel_ms" gpurun_out/r02_g6_cfg3.log | cut -c1-420
timeout -k 10 900 python -m pytest tests/test_cpp_launchers.py tests/test_gpu_fullsize.py -x -q -s > gpurun_out/r02_g6_new_tests.log 2>&1; tail -25 gpurun_out/r02_g6_new_tests.log
