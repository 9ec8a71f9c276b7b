set -x
timeout -k 10 600 python -m pytest tests/test_dist_native_gpu.py -x -q > gpurun_out/r02_g5_dist.log 2>&1; tail -5 gpurun_out/r02_g5_dist.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "block_plan or spmmv" > gpurun_out/r02_g5_block_tests.log 2>&1; tail -5 gpurun_out/r02_g5_block_tests.log
for tune in "spmmv_variant=6" "spmmv_variant=6,spmmv_unroll=8" "spmmv_variant=6,spmmv_swizzle=1" "spmmv_variant=6,ablate=1" "spmmv_variant=6,ablate=2" "spmmv_variant=6,xcd_remap=32" "spmmv_variant=6,xcd_remap=0"; do
  echo "== cfg3 $tune" >> gpurun_out/r02_g5_cfg3.log
  timeout -k 10 300 python tools/bench_configs.py --configs 3 --no-check --tune $tune >> gpurun_out/r02_g5_cfg3.log 2>&1
done
grep -E "^==|kernel_ms" gpurun_out/r02_g5_cfg3.log | cut -c1-420
