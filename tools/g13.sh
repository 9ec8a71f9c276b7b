set -x
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "block_plan or spmmv" > gpurun_out/r02_g13_block_tests.log 2>&1; tail -5 gpurun_out/r02_g13_block_tests.log
for tune in "spmmv_variant=8" "spmmv_variant=8,spmmv_phase_rows=512" "spmmv_variant=8,xcd_remap=64" "spmmv_variant=6"; do
  echo "== cfg3 $tune" >> gpurun_out/r02_g13_cfg3.log
  USPMV_VERBOSE=1 timeout -k 10 300 python tools/bench_configs.py --configs 3 --tune $tune >> gpurun_out/r02_g13_cfg3.log 2>&1
done
grep -E "^==|kernel_ms|phased" gpurun_out/r02_g13_cfg3.log | cut -c1-420
