set -x
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "block_plan or spmmv" > gpurun_out/r02_g17_block_tests.log 2>&1; tail -5 gpurun_out/r02_g17_block_tests.log
for tune in "spmmv_xcol=1" "spmmv_xcol=0" "spmmv_xcol=1,spmmv_phase_rows=512"; do
  echo "== cfg3 $tune" >> gpurun_out/r02_g17_cfg3.log
  timeout -k 10 300 python tools/bench_configs.py --configs 3 --tune $tune >> gpurun_out/r02_g17_cfg3.log 2>&1
done
grep -E "^==|kernel_ms" gpurun_out/r02_g17_cfg3.log | cut -c1-420
