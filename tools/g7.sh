set -x
for tune in "spmmv_variant=6,ablate=4" "spmmv_variant=6,ablate=5" "spmmv_variant=6" "spmmv_variant=6,spmmv_lds_kb=40" "spmmv_variant=6,spmmv_lds_kb=36"; do
  echo "== cfg3 $tune" >> gpurun_out/r02_g7_cfg3.log
  USPMV_VERBOSE=1 timeout -k 10 300 python tools/bench_configs.py --configs 3 --no-check --tune $tune >> gpurun_out/r02_g7_cfg3.log 2>&1
done
grep -E "^==|kernel_ms|block plan" gpurun_out/r02_g7_cfg3.log | cut -c1-330
timeout -k 10 900 python -m pytest tests/test_cli_solve_gpu.py -x -q > gpurun_out/r02_g7_solve.log 2>&1; tail -5 gpurun_out/r02_g7_solve.log
