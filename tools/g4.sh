set -x
export USPMV_VERBOSE=1
timeout -k 10 300 python -m pytest tests/test_dist_native_gpu.py -x -q -k "cli" > gpurun_out/r02_g4_dist_cli.log 2>&1; tail -30 gpurun_out/r02_g4_dist_cli.log
timeout -k 10 300 python -m pytest "tests/test_dist_native_gpu.py::test_native_step_loopback_bitexact[2-shape0-32-512]" -x -q -s > gpurun_out/r02_g4_dist_py.log 2>&1; grep -E "uspmv|passed|failed|Segmentation|Error" gpurun_out/r02_g4_dist_py.log | head -30
unset USPMV_VERBOSE
for tune in "spmmv_variant=6,ablate=1" "spmmv_variant=6,ablate=2" "spmmv_variant=6,ablate=3" "spmmv_variant=6,spmmv_unroll=8,ablate=1"; do
  echo "== cfg3 $tune" >> gpurun_out/r02_g4_cfg3.log
  timeout -k 10 300 python tools/bench_configs.py --configs 3 --no-check --tune $tune >> gpurun_out/r02_g4_cfg3.log 2>&1
done
grep -E "^==|kernel_ms" gpurun_out/r02_g4_cfg3.log | cut -c1-420
bash tools/pmc_sq.sh gpurun_out/r02_pmc_cfg3_quad 3 "spmmv_variant=6"
grep -A24 "scs_spmmv_quad<double, 8, true, false" gpurun_out/r02_pmc_cfg3_quad/summary.txt
