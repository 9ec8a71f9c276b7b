set -x
timeout -k 10 900 python -m pytest tests/test_gpu_sweep.py -x -q > gpurun_out/r02_g14_tests.log 2>&1; tail -5 gpurun_out/r02_g14_tests.log
for tune in "sweep_tile_rows=1024" "sweep_tile_rows=2048" "sweep_tile_rows=4096" "sweep_tile_rows=2048,sweep_wlog=12" "sweep_tile_rows=2048,sweep_remap=4"; do
  echo "== cfg4b $tune" >> gpurun_out/r02_g14_cfg4b.log
  timeout -k 10 400 python tools/bench_configs.py --configs 4b --tune $tune >> gpurun_out/r02_g14_cfg4b.log 2>&1
done
grep -E "^==|kernel_ms" gpurun_out/r02_g14_cfg4b.log | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('=='): print(l.strip()); continue
    d=json.loads(l); print('   ap', d['kernel_ms'], d['frac_of_8TBs'], '| dp plan', d['plain_dp_tlc_ms'], d['plain_dp_plan_frac_of_8TBs'], 'ok', d['bitexact_vs_oracle'], d['plain_dp_plan_bitexact_vs_gather'])
"
