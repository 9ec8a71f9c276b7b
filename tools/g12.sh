set -x
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "device_plan" > gpurun_out/r02_g12_t.log 2>&1; tail -3 gpurun_out/r02_g12_t.log
for tune in "sweep_loop=0" "sweep_loop=2" "sweep_loop=2,sweep_unroll=4" ; do
  echo "== cfg4b $tune" >> gpurun_out/r02_g12_cfg4b.log
  timeout -k 10 400 python tools/bench_configs.py --configs 4b --tune $tune >> gpurun_out/r02_g12_cfg4b.log 2>&1
done
grep -E "^==|kernel_ms" gpurun_out/r02_g12_cfg4b.log | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('=='): print(l.strip()); continue
    d=json.loads(l); print('   ap', d['kernel_ms'], d['frac_of_8TBs'], '| dp plan', d['plain_dp_tlc_ms'], d['plain_dp_plan_frac_of_8TBs'], 'ok', d['bitexact_vs_oracle'], d['plain_dp_plan_bitexact_vs_gather'])
"
find / -xdev \( -iname "nlpkkt*" -o -iname "Queen_4147*" -o -iname "HV15R*" \) 2>/dev/null | head -5; echo "suitesparse probe done"; find / -xdev -name "*.mtx" -size +10M 2>/dev/null | head -5
