cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
(time python -m pytest tests/ -x -q -m gpu) > gpurun_out/r04/pytest_gpu_full.txt 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/r04/pytest_gpu_full.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_n1_b.json 2> gpurun_out/r04/bench_n1_b.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r04/bench_n1_b.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','bitexact_vs_reference_cpu')}, d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['stream_same_run_GBs'])
s=d['gpu_state']['during_the_kernel_timing']; print({k:s.get(k) for k in ('sclk','hwmon_power1_input','hwmon_temp3_input','hwmon_temp2_input')})
for o in d['other_configs']:
    print(o['config'], o['ms_per_step'], o['kernel_ms'], o['roofline']['frac'], o['bitexact_vs_reference_cpu'], o['setup_s'])
P
