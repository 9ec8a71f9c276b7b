set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_n1.json 2> gpurun_out/r04/bench_n1.err; echo "bench rc=$?"
tail -c 600 gpurun_out/r04/bench_n1.err
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r04/bench_n1.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','bitexact_vs_reference_cpu')}, d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['stream_same_run_GBs'])
print(json.dumps(d['gpu_state'])[:1500])
for o in d['other_configs']:
    print(o['config'], o['ms_per_step'], o['kernel_ms'], o['roofline']['frac'], o['bitexact_vs_reference_cpu'], o['setup_s'])
P
