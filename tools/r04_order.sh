cd $GRAFT_REPO_ROOT
q='import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(sys.argv[1], d["roofline"]["kernel_ms"], d["ms_per_step"], d["roofline"].get("traffic"))'
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-traffic --no-vendor-baseline --no-cpu-baseline --other-configs "" 2>/dev/null | python -c "$q" no-traffic
python bench.py --steps 20 --warmup 5 --no-vendor-baseline --no-cpu-baseline --other-configs "" 2>/dev/null | python -c "$q" traffic-children-first
done
python bench.py --steps 20 --warmup 5 --no-traffic --no-vendor-baseline --no-cpu-baseline --other-configs "" 2>/dev/null | python -c "$q" no-traffic
