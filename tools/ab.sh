#!/bin/bash
# A/B of two builds on one box, alternating: usage ab.sh <variant.so> <rounds> <bench_configs args...>
V=$1; R=$2; shift 2
for i in $(seq 1 $R); do
  for lib in "" "$V"; do
    tag=$([ -z "$lib" ] && echo new || echo old)
    USPMV_LIB=$lib timeout -k 10 300 python tools/bench_configs.py "$@" --no-check 2>/dev/null | grep "^{" | TAG=$tag python3 -c "
import sys, json, os
for l in sys.stdin:
    d=json.loads(l); print(os.environ['TAG'], d['config'], d['kernel_ms'], d.get('rowwise_ms'))"
  done
done
