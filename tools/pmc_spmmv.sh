#!/bin/bash
# PMC passes over tools/bench_configs.py --configs 3 (SpMMV b=8 dp) for one tuning; one rocprofv3 run per group.
# usage: tools/pmc_spmmv.sh <outdir> '<tune>' [extra bench_configs flags, e.g. --sp]
OUT=$1; TUNE=$2; EXTRA=$3
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in \
 "FETCH_SIZE WRITE_SIZE" \
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" ; do
  i=$((i+1))
  if [ -n "$TUNE" ]; then
    timeout -k 5 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$ROOT/$OUT/g$i" -- python3 "$ROOT/tools/bench_configs.py" --configs 3 --reps 5 --no-check $EXTRA --tune "$TUNE" > "$ROOT/$OUT/g$i.log" 2>&1 || echo "group $i failed"
  else
    timeout -k 5 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$ROOT/$OUT/g$i" -- python3 "$ROOT/tools/bench_configs.py" --configs 3 --reps 5 --no-check $EXTRA > "$ROOT/$OUT/g$i.log" 2>&1 || echo "group $i failed"
  fi
  echo "group $i done"
done
