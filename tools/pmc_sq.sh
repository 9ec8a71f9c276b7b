#!/bin/bash
# SQ / LDS / L1 / L2 counter passes over tools/bench_configs.py for one config and tuning; one rocprofv3 run per group
# (--pmc with --kernel-trace only).  usage: tools/pmc_sq.sh <outdir> <config> '<tune>' [extra bench_configs flags]
OUT=$1; CFG=$2; TUNE=$3; EXTRA=$4
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAVES" \
 "FETCH_SIZE" "WRITE_SIZE" \
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$ROOT/$OUT/g$i" -- python3 "$ROOT/tools/bench_configs.py" --configs $CFG --reps 5 --no-check $EXTRA ${TUNE:+--tune "$TUNE"} > "$ROOT/$OUT/g$i.log" 2>&1 || echo "group $i failed"
  echo "group $i done"
done
python3 "$ROOT/tools/pmc_summary.py" "$ROOT/$OUT" > "$ROOT/$OUT/summary.txt" 2>&1
rm -rf "$ROOT/$OUT"/g*/   # keep the summary and logs only (the raw csv trees are large)
