#!/bin/bash
# PMC passes over tools/bench_configs.py --configs 3 (SpMMV b=8 dp) for one tuning; one rocprofv3 run per group.
# usage: tools/pmc_spmmv.sh <outdir> '<tune>'
OUT=$1; TUNE=$2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in \
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE GRBM_TA_BUSY" \
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_ACTIVE_INST_VALU" \
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" ; do
  i=$((i+1))
  if [ -n "$TUNE" ]; then
    timeout -k 5 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$ROOT/$OUT/g$i" -- python3 "$ROOT/tools/bench_configs.py" --configs 3 --reps 5 --no-check --tune "$TUNE" > "$ROOT/$OUT/g$i.log" 2>&1 || echo "group $i failed"
  else
    timeout -k 5 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$ROOT/$OUT/g$i" -- python3 "$ROOT/tools/bench_configs.py" --configs 3 --reps 5 --no-check > "$ROOT/$OUT/g$i.log" 2>&1 || echo "group $i failed"
  fi
  echo "group $i done"
done
