#!/bin/bash
# PMC passes (one rocprofv3 run per counter group, --kernel-trace only) over tools/sweep.py.
# usage: tools/pmc.sh <outdir> '<variants>' [grid]
OUT=$1; VAR=$2; GRID=${3:-253}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in \
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE GRBM_TA_BUSY" \
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_ACTIVE_INST_VALU" \
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
 "TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum" \
 "TCC_BUSY_sum TCC_CYCLE_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
 "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_STREAMING_REQ_sum TCC_NC_REQ_sum" ; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$ROOT/$OUT/g$i" -- python3 "$ROOT/tools/sweep.py" --grid $GRID --rounds 1 --reps 5 --variants "$VAR" > "$ROOT/$OUT/g$i.log" 2>&1 || echo "group $i failed"
  echo "group $i done"
done
