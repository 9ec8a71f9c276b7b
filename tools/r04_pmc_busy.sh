#!/bin/bash
# which unit of the vector-memory path is busy under the phased SpMMV kernel (config 3)?  busy / stall counters of TA, TD, TCP and the address
# translation, at most two counters of a block per pass (more: "exceeds the capabilities of the hardware"), one rocprofv3 run per group
ROOT=$GRAFT_REPO_ROOT
OUT=gpurun_out/r04b/pmc_busy_${PMC_TAG:-cfg3}
rm -rf $ROOT/$OUT; mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in \
 "TA_BUSY_avr TA_BUSY_max" \
 "TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
 "TD_TD_BUSY_sum TD_TC_STALL_sum" \
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" \
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" ; do
  i=$((i+1))
  timeout -k 5 70 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$ROOT/$OUT/g$i" -- python3 "$ROOT/tools/${PMC_WORKLOAD:-pmc_cfg3_min.py}" > "$ROOT/$OUT/g$i.log" 2>&1 || echo "group $i failed: $(grep -m1 -i 'error code\|exceeds' $ROOT/$OUT/g$i.log)"
  echo "group $i done"
done
cd $ROOT
python3 tools/pmc_summary.py $OUT > $OUT.txt 2>&1
find $OUT -name "*.csv" -size +2M -delete
grep -A 16 "${PMC_KERNEL:-quadph}" $OUT.txt | head -60
