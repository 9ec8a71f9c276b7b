cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04/pmc_place
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_available.txt 2>&1
grep -c "" $OUT/counters_available.txt
python3 $GRAFT_REPO_ROOT/tools/placement_pmc.py > $OUT/unprofiled.txt 2>&1; cat $OUT/unprofiled.txt | grep -v amdgpu
i=0
for grp in \
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
 "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_EA0_WRREQ_sum" \
 "TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum" \
 "TCC_BUSY_sum TCC_CYCLE_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum" \
 "TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
 "TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_EA0_RDREQ_IO_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum" \
 "TCC_WRITEBACK_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_STREAMING_REQ_sum TCC_NC_REQ_sum" \
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" ; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 $GRAFT_REPO_ROOT/tools/placement_pmc.py > $OUT/g$i.log 2>&1 || echo "group $i failed: $(tail -2 $OUT/g$i.log | cut -c1-200)"
  echo "group $i done"
done
python3 $GRAFT_REPO_ROOT/tools/placement_pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
rm -rf $OUT/g*/
