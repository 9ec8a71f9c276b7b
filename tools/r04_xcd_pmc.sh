cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04/pmc_xcd_cfg3
mkdir -p $OUT
python -m pytest tests/test_gpu_convert_device.py -x -q -m gpu -s > gpurun_out/r04/pytest_convert_device.txt 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r04/pytest_convert_device.txt
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" ; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 $GRAFT_REPO_ROOT/tools/xcd_pmc_cfg3.py > $OUT/g$i.log 2>&1 || echo "group $i failed"
  grep xcd_remap $OUT/g$i.log | tr '\n' ' '; echo
done
python3 $GRAFT_REPO_ROOT/tools/placement_pmc_summary.py $OUT scs_spmmv_quadph > $OUT/summary.txt 2>&1
sed 's/fast/G256/; s/slow/G72 /; s/slow\/fast/G72\/G256/' $OUT/summary.txt
rm -rf $OUT/g*/
