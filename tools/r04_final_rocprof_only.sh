OUT=$GRAFT_REPO_ROOT/gpurun_out/r04/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-traffic --no-vendor-baseline --other-configs 3,4b > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err; echo "rocprof rc=$?"
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/bench_kernel_stats.csv; grep "scs_spmv_tlc" $OUT/bench_kernel_stats.csv | cut -c1-60,330-420
rm -rf $OUT/prof
