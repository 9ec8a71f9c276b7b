# the evidence run of the round: `python bench.py` as the driver runs it, then the same command under rocprofv3 --kernel-trace --stats
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04/final
mkdir -p $OUT
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-traffic --no-vendor-baseline --other-configs 3,4b > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err; echo "rocprof rc=$?"
find $OUT/prof -name "*kernel_stats.csv" | head -3
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/bench_kernel_stats.csv 2>/dev/null; head -12 $OUT/bench_kernel_stats.csv | cut -c1-200
rm -rf $OUT/prof
