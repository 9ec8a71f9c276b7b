# the headline kernel alone on its matrix under rocprofv3 --kernel-trace --stats (2k / 5one of the default line run the same kernel on other matrices)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof2 -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-traffic --no-vendor-baseline --other-configs 3,4b > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err; echo "rocprof rc=$?"
f=$(find $OUT/prof2 -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/bench_kernel_stats.csv; head -4 $OUT/bench_kernel_stats.csv | cut -c1-60,300-420
rm -rf $OUT/prof2
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_dist_native_gpu.py -x -q -m gpu -k "bench_launcher or seg_metis" 2>&1 | tail -4
