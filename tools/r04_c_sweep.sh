# chunk heights and precisions off the BASELINE configurations: the harness in bench mode on the nlpkkt200-class stencil (no perf cliffs?)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04b
OUT=gpurun_out/r04b/c_sweep.txt
: > $OUT
for prec in -dp -sp; do
  for fmt in "crs" "scs -c 1 -s 1" "scs -c 8 -s 64" "scs -c 16 -s 512" "scs -c 32 -s 1" "scs -c 32 -s 512" "scs -c 32 -s 4096" "scs -c 64 -s 512" "scs -c 128 -s 512" "scs -c 256 -s 512"; do
    r=$(cd /tmp && timeout -k 5 120 $GRAFT_REPO_ROOT/ultimate-spmv_amd/uspmv gen:253x253x253 $fmt -mode b -bench_time 0.3 $prec 2>&1 | grep -E "Total Gflops|Achieved|rror" | tr '\n' ' ')
    echo "$prec | $fmt | $r" | tee -a $OUT
  done
done
