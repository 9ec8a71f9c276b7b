# which of C and sigma makes `scs -c 8 -s 64` faster than `scs -c 32 -s 512` on the nlpkkt200-class stencil?  (two rounds, same order: run-to-run spread)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04b
OUT=gpurun_out/r04b/c_sweep2.txt
: > $OUT
for round in 1 2; do
  for fmt in "scs -c 32 -s 512" "scs -c 8 -s 64" "scs -c 32 -s 64" "scs -c 8 -s 512" "scs -c 16 -s 64" "scs -c 4 -s 64" "scs -c 8 -s 8" "scs -c 32 -s 32" "scs -c 64 -s 64" "crs"; do
    r=$(cd /tmp && USPMV_VERBOSE=1 timeout -k 5 120 $GRAFT_REPO_ROOT/ultimate-spmv_amd/uspmv gen:253x253x253 $fmt -mode b -bench_time 0.3 -dp 2>&1 | grep -E "Total Gflops|tile-local|lines|rror" | tr '\n' ' ' | cut -c1-420)
    echo "$round | $fmt | $r" | tee -a $OUT
  done
done
