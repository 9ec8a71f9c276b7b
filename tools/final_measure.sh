# Final evidence run of a round: full GPU test suite, the headline bench with and without rocprofv3 kernel statistics, every
# single-GPU BASELINE configuration with the reference CPU kernel beside it, counter passes, the distributed step cost.
# usage: bash tools/final_measure.sh <out dir under gpurun_out/>
OUT=gpurun_out/$1; mkdir -p $OUT
set -x
timeout -k 10 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
timeout -k 10 600 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; cut -c1-300 $OUT/bench_n1.json
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-traffic --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.err; find /tmp/prof_bench -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/$OUT/bench_kernel_stats.csv \; )
head -5 $OUT/bench_kernel_stats.csv
timeout -k 10 900 python tools/bench_configs.py --configs 2,3,4,4b --cpu-seconds 5 > $OUT/configs.txt 2> $OUT/configs.err; cut -c1-200 $OUT/configs.txt
timeout -k 10 600 python tools/dist_step_cost.py > $OUT/dist_step_cost.json 2> $OUT/dist_step_cost.err; cat $OUT/dist_step_cost.json
bash tools/pmc_sq.sh $OUT/pmc_cfg3 3 ""
bash tools/pmc_sq.sh $OUT/pmc_cfg4b 4b ""
# two ranks on the one GPU (gloo; the C++ RCCL object refuses duplicate devices, so this rehearses bench.py's N > 1 control flow on the Python step)
( export USPMV_BENCH_ONE_DEVICE=1; timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --grid 96 --grid2 64 > $OUT/bench_2rank_1gpu_rehearsal.json 2> $OUT/bench_2rank_1gpu_rehearsal.err ); cut -c1-400 $OUT/bench_2rank_1gpu_rehearsal.json
# the same code path with ONE rank and the real thing (torch's nccl = RCCL process group, the C++ RCCL step object, the guarded first step)
( export USPMV_BENCH_WORLD1=1; timeout -k 10 600 python bench.py --gpus 2 --steps 20 --warmup 5 --grid 96 --grid2 64 > $OUT/bench_dist_path_1rank.json 2> $OUT/bench_dist_path_1rank.err ); cut -c1-400 $OUT/bench_dist_path_1rank.json
