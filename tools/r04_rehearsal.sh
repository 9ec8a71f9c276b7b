set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python -m pytest tests/test_dist_native_gpu.py -x -q -m gpu -k "config5 or autotune or step_forms or real_ranks_host_exchange" > gpurun_out/r04/pytest_dist_new.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/pytest_dist_new.txt
tail -5 gpurun_out/r04/pytest_dist_new.txt
# self-launched N = 2 on ONE GPU: RCCL refuses duplicate devices -> the host-staged tier; single-GPU reference first
timeout -k 10 500 python bench.py --gpus 2 --steps 20 --warmup 5 --budget-s 420 > gpurun_out/r04/bench_n2_selflaunch_one_gpu.json 2> gpurun_out/r04/bench_n2_selflaunch_one_gpu.err; echo "self rc=$?"
tail -c 3000 gpurun_out/r04/bench_n2_selflaunch_one_gpu.json
tail -c 1500 gpurun_out/r04/bench_n2_selflaunch_one_gpu.err
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29701 bench.py --gpus 2 --steps 20 --warmup 5 --budget-s 300 --grid 160 --grid2 128 > gpurun_out/r04/bench_n2_torchrun_one_gpu.json 2> gpurun_out/r04/bench_n2_torchrun_one_gpu.err; echo "torchrun rc=$?"
tail -c 2000 gpurun_out/r04/bench_n2_torchrun_one_gpu.json
