cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_spmmv_sweep.py -x -q -m gpu 2>&1 | tail -15
USPMV_VERBOSE=1 timeout -k 10 600 python tools/spmmv_sweep_probe.py 2>&1 | grep -v "amdgpu.ids\|tlc plan\|phased\b.*plan:" | tee gpurun_out/r04/spmmv_sweep_probe.txt | cut -c1-330
