cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python tools/xcd_group_probe.py 3 256 128 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/spmmv_listahead_timing.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "spmmv or block or Queen or queen" > gpurun_out/r04/pytest_spmmv.txt 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r04/pytest_spmmv.txt
