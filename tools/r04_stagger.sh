cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for k in 1 2 3; do python tools/stagger_probe.py 2 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04/stagger_probe_cfg2.txt; done
python tools/stagger_probe.py 3 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04/stagger_probe_cfg3.txt
python -m pytest tests/test_gpu_convert_device.py -x -q -m gpu -s > gpurun_out/r04/pytest_convert_device.txt 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r04/pytest_convert_device.txt
