set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_convert_device.py -x -q -m gpu -s > gpurun_out/r04/pytest_convert_device.txt 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/r04/pytest_convert_device.txt
python tools/placement_probe.py > gpurun_out/r04/placement_probe.txt 2>&1; echo "probe rc=$?"
grep -v amdgpu.ids gpurun_out/r04/placement_probe.txt | cut -c1-600
