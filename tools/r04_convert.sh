set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_convert_device.py -x -q -m gpu -s > gpurun_out/r04/pytest_convert_device.txt 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r04/pytest_convert_device.txt
python tools/sigma_study.py > gpurun_out/r04/sigma_study.txt 2>&1; echo "sigma rc=$?"
grep -v amdgpu.ids gpurun_out/r04/sigma_study.txt | cut -c1-400
