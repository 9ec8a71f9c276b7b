set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python tools/xcd_group_probe.py 3 > gpurun_out/r04/xcd_group_cfg3.txt 2>&1 && tail -20 gpurun_out/r04/xcd_group_cfg3.txt && \
python tools/xcd_group_probe.py 2 > gpurun_out/r04/xcd_group_cfg2.txt 2>&1 && tail -20 gpurun_out/r04/xcd_group_cfg2.txt && \
python tools/xcd_group_probe.py 5 > gpurun_out/r04/xcd_group_cfg5.txt 2>&1 && tail -14 gpurun_out/r04/xcd_group_cfg5.txt
