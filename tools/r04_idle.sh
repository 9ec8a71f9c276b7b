set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
ls /sys/class/drm/ > gpurun_out/r04/sysfs_probe.txt 2>&1
for f in /sys/class/drm/card*/device/pp_dpm_sclk /sys/class/drm/card*/device/pp_dpm_mclk /sys/class/drm/card*/device/pp_dpm_fclk /sys/class/drm/card*/device/pp_dpm_socclk /sys/class/drm/card*/device/hwmon/hwmon*/*; do echo "== $f"; cat $f 2>&1 | head -12; done >> gpurun_out/r04/sysfs_probe.txt 2>&1
(rocm-smi --showclocks --showpower --showtemp --showmaxpower --json 2>&1 | head -c 4000) >> gpurun_out/r04/sysfs_probe.txt
python tools/idle_ramp_probe.py 3 > gpurun_out/r04/idle_ramp_cfg3.txt 2>&1 && cat gpurun_out/r04/idle_ramp_cfg3.txt | cut -c1-600 && \
python tools/idle_ramp_probe.py 2 > gpurun_out/r04/idle_ramp_cfg2.txt 2>&1 && cat gpurun_out/r04/idle_ramp_cfg2.txt | cut -c1-600
