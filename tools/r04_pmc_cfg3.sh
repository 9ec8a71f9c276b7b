#!/bin/bash
# round 4: counter passes over config 3 with the default block plan (flat row patches + DP cuts) and with the device builder's plan (ties undone, greedy cuts)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
bash tools/pmc_spmmv.sh gpurun_out/r04/pmc_cfg3_patches "" > gpurun_out/r04/pmc_cfg3_patches.log 2>&1
python3 tools/pmc_summary.py gpurun_out/r04/pmc_cfg3_patches > gpurun_out/r04/pmc_cfg3_patches.txt 2>&1
bash tools/pmc_spmmv.sh gpurun_out/r04/pmc_cfg3_ties "spmmv_reorder=1,spmmv_phase_dp=0" > gpurun_out/r04/pmc_cfg3_ties.log 2>&1
python3 tools/pmc_summary.py gpurun_out/r04/pmc_cfg3_ties > gpurun_out/r04/pmc_cfg3_ties.txt 2>&1
find gpurun_out/r04/pmc_cfg3_patches gpurun_out/r04/pmc_cfg3_ties -name "*.csv" -size +2M -delete
tail -30 gpurun_out/r04/pmc_cfg3_patches.txt; tail -30 gpurun_out/r04/pmc_cfg3_ties.txt
