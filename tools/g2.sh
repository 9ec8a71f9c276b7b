set -x
for tune in "spmmv_variant=4,spmmv_tile_rows=64,spmmv_swizzle=1" "spmmv_variant=4,spmmv_swizzle=1" "spmmv_variant=5" "spmmv_variant=5,spmmv_prefetch=0" "spmmv_variant=5,block=128"; do
  echo "== cfg3 $tune" >> gpurun_out/r02_g2_cfg3.log
  timeout -k 10 300 python tools/bench_configs.py --configs 3 --no-check --tune $tune >> gpurun_out/r02_g2_cfg3.log 2>&1
done
grep -E "^==|kernel_ms" gpurun_out/r02_g2_cfg3.log | cut -c1-400
for tune in "sweep_nbuf=1" "sweep_nbuf=1,sweep_wlog=12" "sweep_nbuf=1,sweep_unroll=4" "sweep_nbuf=1,sweep_remap=32" "sweep_nbuf=1,sweep_remap=0,xcd_remap=0"; do
  echo "== cfg4b $tune" >> gpurun_out/r02_g2_cfg4b.log
  timeout -k 10 400 python tools/bench_configs.py --configs 4b --no-check --tune $tune >> gpurun_out/r02_g2_cfg4b.log 2>&1
done
grep -E "^==|kernel_ms" gpurun_out/r02_g2_cfg4b.log | cut -c1-100
bash tools/pmc_sq.sh gpurun_out/r02_pmc_cfg3_plan64 3 "spmmv_variant=4,spmmv_tile_rows=64"
cat gpurun_out/r02_pmc_cfg3_plan64/summary.txt
bash tools/pmc_sq.sh gpurun_out/r02_pmc_cfg4b_sweep 4b "sweep_nbuf=1"
cat gpurun_out/r02_pmc_cfg4b_sweep/summary.txt
