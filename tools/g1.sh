set -x
python -m pytest tests/test_gpu_sweep.py -x -q > gpurun_out/r02_g1_sweep_tests.log 2>&1; tail -5 gpurun_out/r02_g1_sweep_tests.log
python -m pytest tests/test_gpu_parity.py -x -q -k "block_plan or spmmv" > gpurun_out/r02_g1_block_tests.log 2>&1; tail -5 gpurun_out/r02_g1_block_tests.log
export USPMV_VERBOSE=1
for tune in "spmmv_variant=3" "spmmv_variant=4,spmmv_reorder=0" "spmmv_variant=4" "spmmv_variant=4,spmmv_tile_rows=64" ; do
  echo "== cfg3 $tune" >> gpurun_out/r02_g1_cfg3.log
  timeout -k 10 300 python tools/bench_configs.py --configs 3 --no-check --tune $tune >> gpurun_out/r02_g1_cfg3.log 2>&1
done
tail -20 gpurun_out/r02_g1_cfg3.log
for tune in "sweep=0" "sweep=1" "sweep_nbuf=1" "sweep_unroll=4" "sweep_tile_rows=512" "sweep_tile_rows=512,sweep_wlog=12" "sweep_remap=0" "sweep_remap=32"; do
  echo "== cfg4b $tune" >> gpurun_out/r02_g1_cfg4b.log
  timeout -k 10 400 python tools/bench_configs.py --configs 4b --tune $tune >> gpurun_out/r02_g1_cfg4b.log 2>&1
done
tail -30 gpurun_out/r02_g1_cfg4b.log
