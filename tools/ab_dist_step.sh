#!/bin/bash
# A/B of the loopback step (block 3 of 8 of the 304^3 stencil) through the CLI: graph x ba_synch x overlap x fused_step x pad_split
cd /tmp
export USPMV_LOOPBACK=8 USPMV_LOOPBACK_RANK=3 USPMV_ID_DIR=/tmp USPMV_JOB_ID=ab
EXE=$GRAFT_REPO_ROOT/ultimate-spmv_amd/uspmv
for rep in 1 2; do
for cfg in "1 1 1 1 1" "0 1 1 1 1" "1 0 1 1 1" "0 0 1 1 1" "1 0 1 0 1" "0 0 1 0 1" "0 0 1 0 0" "1 0 0 0 0" "0 0 0 0 0"; do
  set -- $cfg
  unset USPMV_NO_OVERLAP
  [ "$3" = "0" ] && export USPMV_NO_OVERLAP=1
  USPMV_FUSED_STEP=$4 USPMV_PAD_SPLIT=$5 $EXE gen:304x304x304 scs -c 32 -s 512 -seg_rows -comm_halos 1 -bench_steps 1500 -bench_warmup 50 -graph $1 -ba_synch $2 -check_y 1 2>&1 | grep -o "[0-9.]* ms per SpMV.*" | sed "s/^/graph=$1 ba_synch=$2 overlap=$3 fused=$4 pad_split=$5 : /" | sed "s/); rank.*ba_synch [01]//" | cut -c1-150
done; done
