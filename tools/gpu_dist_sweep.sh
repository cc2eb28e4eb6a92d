#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
export MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
run() { name=$1; shift; env "$@" MASTER_PORT=$((29600 + RANDOM % 300)) timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-roofline > $O/ds_$name.json 2>$O/ds_$name.err || { echo "$name FAILED"; tail -3 $O/ds_$name.err; return; }
  python3 -c "import json; d=json.load(open('$O/ds_$name.json')); print('$name', d['value'], d['ms_per_step'])"; }
run nodist X=1
run dist DVSOF_FORCE_DIST=1
run dist_direct DVSOF_FORCE_DIST=1 DVSOF_DIRECT_RCCL=1
run dist_ch4 DVSOF_FORCE_DIST=1 NCCL_MAX_NCHANNELS=4
run dist_fusedopt DVSOF_FORCE_DIST=1 DVSOF_BENCH_FUSED=1
run nodist X=1
