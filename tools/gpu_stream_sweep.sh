#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
run() { name=$1; shift; env "$@" timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-roofline > $O/ss_$name.json 2>$O/ss_$name.err || { echo "$name FAILED"; tail -3 $O/ss_$name.err; return; }
  python3 -c "import json; d=json.load(open('$O/ss_$name.json')); print('$name', d['value'], d['ms_per_step'])"; }
run s1 DVSOF_WGRAD_STREAMS=1
run s2 DVSOF_WGRAD_STREAMS=2
run s3 DVSOF_WGRAD_STREAMS=3
run s2_side0 DVSOF_WGRAD_STREAMS=2 DVSOF_ENC_SIDE_FROM=0
run s2_side4 DVSOF_WGRAD_STREAMS=2 DVSOF_ENC_SIDE_FROM=4
run s1_fused DVSOF_WGRAD_STREAMS=1 FUSED=1
run s1 DVSOF_WGRAD_STREAMS=1
