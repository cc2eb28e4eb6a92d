#!/bin/bash
# longer runs of the captured step in every launch configuration (rare faults / hangs)
cd "${GRAFT_REPO_ROOT:-.}"
for cfg in "f32 DVSOF_X=1" "bf16s DVSOF_X=1" "f32 DVSOF_FORCE_DIST=1" "bf16s DVSOF_FORCE_DIST=1" "bf16 DVSOF_X=1" "bf16x3 DVSOF_X=1"; do
  set -- $cfg
  env $2 timeout -k 10 300 python3 bench.py --dtype $1 --steps 1500 --warmup 20 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('soak $1 $2', d['ms_per_step'], d['value'], d['config']['final_loss'])" || { echo "FAILED $cfg"; exit 1; }
done
