#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for i in 1 2 3; do for fo in none buckets; do
  F=""; [ $fo != none ] && F="--fused-optimizer $fo"
  timeout -k 10 200 python3 bench.py $F --steps 40 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('bench f32 fused=$fo', d['ms_per_step'], d['value'], d['config']['launch'][-20:])" || exit 1
done; done
