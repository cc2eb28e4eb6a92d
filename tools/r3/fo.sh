#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for dt in f32 bf16s; do for fo in none coarse buckets; do
  F=""; [ $fo != none ] && F="--fused-optimizer $fo"
  timeout -k 10 200 python3 bench.py --dtype $dt $F --steps 40 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('bench $dt fused=$fo', d['ms_per_step'], d['value'], d['config']['final_loss'], d['config']['launch'][:60])" || exit 1
done; done
