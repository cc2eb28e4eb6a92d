#!/bin/bash
# lane plans of the step executor: each plan fixed, then the timed choice
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r3y; mkdir -p $OUT
for dt in bf16s f32; do for pl in paths list chain auto; do
  E="DVSOF_EXEC_PLAN=$pl"
  env $E python3 bench.py --dtype $dt --steps 40 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('bench $dt plan=$pl', d['ms_per_step'], d['value'], d['config']['launch'][-60:])" || exit 1
done; done
