#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for dt in f32 bf16x3 bf16 bf16s; do
  for i in 1 2; do python3 bench.py --dtype $dt --steps 40 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('bench $dt', d['ms_per_step'], d['value'])"; done
done
DVSOF_GCONV_XCD=0 DVSOF_WGRAD_XCD=0 python3 bench.py --steps 40 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('bench f32 no-xcd', d['ms_per_step'], d['value'])"
