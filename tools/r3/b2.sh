#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for i in 1 2; do for e in DVSOF_X=1 DVSOF_NO_FWD_PATCH_F32=1; do
  env $e timeout -k 10 200 python3 bench.py --steps 40 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('bench f32 $e', d['ms_per_step'], d['value'])" || exit 1
done; done
