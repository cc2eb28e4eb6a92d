#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for dt in bf16s f32; do for e in "DVSOF_X=1" "DVSOF_FORCE_DIST=1" "DVSOF_FORCE_DIST=1 DVSOF_EXCHANGE_ON_WGRAD_STREAM=1"; do
  env $e timeout -k 10 200 python3 bench.py --dtype $dt --steps 60 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('bench $dt [$e]', d['ms_per_step'], d['value'])" || exit 1
done; done
