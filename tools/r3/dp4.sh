#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 1100 python3 -m pytest tests/test_gpu_capture.py tests/test_gpu_configs.py -x -q -k "parallel or rccl or accumulation or optimizer_inside" 2>&1 | tail -2
bash tools/r3/dp1.sh
for e in "DVSOF_FORCE_DIST=1"; do
  env $e timeout -k 10 200 python3 bench.py --fused-optimizer buckets --steps 60 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('bench f32 fused buckets [$e]', d['ms_per_step'], d['value'])" || exit 1
done
