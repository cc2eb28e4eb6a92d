#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for cfg in "512 64" "512 96" "512 128" "768 128" "1024 128"; do
  set -- $cfg
  for i in 1 2; do
  DVSOF_WGRAD_PATCH_WGS=$1 DVSOF_WGRAD_PATCH_MAXS=$2 python3 bench.py --dtype bf16s --steps 40 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('bf16s wgs=$1 maxs=$2', d['ms_per_step'], d['value'])" || exit 1
  done
done
for cfg in "512 64" "512 128" "256 64"; do
  set -- $cfg
  DVSOF_WGRAD_PATCH_F32_WGS=$1 DVSOF_WGRAD_PATCH_MAXS=$2 python3 bench.py --steps 40 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('f32 wgs=$1 maxs=$2', d['ms_per_step'], d['value'])" || exit 1
done
