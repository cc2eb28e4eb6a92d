#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for m in 64 96 128 64 128; do python3 -c "print('maxS $m', end=' ')"; DVSOF_WGRAD_PATCH_MAXS=$m python3 bench.py --dtype bf16s --steps 40 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('bench bf16s', d['ms_per_step'], d['value'])"; done
DVSOF_NO_WGRAD_PATCH=1 python3 bench.py --dtype bf16s --steps 40 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('no patch: bench bf16s', d['ms_per_step'], d['value'])"
