#!/bin/bash
# does the C-ABI collective's +53 % go away with more hardware queues? (stream -> HW queue collisions)
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r3b; mkdir -p $OUT
ARGS="bench.py --eager --steps 20 --warmup 3 --no-roofline --no-other-modes --no-cpu-baseline"
for q in 4 8 12 16 24; do
  for mode in dist direct; do
    extra=""; [ $mode = direct ] && extra="DVSOF_DIRECT_RCCL=1"
    env GPU_MAX_HW_QUEUES=$q DVSOF_FORCE_DIST=1 $extra python3 $ARGS > $OUT/${mode}_q$q.json 2> $OUT/${mode}_q$q.err || { tail -5 $OUT/${mode}_q$q.err; exit 1; }
    echo "$mode q=$q $(python3 -c "import json;print(json.load(open('$OUT/${mode}_q$q.json'))['ms_per_step'])")"
  done
done
env GPU_MAX_HW_QUEUES=8 python3 $ARGS | python3 -c "import json,sys;print('nodist q=8', json.loads(sys.stdin.read())['ms_per_step'])"
