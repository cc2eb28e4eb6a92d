#!/bin/bash
# executor under data parallelism: tests, then step time with / without the 1-rank RCCL group
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r3c; mkdir -p $OUT
timeout -k 10 1500 python3 -m pytest tests/test_gpu_capture.py -x -q > $OUT/pytest.txt 2>&1; rc=$?
tail -25 $OUT/pytest.txt
[ $rc -ne 0 ] && exit $rc
ARGS="bench.py --steps 40 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline"
b() {  # label env...
  l=$1; shift
  env "$@" python3 $ARGS > $OUT/$l.json 2> $OUT/$l.err || { tail -5 $OUT/$l.err; return 1; }
  python3 -c "import json;d=json.load(open('$OUT/$l.json'));print('$l', d['ms_per_step'], d['value'], d['config']['launch'][:150])"
}
b exec_nodist X=0 && b exec_dist_q8 DVSOF_FORCE_DIST=1 && b exec_dist_q4 DVSOF_FORCE_DIST=1 GPU_MAX_HW_QUEUES=4 \
 && b exec_dist_xlane DVSOF_FORCE_DIST=1 DVSOF_EXCHANGE_ON_WGRAD_STREAM=1 \
 && b eager_dist DVSOF_FORCE_DIST=1 DVSOF_EAGER=1 && b exec_dist_bf16s DVSOF_FORCE_DIST=1 DVSOF_DTYPE=bf16s && b exec_nodist_bf16s DVSOF_DTYPE=bf16s
