#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r3fo; mkdir -p $OUT
timeout -k 10 1100 python3 -m pytest tests/test_gpu_capture.py tests/test_gpu_model.py tests/test_gpu_configs.py -x -q > $OUT/pytest.txt 2>&1; rc=$?
tail -6 $OUT/pytest.txt
[ $rc -ne 0 ] && exit $rc
for e in DVSOF_X=1 DVSOF_FORCE_DIST=1; do for i in 1 2; do
  env $e timeout -k 10 200 python3 bench.py --steps 40 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('bench f32 $e', d['ms_per_step'], d['value'], d['config']['launch'][-110:])" || exit 1
done; done
