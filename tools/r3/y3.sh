#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r3y; mkdir -p $OUT
timeout -k 10 1500 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_capture.py -x -q > $OUT/pytest.txt 2>&1 || { tail -30 $OUT/pytest.txt; exit 1; }
tail -2 $OUT/pytest.txt
for dt in bf16s f32; do for e in 0 1; do
  E="DVSOF_X=1"; [ $e = 1 ] && E="DVSOF_NO_STEP_BEGIN=1"
  env $E python3 bench.py --dtype $dt --steps 40 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('bench $dt no_begin=$e', d['ms_per_step'], d['value'])" || exit 1
done; done
