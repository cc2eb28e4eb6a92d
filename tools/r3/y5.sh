#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r3y; mkdir -p $OUT
timeout -k 10 1500 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_capture.py -x -q > $OUT/pytest.txt 2>&1 || { tail -30 $OUT/pytest.txt; exit 1; }
tail -2 $OUT/pytest.txt
for dt in bf16s f32; do for i in 1 2; do
  python3 bench.py --dtype $dt --steps 40 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('bench $dt', d['ms_per_step'], d['value'], d['config']['launch'][-40:])" || exit 1
done; done
