#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r3h; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_conv.py -x -q -k "test_conv_fwd_dgrad_wgrad" > $OUT/pytest_conv.txt 2>&1 || { tail -30 $OUT/pytest_conv.txt; exit 1; }
tail -2 $OUT/pytest_conv.txt
for dt in f32 bf16s; do
  DVSOF_WGRAD_STREAM=0 python3 tools/conv_bench.py --dtype $dt > $OUT/conv_$dt.txt 2>/dev/null || exit 1
  echo "== $dt"; awk '$4==45 {print}' $OUT/conv_$dt.txt; tail -1 $OUT/conv_$dt.txt
  python3 bench.py --dtype $dt --steps 40 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('bench $dt', d['ms_per_step'], d['value'])"
done
