#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r3wp; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_conv.py -x -q -k "fwd_dgrad_wgrad or twins or folded" > $OUT/pytest.txt 2>&1; rc=$?
tail -5 $OUT/pytest.txt
[ $rc -ne 0 ] && exit $rc
for dt in f32 bf16s; do
  DVSOF_WGRAD_STREAM=0 timeout -k 10 200 python3 tools/conv_bench.py --dtype $dt > $OUT/conv_$dt.txt 2>/dev/null || exit 1
  echo "== $dt: $(tail -1 $OUT/conv_$dt.txt)"; grep "^wgrad" $OUT/conv_$dt.txt | awk '{printf "%s ", $8} END {print ""}'
  for i in 1 2; do timeout -k 10 200 python3 bench.py --dtype $dt --steps 40 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('bench $dt', d['ms_per_step'], d['value'])" || exit 1; done
done
