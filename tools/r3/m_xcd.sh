#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r3m; mkdir -p $OUT
DVSOF_GCONV_XCD=1 DVSOF_WGRAD_XCD=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_conv.py -x -q -k "test_conv_fwd_dgrad_wgrad or twins or folded" > $OUT/pytest.txt 2>&1 || { tail -20 $OUT/pytest.txt; exit 1; }
tail -1 $OUT/pytest.txt
for dt in bf16s f32 bf16; do for x in 0 1; do
  DVSOF_WGRAD_XCD=$x DVSOF_WGRAD_STREAM=0 python3 tools/conv_bench.py --dtype $dt > $OUT/conv_${dt}_w$x.txt 2>/dev/null || exit 1
  echo "== $dt wgrad xcd=$x: $(tail -1 $OUT/conv_${dt}_w$x.txt)"; grep "^wgrad" $OUT/conv_${dt}_w$x.txt | awk '{printf "%s ", $8} END {print ""}'
  DVSOF_WGRAD_XCD=$x python3 bench.py --dtype $dt --steps 40 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('bench $dt wgrad xcd=$x', d['ms_per_step'], d['value'])"
done; done
