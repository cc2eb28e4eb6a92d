#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r3ff; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_conv.py -x -q -k "(fwd_dgrad_wgrad and f32) or finest or folded" > $OUT/pytest.txt 2>&1; rc=$?
tail -15 $OUT/pytest.txt
[ $rc -ne 0 ] && exit $rc
for x in 1 0; do
  E="DVSOF_X=1"; [ $x = 0 ] && E="DVSOF_NO_FWD_PATCH_F32=1"
  env $E DVSOF_WGRAD_STREAM=0 timeout -k 10 200 python3 tools/conv_bench.py > $OUT/conv_p$x.txt 2>/dev/null || exit 1
  echo "== patch=$x: $(tail -1 $OUT/conv_p$x.txt)"; grep "^fwd" $OUT/conv_p$x.txt | awk '{printf "%s ", $8} END {print ""}'
  env $E timeout -k 10 200 python3 bench.py --steps 40 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('bench f32 fwd_patch_f32=$x', d['ms_per_step'], d['value'])" || exit 1
done
