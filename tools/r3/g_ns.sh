#!/bin/bash
# bf16 twins: ring depth of the one-workgroup-per-CU (K64) form; loss reduce check
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r3g; mkdir -p $OUT
python3 -m pytest tests/test_gpu_loss.py tests/test_oracle_loss.py -x -q > $OUT/pytest_loss.txt 2>&1 || { tail -20 $OUT/pytest_loss.txt; exit 1; }
tail -2 $OUT/pytest_loss.txt
for ns in 2 3 4; do
  DVSOF_GCONV_K64_NS=$ns DVSOF_WGRAD_STREAM=0 python3 tools/conv_bench.py --dtype bf16s > $OUT/conv_ns$ns.txt 2>/dev/null || exit 1
  echo "== NS=$ns"; awk '$2==2048 && $3==512 {print}' $OUT/conv_ns$ns.txt; tail -1 $OUT/conv_ns$ns.txt
  DVSOF_GCONV_K64_NS=$ns python3 bench.py --dtype bf16s --steps 40 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('bench bf16s NS=$ns', d['ms_per_step'], d['value'])"
done
python3 tools/hbm_bench.py 2>/dev/null | tee $OUT/hbm.txt
