#!/bin/bash
export DVSOF_WGRAD_STREAM=0
timeout -k 10 300 python tools/conv_bench.py 2>&1 | awk '$6==1 {printf "%s %s | ", $1, $8} END{print ""}'
unset DVSOF_WGRAD_STREAM
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py -x -q -k "(test_conv_fwd_dgrad_wgrad and f32) or nine_product" 2>&1 | tail -2
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'], d['ms_per_step'])"
