#!/bin/bash
OUT=gpurun_out/r4e; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_conv.py -x -q -k "(test_conv_fwd_dgrad_wgrad and f32) or nine_product or epilogue" > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.txt
export DVSOF_WGRAD_STREAM=0
timeout -k 10 300 python tools/conv_bench.py > $OUT/cb_new.txt 2>&1; echo "cb rc=$?"; grep "^dgrad" $OUT/cb_new.txt
DVSOF_NO_DGRAD_MIN=1 timeout -k 10 300 python tools/conv_bench.py > $OUT/cb_old.txt 2>&1; grep "^dgrad" $OUT/cb_old.txt | tail -5
unset DVSOF_WGRAD_STREAM
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'], d['ms_per_step'])"
