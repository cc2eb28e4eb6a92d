#!/bin/bash
OUT=gpurun_out/r4i; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_loss.py -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.txt
timeout -k 10 300 python tools/hbm_bench.py 2>&1 | grep loss
DVSOF_PYR_BIG=0 timeout -k 10 300 python tools/hbm_bench.py 2>&1 | grep loss
