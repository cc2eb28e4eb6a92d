#!/bin/bash
# round 4, first GPU call: the exchange-ordering tests (loopback communicator), the 1-rank RCCL
# tests on the direct communicator, bench.py's self-launcher on a 1-GPU box, RCCL with 2 ranks on 1 GPU
OUT=gpurun_out/r4a; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_capture.py -x -q -k "non_identity or rerecording or exchange or optimizer_inside or accumulation" > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.txt
timeout -k 10 120 python bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/bench2.out 2> $OUT/bench2.err; echo "bench --gpus 2 rc=$?"; tail -3 $OUT/bench2.err
timeout -k 10 200 python tools/r4/rccl_share_gpu.py > $OUT/share.out 2>&1; echo "share rc=$?"; grep -v "^$" $OUT/share.out | tail -8
