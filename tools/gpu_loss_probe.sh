#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_loss.py tests/test_gpu_conv.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -2
for f in 0 100000; do
DVSOF_LOSS_FOLD_MAX=$f python3 tools/hbm_bench.py 2>/dev/null | grep loss
done
