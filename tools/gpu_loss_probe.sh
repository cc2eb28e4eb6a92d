#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_loss.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -1
DVSOF_LOSS_FOLD_MAX=0 timeout -k 10 300 python -m pytest tests/test_gpu_loss.py -m gpu -q -x 2>&1 | tail -1
DVSOF_LOSS_STRICT=1 timeout -k 10 300 python -m pytest tests/test_gpu_loss.py -m gpu -q -x 2>&1 | tail -1
python3 tools/hbm_bench.py 2>/dev/null | grep loss
