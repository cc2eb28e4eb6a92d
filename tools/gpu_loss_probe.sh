#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_loss.py -m gpu -q -x 2>&1 | tail -1
for d in 0 128 384 640; do
  echo -n "dbg=$d B=64: "; DVSOF_LOSS_DBG=$d python3 tools/loss_probe.py 64 256 256 2>/dev/null
done
python3 tools/hbm_bench.py 2>/dev/null | grep loss
