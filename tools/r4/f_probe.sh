#!/bin/bash
export DVSOF_WGRAD_STREAM=0
for v in base fm_nosched; do
  if [ $v = base ]; then unset DVSOF_LIB_PATH; else export DVSOF_LIB_PATH=dvs_of_training_framework_amd/csrc/variants/$v/libdvsof_hip.so; fi
  echo "== $v"; timeout -k 10 300 python tools/conv_bench.py 2>&1 | grep "^fwd" | awk '$6==1 {printf "%s ", $8} END{print ""}'
done
unset DVSOF_LIB_PATH
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py -x -q -k "(test_conv_fwd_dgrad_wgrad and f32)" 2>&1 | tail -2
