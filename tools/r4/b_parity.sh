#!/bin/bash
OUT=gpurun_out/r4b; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_loss.py -x -q -k "config1 or config2_bf16 or config3 or baseline_size" > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.txt
timeout -k 10 600 python -m pytest tests/test_gpu_capture.py -x -q -k "non_identity" > $OUT/pytest2.txt 2>&1; echo "pytest2 rc=$?"; tail -3 $OUT/pytest2.txt
for v in base wpe4 wpe5; do
  if [ $v = base ]; then unset DVSOF_LIB_PATH; else export DVSOF_LIB_PATH=dvs_of_training_framework_amd/csrc/variants/$v/libdvsof_hip.so; fi
  echo "== $v"; timeout -k 10 300 python tools/hbm_bench.py 2>&1 | grep "loss"
done
