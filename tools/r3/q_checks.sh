#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r3q; mkdir -p $OUT
echo "fused pyramid:"; python3 tools/r3/p_pyr.py 2>/dev/null
echo "one launch per level:"; DVSOF_LOSS_NO_FUSED_PYRAMID=1 python3 tools/r3/p_pyr.py 2>/dev/null
timeout -k 10 1500 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_configs.py tests/test_gpu_capture.py -x -q -k "cli or rccl or exchange or accumulation" > $OUT/pytest.txt 2>&1 || { tail -30 $OUT/pytest.txt; exit 1; }
tail -2 $OUT/pytest.txt
