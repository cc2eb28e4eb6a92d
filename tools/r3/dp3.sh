#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for i in 1 2 3; do
timeout -k 10 1100 python3 -m pytest tests/test_gpu_capture.py tests/test_gpu_configs.py -x -q -k "parallel or rccl or accumulation or optimizer_inside" 2>&1 | tail -1
done
bash tools/r3/dp2.sh | grep "rc=" | sort | uniq -c
