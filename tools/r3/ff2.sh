#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r3ff; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_conv.py -x -q -k "(fwd_dgrad_wgrad and f32) or finest or folded" > $OUT/pytest.txt 2>&1; rc=$?
tail -5 $OUT/pytest.txt
[ $rc -ne 0 ] && exit $rc
bash tools/r3/fq_probe.sh
