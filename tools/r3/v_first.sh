#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 600 python3 -m pytest tests/test_gpu_conv.py -x -q -k "test_conv_fwd_dgrad_wgrad" 2>&1 | tail -1
for d in 0 12; do echo -n "DBG=$d "; DVSOF_PROBE_LIB=1 DVSOF_FIRST_DBG=$d python3 tools/r3/v_firstprobe.py 2>/dev/null; done
echo -n "product lib: "; python3 tools/r3/v_firstprobe.py 2>/dev/null
