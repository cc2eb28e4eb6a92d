#!/bin/bash
cd $GRAFT_REPO_ROOT
# (probes live in the probe build: make -C dvs_of_training_framework_amd/csrc probes)
export DVSOF_WGRAD_STREAM=0
for d in 0 1 2 3; do
  echo "== DVSOF_GCONV_DBG=$d"; DVSOF_PROBE_LIB=1 DVSOF_GCONV_DBG=$d python3 tools/conv_bench.py 2>/dev/null | awk '$1=="fwd"||$1=="dgrad"{print}' | head -24
done
