#!/bin/bash
# probe build: what the f32 patch-resident forward spends its time on (results are wrong by construction)
cd "${GRAFT_REPO_ROOT:-.}"
export DVSOF_PROBE_LIB=1 DVSOF_WGRAD_STREAM=0
for d in ${DBGS:-0 1 2 8 16 18 19 27}; do
  DVSOF_FWD_PATCH_DBG=$d timeout -k 10 200 python3 tools/conv_bench.py > /tmp/c.txt 2>/dev/null || exit 1
  echo "dbg=$d finest fwd us: $(grep '^fwd' /tmp/c.txt | tail -1 | awk '{print $8}')"
done
