#!/bin/bash
# forced forward / data-gradient tiles, per launch alone (f32)
cd "${GRAFT_REPO_ROOT:-.}"
export DVSOF_WGRAD_STREAM=0
for t in 0 1 2 4; do
  E="DVSOF_X=1"; [ $t != 0 ] && E="DVSOF_GCONV_TILE=$t"
  env $E timeout -k 10 200 python3 tools/conv_bench.py > /tmp/c.txt 2>/dev/null || { echo "tile $t failed"; continue; }
  echo "tile=$t fwd:   $(grep '^fwd' /tmp/c.txt | awk '{printf "%s ", $8}')"
  echo "tile=$t dgrad: $(grep '^dgrad' /tmp/c.txt | awk '{printf "%s ", $8}')"
done
