#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r3s; mkdir -p $OUT
for t in 0 1 2 3 5; do
  E=""; [ $t -ne 0 ] && E="DVSOF_GCONV_TILE=$t"
  env $E DVSOF_WGRAD_STREAM=0 python3 tools/conv_bench.py --dtype bf16s > $OUT/conv_t$t.txt 2>/dev/null || { echo "tile $t failed"; continue; }
  echo "== gconv tile $t: $(tail -1 $OUT/conv_t$t.txt)"; grep "^fwd\|^dgrad" $OUT/conv_t$t.txt | awk '{printf "%s ", $8} END {print ""}'
done
for t in 1 2 3 4 5; do
  env DVSOF_WGRAD_TILE=$t DVSOF_WGRAD_STREAM=0 python3 tools/conv_bench.py --dtype bf16s > $OUT/conv_w$t.txt 2>/dev/null || { echo "wtile $t failed"; continue; }
  echo "== wgrad tile $t: $(tail -1 $OUT/conv_w$t.txt)"; grep "^wgrad" $OUT/conv_w$t.txt | awk '{printf "%s ", $8} END {print ""}'
done
