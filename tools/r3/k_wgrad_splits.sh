#!/bin/bash
# bf16 twins: K splits of the weight-gradient GEMMs (the occupancy model prices a step at the f32 matrix rate)
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r3k; mkdir -p $OUT
python3 tools/r3/j_compact.py 2>/dev/null | tee $OUT/compact.txt
for s in 0 2 4 8 16; do
  E=""; [ $s -ne 0 ] && E="DVSOF_WGRAD_SPLITS=$s"
  env $E DVSOF_WGRAD_STREAM=0 python3 tools/conv_bench.py --dtype bf16s > $OUT/conv_s$s.txt 2>/dev/null || exit 1
  echo "== splits $s"; grep "^wgrad" $OUT/conv_s$s.txt | awk '{printf "%s:%s ", $2, $8} END {print ""}'; tail -1 $OUT/conv_s$s.txt
done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $PWD/$OUT/feed -o t -- python3 tools/r3/l_feed_trace.py > $OUT/feed.log 2>&1
f=$(find $OUT/feed -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $f 30 > $OUT/feed_timeline.txt 2>&1
head -4 $OUT/feed_timeline.txt; grep copy $OUT/feed_timeline.txt | head -20
grep "feeder\|samples_per_s" $OUT/feed.log | tail -3
rm -rf $OUT/feed
echo "== loss probes B=64 256x256 (probe build)"
for d in 0 256 512 128 16; do echo -n "DBG=$d: "; DVSOF_LOSS_DBG=$d DVSOF_PROBE_LIB=1 python3 tools/loss_probe.py 64 256 256 2>/dev/null; done
