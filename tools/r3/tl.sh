#!/bin/bash
# per-queue timeline of one executor step (f32 default command) + bf16s
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=$PWD; OUT=$R/gpurun_out/r3tl; mkdir -p $OUT
for dt in f32 bf16s; do
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o s -- python3 $R/bench.py --dtype $dt --steps 10 --warmup 3 --no-cpu-baseline --no-other-modes --no-train-loop --no-roofline > $OUT/tr.log 2>&1 || { tail -5 $OUT/tr.log; exit 1; }
python3 tools/timeline.py $(find $OUT/tr -name "*kernel_trace.csv" | head -1) > $OUT/timeline_$dt.txt 2>&1
rm -rf $OUT/tr
head -3 $OUT/timeline_$dt.txt
done
