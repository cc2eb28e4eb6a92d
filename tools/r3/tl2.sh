#!/bin/bash
# per-queue timeline of one executor step inside a 1-rank RCCL group
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp DVSOF_FORCE_DIST=1
R=$PWD; OUT=$R/gpurun_out/r3tl; mkdir -p $OUT
dt=${DT:-bf16s}
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o s -- python3 $R/bench.py --dtype $dt --steps 10 --warmup 3 --no-cpu-baseline --no-other-modes --no-train-loop --no-roofline > $OUT/tr.log 2>&1 || { tail -5 $OUT/tr.log; exit 1; }
python3 tools/timeline.py $(find $OUT/tr -name "*kernel_trace.csv" | head -1) > $OUT/timeline_dist_$dt.txt 2>&1
rm -rf $OUT/tr
head -5 $OUT/timeline_dist_$dt.txt
