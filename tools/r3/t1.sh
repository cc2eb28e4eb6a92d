#!/bin/bash
# one executor step of a dtype as a per-queue timeline + the executor's plan
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=$PWD; dt=${DT:-bf16s}
OUT=$R/gpurun_out/r3t; mkdir -p $OUT
python3 tools/exec_nodes.py --dtype $dt > $OUT/exec_nodes_$dt.txt 2> $OUT/exec_nodes_$dt.err || { tail -5 $OUT/exec_nodes_$dt.err; exit 1; }
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o s -- python3 $R/bench.py --dtype $dt --steps 10 --warmup 3 --no-cpu-baseline --no-other-modes --no-train-loop --no-roofline > $OUT/tr.log 2>&1 || { tail -5 $OUT/tr.log; exit 1; }
python3 tools/timeline.py $(find $OUT/tr -name "*kernel_trace.csv" | head -1) > $OUT/timeline_$dt.txt 2>&1
rm -rf $OUT/tr
head -3 $OUT/timeline_$dt.txt
