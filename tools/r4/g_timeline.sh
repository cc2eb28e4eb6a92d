#!/bin/bash
# kernel trace of the benchmark step -> timeline of one late step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-other-modes --no-roofline --no-train-loop ${BENCH_ARGS} > $O/trace.log 2>&1
cd $R
python3 tools/timeline.py $(find $O/t -name "*kernel_trace.csv") > $O/timeline.txt
rm -rf $O/t
head -4 $O/timeline.txt
