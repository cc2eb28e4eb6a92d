#!/bin/bash
# usage: tools/pmc.sh <outdir-name> "<counters>" -- collects PMC for one conv_bench pass
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT -o pmc -- python3 $GRAFT_REPO_ROOT/tools/conv_bench.py --reps 2 > $OUT.log 2>&1
ls $OUT | head
