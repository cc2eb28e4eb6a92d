#!/bin/bash
# usage (GPU box, repo root): tools/gpu_trace.sh <stage> <bench args...> -- rocprofv3 kernel trace (+ memory copies) of bench.py
S=${1:-x}; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/${S}_trace -o t -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline "$@" > $O/${S}_trace.log 2>&1
echo rc=$?; tail -2 $O/${S}_trace.log | cut -c1-300; ls $O/${S}_trace
