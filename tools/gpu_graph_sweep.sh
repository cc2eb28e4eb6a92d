#!/bin/bash
S=${1:-x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python3 bench.py --graph --no-cpu-baseline --no-roofline > $O/${S}_g_$name.json 2> $O/${S}_g_$name.err || { echo "$name FAILED"; tail -3 $O/${S}_g_$name.err; return; }
  python3 -c "import json,sys; d=json.load(open('$O/${S}_g_$name.json')); print('$name', d['value'], d['ms_per_step'])"
}
run default
run pc0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run pc1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run q1 DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run q2 DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run q4 DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run onestream DVSOF_WGRAD_STREAM=0
run batch1 DEBUG_HIP_GRAPH_BATCH_SIZE=1
run batch64 DEBUG_HIP_GRAPH_BATCH_SIZE=64
python3 bench.py --no-cpu-baseline --no-roofline > $O/${S}_g_eager.json 2>/dev/null; python3 -c "import json; d=json.load(open('$O/${S}_g_eager.json')); print('eager', d['value'], d['ms_per_step'])"
DVSOF_WGRAD_STREAM=0 python3 bench.py --no-cpu-baseline --no-roofline > $O/${S}_g_eager1.json 2>/dev/null; python3 -c "import json; d=json.load(open('$O/${S}_g_eager1.json')); print('eager one stream', d['value'], d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/${S}_gtrace -o t -- python3 $R/bench.py --graph --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $O/${S}_gtrace.log 2>&1
ls $O/${S}_gtrace
