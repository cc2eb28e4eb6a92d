#!/bin/bash
# One script for the GPU box (gpurun -- bash tools/gpu.sh <command> [args]); replaces the one-off
# scripts of earlier rounds.  Everything is written under gpurun_out/<tag>/ (tag = $TAG or the
# command name); copy what is to be kept into profiles/.
#   tests [pytest -k expression]      GPU tests (all, or a selection)
#   bench [bench.py args]             the driver's command (default) or any other bench run
#   quick [ENV=VAL ...]               short headline run (no roofline / modes / baseline) under an environment
#   conv [ENV=VAL ...]                per-launch conv table of one step, one stream
#   variants <name> [<name> ...]      `conv` under variant libraries built by tools/variant.sh
#   hbm [ENV=VAL ...]                 voxeliser / loss call paths at the BASELINE.json sizes
#   lossprobe B H W bits...           loss path under the probe build's DVSOF_LOSS_DBG bits
#   timeline [bench args]             rocprofv3 kernel trace of a short run -> one step per queue
#   feedtrace [wire|compact]          kernel + memory-copy trace of the train loop fed from host memory
#   nodes [bench args]                the executor's plan (lane, stand-alone us of every kernel)
#   collect <stage>                   everything under profiles/<round>/: tools/collect_profiles.sh +
#                                     HBM-path PMC passes + plan + timeline + loopback / 1-rank-group runs
#   rccl2                             does RCCL accept two ranks on one GPU? (it does not)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=$PWD; cmd=${1:-tests}; shift
O=$R/gpurun_out/${TAG:-$cmd}; mkdir -p $O
envs=(); while [[ "$1" == *=* && "$1" != -* ]]; do envs+=("$1"); shift; done
short="--steps 30 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop"
case $cmd in
tests)
  if [ -n "$1" ]; then timeout -k 10 1100 python -m pytest tests -x -q -m gpu -k "$1" > $O/pytest.txt 2>&1
  else timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; fi
  echo "pytest rc=$?"; tail -8 $O/pytest.txt ;;
bench)
  timeout -k 10 900 env "${envs[@]}" python bench.py "$@" > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
  python - $O/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d['value'], d['unit'], d['ms_per_step'], 'ms |', d['config'].get('launch', '')[:90])
r = d.get('roofline')
if r: print(r['kernel'], r['achieved'], r['frac'], '| step', r['step'])
PY
  ;;
quick)
  timeout -k 10 600 env "${envs[@]}" python bench.py $short "$@" 2> $O/quick.err | tee $O/quick.json | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'], d['ms_per_step'])" ;;
conv)
  timeout -k 10 300 env DVSOF_WGRAD_STREAM=0 "${envs[@]}" python tools/conv_bench.py "$@" > $O/conv.txt 2>&1; cat $O/conv.txt ;;
variants)
  for v in base "$@"; do
    lib=; [ $v != base ] && lib=DVSOF_LIB_PATH=dvs_of_training_framework_amd/csrc/variants/$v/libdvsof_hip.so
    echo "== $v"; timeout -k 10 300 env DVSOF_WGRAD_STREAM=0 $lib python tools/conv_bench.py 2>&1 | awk '$6==1 {printf "%s %s | ", $1, $8} END{print ""}'
  done ;;
hbm)
  timeout -k 10 300 env "${envs[@]}" python tools/hbm_bench.py 2>&1 | tee $O/hbm.txt ;;
lossprobe)
  B=$1; H=$2; W=$3; shift 3
  for bits in "$@"; do echo -n "DBG=$bits: "; DVSOF_LOSS_DBG=$bits DVSOF_PROBE_LIB=1 timeout -k 10 120 python tools/loss_probe.py $B $H $W 2>&1 | tail -1; done ;;
timeline)
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-other-modes --no-roofline --no-train-loop "$@" > $O/trace.log 2>&1 || { tail -3 $O/trace.log; exit 1; }
  cd $R; python3 tools/timeline.py $(find $O/t -name "*kernel_trace.csv") > $O/timeline.txt; rm -rf $O/t; head -4 $O/timeline.txt ;;
feedtrace)
  leg=${1:-wire}
  timeout -k 10 300 python3 tools/feed_trace.py $leg 90 > $O/plain_$leg.json 2> $O/plain.err || { tail -3 $O/plain.err; exit 1; }
  cat $O/plain_$leg.json; echo
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/t -o t -- python3 $R/tools/feed_trace.py $leg 45 > $O/trace.log 2>&1 || { tail -3 $O/trace.log; exit 1; }
  cd $R; python3 tools/feed_timeline.py $O/t > $O/feed_$leg.txt; rm -rf $O/t; head -40 $O/feed_$leg.txt ;;
nodes)
  python3 tools/exec_nodes.py "$@" | tee $O/exec_nodes.txt | tail -3 ;;
collect)
  bash tools/collect_round.sh ;;
rccl2)
  timeout -k 10 200 python tools/rccl_share_gpu.py 2>&1 | grep -v "^$" | tail -8 ;;
*) echo "unknown command $cmd"; exit 2 ;;
esac
