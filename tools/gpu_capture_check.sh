#!/bin/bash
S=${1:-x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
rc=0
for sc in train loop infer; do
  timeout -k 10 300 python tests/capture_child.py $sc > $O/${S}_capture_$sc.txt 2>&1; r=$?; echo "$sc rc=$r"; tail -c 1200 $O/${S}_capture_$sc.txt
  if [ $r -ne 0 ]; then rc=1; break; fi
done
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python3 bench.py --graph --no-cpu-baseline > $O/${S}_bench_graph.json 2> $O/${S}_bench_graph.err || exit 1
cut -c1-330 $O/${S}_bench_graph.json; tail -3 $O/${S}_bench_graph.err
