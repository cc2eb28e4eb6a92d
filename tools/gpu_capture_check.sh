#!/bin/bash
S=${1:-x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_voxel.py tests/test_gpu_quantize.py tests/test_encoding.py tests/test_gpu_model.py -m gpu -q -x > $O/${S}_pytest_a.txt 2>&1; tail -3 $O/${S}_pytest_a.txt
for sc in train loop infer; do
  timeout -k 10 300 python tests/capture_child.py $sc > $O/${S}_capture_$sc.txt 2>&1; echo "$sc rc=$?"; tail -c 1500 $O/${S}_capture_$sc.txt
done
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $O/${S}_bench_eager.json 2> $O/${S}_bench_eager.err; cut -c1-330 $O/${S}_bench_eager.json
timeout -k 10 300 python3 bench.py --graph --no-cpu-baseline > $O/${S}_bench_graph.json 2> $O/${S}_bench_graph.err; cut -c1-330 $O/${S}_bench_graph.json; tail -3 $O/${S}_bench_graph.err
timeout -k 10 300 python3 tools/hbm_bench.py > $O/${S}_hbm_kernels.txt 2>&1; cat $O/${S}_hbm_kernels.txt
