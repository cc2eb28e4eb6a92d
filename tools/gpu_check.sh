#!/bin/bash
# usage (GPU box, repo root): tools/gpu_check.sh <stage>  -- tests, bench, kernel stats, HBM paths
S=${1:-x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/${S}_pytest.txt 2>&1
echo "pytest rc=$?" | tee -a $O/${S}_pytest.txt
tail -5 $O/${S}_pytest.txt
timeout -k 10 600 python3 bench.py > $O/${S}_bench.json 2> $O/${S}_bench.err || { echo bench failed; tail -20 $O/${S}_bench.err; exit 1; }
cut -c1-600 $O/${S}_bench.json
timeout -k 10 300 python3 tools/hbm_bench.py > $O/${S}_hbm_kernels.txt 2>&1
DVSOF_LOSS_STRICT=1 timeout -k 10 300 python -m pytest tests/test_gpu_loss.py -m gpu -q -x > $O/${S}_pytest_loss_strict.txt 2>&1; tail -2 $O/${S}_pytest_loss_strict.txt
cat $O/${S}_hbm_kernels.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${S}_stats -o s -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/${S}_stats.log 2>&1
ls $O/${S}_stats | head
