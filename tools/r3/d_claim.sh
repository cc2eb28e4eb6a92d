#!/bin/bash
# 1a: the C-ABI collective path with / without claiming the step's streams before the communicators
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r3d; mkdir -p $OUT
ARGS="bench.py --steps 40 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline"
b() {  # label env...
  l=$1; shift
  env "$@" python3 $ARGS > $OUT/$l.json 2> $OUT/$l.err || { tail -5 $OUT/$l.err; return 1; }
  python3 -c "import json;d=json.load(open('$OUT/$l.json'));print('$l', d['ms_per_step'], d['value'], d['config']['launch'][:60])"
}
b nodist_exec X=0 && b nodist_eager DVSOF_EAGER=1 \
 && b eager_direct_noclaim DVSOF_FORCE_DIST=1 DVSOF_EAGER=1 DVSOF_DIRECT_RCCL=1 DVSOF_NO_STREAM_CLAIM=1 \
 && b eager_direct_claim DVSOF_FORCE_DIST=1 DVSOF_EAGER=1 DVSOF_DIRECT_RCCL=1 \
 && b eager_torch_noclaim DVSOF_FORCE_DIST=1 DVSOF_EAGER=1 DVSOF_NO_STREAM_CLAIM=1 \
 && b eager_torch_claim DVSOF_FORCE_DIST=1 DVSOF_EAGER=1 \
 && b exec_dist_noclaim DVSOF_FORCE_DIST=1 DVSOF_NO_STREAM_CLAIM=1 \
 && b exec_dist_claim DVSOF_FORCE_DIST=1 \
 && b exec_dist_claim_bf16s DVSOF_FORCE_DIST=1 DVSOF_DTYPE=bf16s && b exec_nodist_bf16s DVSOF_DTYPE=bf16s || exit 1
for l in eager_direct_noclaim eager_direct_claim exec_dist_claim; do
  case $l in
    eager_direct_noclaim) E="DVSOF_FORCE_DIST=1 DVSOF_EAGER=1 DVSOF_DIRECT_RCCL=1 DVSOF_NO_STREAM_CLAIM=1";;
    eager_direct_claim) E="DVSOF_FORCE_DIST=1 DVSOF_EAGER=1 DVSOF_DIRECT_RCCL=1";;
    exec_dist_claim) E="DVSOF_FORCE_DIST=1";;
  esac
  ( export $E; rocprofv3 --kernel-trace --output-format csv -d $OUT/t_$l -o t -- python3 bench.py --steps 10 --warmup 2 --no-roofline --no-other-modes --no-cpu-baseline > $OUT/t_$l.json 2> $OUT/t_$l.err )
  f=$(find $OUT/t_$l -name "*kernel_trace.csv" | sort | tail -1)
  python3 tools/timeline.py $f --brief > $OUT/t_$l.brief.txt 2>&1
  python3 tools/timeline.py $f > $OUT/t_$l.timeline.txt 2>&1
  echo "== $l"; cat $OUT/t_$l.brief.txt
  rm -rf $OUT/t_$l
done
