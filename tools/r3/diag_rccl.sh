#!/bin/bash
# Round 3, item 1a: why does the C-ABI collective path cost +53 % on one rank?
# kernel + HIP API traces of bench.py --eager in three configurations.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r3a
mkdir -p $OUT
ARGS="bench.py --eager --steps 10 --warmup 2 --no-roofline --no-other-modes --no-cpu-baseline"
run() {   # name, env...
  name=$1; shift
  ( export "$@"; python3 $ARGS > $OUT/$name.json 2> $OUT/$name.err ) || return 1
  ( export "$@"; rocprofv3 --kernel-trace --hip-trace --memory-copy-trace --output-format csv -d $OUT/$name -o t -- python3 $ARGS > $OUT/$name.prof.json 2> $OUT/$name.prof.err ) || return 1
  f=$(find $OUT/$name -name "*kernel_trace.csv" | sort | tail -1)
  python3 tools/timeline.py $f > $OUT/$name.timeline.txt 2>&1
  python3 tools/timeline.py $f --brief > $OUT/$name.brief.txt 2>&1
  cat $OUT/$name.json | head -c 300; echo; cat $OUT/$name.brief.txt
}
run nodist DVSOF_X=0 && run dist DVSOF_FORCE_DIST=1 && run direct DVSOF_FORCE_DIST=1 DVSOF_DIRECT_RCCL=1
# keep what was asked for small: HIP API summary per configuration
for n in nodist dist direct; do
  f=$(find $OUT/$n -name "*hip_api_trace.csv" | sort | tail -1)
  [ -n "$f" ] && python3 - "$f" > $OUT/$n.hip_api.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
c = collections.Counter(r['Function'] for r in rows)
t = collections.Counter()
for r in rows:
    t[r['Function']] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for k, v in c.most_common(40):
    print(f'{v:8d} {t[k]/1e6:10.2f} ms {k}')
PY
done
du -sh $OUT
