#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r3e; mkdir -p $OUT
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-modes --no-roofline > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python3 -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'], json.dumps(d['train_loop'], indent=1))"
