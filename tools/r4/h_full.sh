#!/bin/bash
OUT=gpurun_out/r4h; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.txt
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d = json.load(open('gpurun_out/r4h/bench.json'))
print(d['value'], d['ms_per_step'], d['config']['launch'][:100])
r = d['roofline']; print(r['kernel'], r['achieved'], r['frac'], r['avg_launch_us'])
print({k: (v['frac'], v['ms_per_step']) for k, v in r['per_kernel'].items()})
print(r['step'], r['hbm'])
print({k: v.get('samples_per_s') for k, v in d['other_modes'].items() if isinstance(v, dict)})
print(d['cpu_baseline']['value'], d['cpu_baseline']['cpu_model'], d['cpu_baseline']['threads']['1']['value'], {k: v['value'] for k, v in d['cpu_baseline']['other_configs'].items()})
print(d.get('train_loop'))
PY
