#!/bin/bash
# final numbers of the round: the driver's command, rocprofv3 stats of the same command, PMC passes
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r3o; mkdir -p $OUT
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
tail -4 $OUT/bench.err
python3 -c "
import json;d=json.load(open('$OUT/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['step'])
print(json.dumps(d['roofline']['hbm'], indent=0)[:1500])
print(d['other_modes']); print(d['train_loop']); print(d['cpu_baseline'])"
DVSOF_FORCE_DIST=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop > $OUT/bench_dist1.json 2> $OUT/bench_dist1.err || { tail -5 $OUT/bench_dist1.err; exit 1; }
python3 -c "
import json;d=json.load(open('$OUT/bench_dist1.json')); print('1-rank RCCL group:', d['value'], d['ms_per_step'], d['config']['launch'])"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-modes --no-train-loop > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 tools/timeline.py $(find $OUT/stats -name "*kernel_trace.csv" | head -1) 8 > $OUT/timeline.txt 2>&1
rm -rf $OUT/stats
python3 tools/exec_nodes.py > $OUT/exec_nodes.txt 2>/dev/null
DVSOF_WGRAD_STREAM=0 python3 tools/conv_bench.py > $OUT/conv_per_launch_serial.txt 2>/dev/null
for pass in fetch:FETCH_SIZE write:WRITE_SIZE "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  name=${pass%%:*}; ctr=${pass#*:}
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_$name -o pmc -- python3 $R/tools/conv_bench.py --reps 2 > $OUT/pmc_$name.log 2>&1 || { tail -5 $OUT/pmc_$name.log; exit 1; }
done
python3 tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/traffic_pmc.csv
python3 tools/pmc_mfma.py $OUT/pmc_mfma > $OUT/mfma_pmc.csv
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma
head -12 $OUT/kernel_stats.csv | cut -c1-150
