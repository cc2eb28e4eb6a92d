#!/bin/bash
# Round 3 item 2: where the bf16 modes spend their step (rocprofv3 stats, PMC passes, executor plan)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r3f; mkdir -p $OUT
for dt in ${DTYPES:-bf16s bf16}; do
  python3 tools/exec_nodes.py --dtype $dt > $OUT/exec_nodes_$dt.txt 2> $OUT/exec_nodes_$dt.err || { tail -5 $OUT/exec_nodes_$dt.err; exit 1; }
  DVSOF_WGRAD_STREAM=0 python3 tools/conv_bench.py --dtype $dt > $OUT/conv_per_launch_serial_$dt.txt 2>/dev/null || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$dt -o s -- python3 $R/bench.py --dtype $dt --steps 10 --warmup 3 --no-cpu-baseline --no-other-modes --no-train-loop > $OUT/stats_$dt.log 2>&1 || { tail -5 $OUT/stats_$dt.log; exit 1; }
  cp $(find $OUT/stats_$dt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$dt.csv
  python3 tools/timeline.py $(find $OUT/stats_$dt -name "*kernel_trace.csv" | head -1) > $OUT/timeline_$dt.txt 2>&1
  rm -rf $OUT/stats_$dt
  for pass in fetch:FETCH_SIZE write:WRITE_SIZE "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
    name=${pass%%:*}; ctr=${pass#*:}
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_${name}_$dt -o pmc -- python3 $R/tools/conv_bench.py --dtype $dt --reps 2 > $OUT/pmc_${name}_$dt.log 2>&1 || { tail -5 $OUT/pmc_${name}_$dt.log; exit 1; }
  done
  python3 tools/pmc_traffic.py $OUT/pmc_fetch_$dt $OUT/pmc_write_$dt > $OUT/traffic_pmc_$dt.csv
  python3 tools/pmc_mfma.py $OUT/pmc_mfma_$dt > $OUT/mfma_pmc_$dt.csv
  rm -rf $OUT/pmc_fetch_$dt $OUT/pmc_write_$dt $OUT/pmc_mfma_$dt
  echo "== $dt"; head -25 $OUT/kernel_stats_$dt.csv | cut -c1-160
done
ls -la $OUT
