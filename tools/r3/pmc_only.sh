#!/bin/bash
# PMC passes of tools/conv_bench.py per operand mode (separate --pmc runs): HBM-side bytes and matrix-pipe utilisation per kernel template
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=$PWD; OUT=$R/gpurun_out/r3p; mkdir -p $OUT
for dt in ${DTYPES:-f32 bf16s}; do
  for pass in fetch:FETCH_SIZE write:WRITE_SIZE "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
    name=${pass%%:*}; ctr=${pass#*:}
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_${name}_$dt -o pmc -- python3 $R/tools/conv_bench.py --dtype $dt --reps 2 > $OUT/pmc_${name}_$dt.log 2>&1 || { tail -5 $OUT/pmc_${name}_$dt.log; exit 1; }
  done
  python3 tools/pmc_traffic.py $OUT/pmc_fetch_$dt $OUT/pmc_write_$dt > $OUT/traffic_pmc_$dt.csv
  python3 tools/pmc_mfma.py $OUT/pmc_mfma_$dt > $OUT/mfma_pmc_$dt.csv
  rm -rf $OUT/pmc_fetch_$dt $OUT/pmc_write_$dt $OUT/pmc_mfma_$dt
  cat $OUT/traffic_pmc_$dt.csv
done
