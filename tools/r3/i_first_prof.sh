#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3i; mkdir -p $OUT
for dt in f32 bf16s; do
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s_$dt -o s -- python3 tools/conv_bench.py --dtype $dt --reps 5 > $OUT/s_$dt.log 2>&1
cp $(find $OUT/s_$dt -name "*kernel_stats.csv") $OUT/stats_$dt.csv
rm -rf $OUT/s_$dt
grep "first_\|gconv_kernel<\|wgrad_flat" $OUT/stats_$dt.csv | cut -c1-160
done
