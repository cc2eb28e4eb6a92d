#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r3z; mkdir -p $OUT
export DVSOF_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_PORT=29617 HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2 3; do
  python3 tests/capture_child.py big:f32 > $OUT/big_$i.out 2> $OUT/big_$i.err; echo "run $i rc=$?"; tail -c 300 $OUT/big_$i.out
done
head -40 $OUT/big_1.err
