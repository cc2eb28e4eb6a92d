#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r3fo; mkdir -p $OUT
export DVSOF_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_PORT=29611 HSA_ENABLE_IPC_MODE_LEGACY=0
for sc in big:f32 big:f32:fused; do
timeout -k 10 300 python3 tests/capture_child.py $sc > $OUT/child_$sc.out 2> $OUT/child_$sc.err; echo "$sc rc=$?"
tail -c 600 $OUT/child_$sc.out; echo; grep -v "^frame\|^$" $OUT/child_$sc.err | tail -12
done
