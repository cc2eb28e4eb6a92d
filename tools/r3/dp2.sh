#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for i in 1 2 3 4 5 6 7 8; do
DVSOF_FORCE_DIST=1 timeout -k 10 200 python3 bench.py --dtype f32 --steps 60 --warmup 5 --no-roofline --no-other-modes --no-cpu-baseline --no-train-loop > gpurun_out/dp2.out 2> gpurun_out/dp2.err; echo "rc=$?"; tail -c 300 gpurun_out/dp2.out | cut -c1-200; grep -v "^frame\|^$" gpurun_out/dp2.err | tail -5
done
