#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r3r; mkdir -p $OUT
python3 tools/exec_nodes.py --dtype bf16s > $OUT/exec_nodes_bf16s.txt 2>/dev/null
tail -1 $OUT/exec_nodes_bf16s.txt
