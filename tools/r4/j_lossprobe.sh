#!/bin/bash
# loss path at batch 64 under the probe build: which part of the sweep costs what
for bits in 0 1 2 3 4 8 128 512 640; do
  echo -n "DBG=$bits: "; DVSOF_LOSS_DBG=$bits DVSOF_PROBE_LIB=1 timeout -k 10 120 python tools/loss_probe.py 64 256 256 2>&1 | tail -1
done
