#!/usr/bin/env python3
"""Fused loss timing at one shape (probe runs: DVSOF_LOSS_DBG bits).  Prints us per call."""
import os
import sys
from pathlib import Path
if os.environ.get("DVSOF_LOSS_DBG"):     # the probes exist in the probe build only (make -C .../csrc probes)
    os.environ.setdefault("DVSOF_PROBE_LIB", "1")
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tools.hbm_bench import loss_case  # noqa: E402
B, H, W = (int(v) for v in sys.argv[1:4])
us, alg = loss_case(B, H, W)
print(f'{us:.1f} us  {alg / us / 1e3:.0f} GB/s')
