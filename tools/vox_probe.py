#!/usr/bin/env python3
"""Voxeliser timing at one shape (probe runs: DVSOF_VOX_DBG bits).  Prints us per call."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tools.hbm_bench import voxel_case  # noqa: E402
B, C, H, W, n = (int(v) for v in sys.argv[1:6])
us, alg = voxel_case(B, C, H, W, n)
print(f'{us:.1f} us  {alg / us / 1e3:.0f} GB/s')
