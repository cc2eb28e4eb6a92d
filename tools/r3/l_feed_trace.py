#!/usr/bin/env python3
"""train() fed by DeviceFeeder for a few steps (for rocprofv3 --kernel-trace --memory-copy-trace)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import bench
a = bench.parse()
a.dtype = 'f32'
r = bench.train_loop_rates(a, torch.device('cuda', 0), steps=40, warm=10)
print(r)
