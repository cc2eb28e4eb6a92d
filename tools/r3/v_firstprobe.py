#!/usr/bin/env python3
"""first_fwd_kernel alone at the benchmark shape (B=8, 256x256x5): us per call (probe bits via DVSOF_FIRST_DBG)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from dvs_of_training_framework_amd import conv as C
from tools.hbm_bench import timeit
B, Cin, H, W = 8, 5, 256, 256
x = torch.randn(B, Cin, H, W, device='cuda')
w = torch.randn(64, 3, 3, Cin, device='cuda')
b = torch.randn(64, device='cuda')
d = C.make_desc([(x, Cin, C.NCHW)], B, H, W, 64, 3, 2, 1, False, C.ACT_RELU)
us = timeit(lambda: C.conv_fwd(d, w, b, 'cuda'), n=50)
print(f'first fwd: {us:.1f} us')
