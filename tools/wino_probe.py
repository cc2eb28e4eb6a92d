#!/usr/bin/env python3
"""The Winograd component GEMM of one residual layer (512 -> 512, 3x3, 16x16,
batch B) in a loop, for rocprofv3 / DVSOF_GCONV_DBG probes:
python tools/wino_probe.py [B] [fwd|dgrad|wgrad]"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from dvs_of_training_framework_amd import conv as C  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    kind = sys.argv[2] if len(sys.argv) > 2 else 'fwd'
    dev = torch.device('cuda', 0)
    cin = cout = 512
    x = torch.randn(B, 16, 16, cin, device=dev)
    w = torch.randn(cout, 3, 3, cin, device=dev) * 0.05
    bias = torch.zeros(cout, device=dev)
    d = C.make_desc([(x, cin, C.NHWC)], B, 16, 16, cout, 3, 1, 1, False, C.ACT_RELU)
    w_fwd, w_dg = C.prepare(d, w, True)
    g = torch.randn(B, 16, 16, cout, device=dev)
    gx = torch.empty(B, 16, 16, cin, device=dev)
    dw = torch.empty(cout, 3, 3, cin, device=dev)
    db = torch.empty(cout, device=dev)
    fn = {'fwd': lambda: C.conv_fwd(d, w_fwd, bias, dev),
          'dgrad': lambda: C.conv_dgrad(d, w_dg, g, [dict(p=gx, actsrc=x)], C.ACT_RELU),
          'wgrad': lambda: C.conv_wgrad(d, g, dw, db)}[kind]
    for _ in range(30):
        fn()
    torch.cuda.synchronize()


if __name__ == '__main__':
    main()
