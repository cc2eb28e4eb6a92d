#!/usr/bin/env python3
"""Capture one piece of the inference path in a HIP graph and replay it
(diagnostic): python tools/graph_probe.py conv|head|voxel|predictor"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def capture(fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    torch.cuda.synchronize()
    print('captured', flush=True)
    g.replay()
    torch.cuda.synchronize()
    print('replayed', flush=True)
    return out


def main():
    what = sys.argv[1]
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    with torch.no_grad():
        if what == 'conv':
            from dvs_of_training_framework_amd import conv as C
            x = torch.randn(8, 16, 16, 512, device=dev)
            w = torch.randn(512, 3, 3, 512, device=dev) * 0.02
            b = torch.zeros(512, device=dev)
            d = C.make_desc([(x, 512, C.NHWC)], 8, 16, 16, 512, 3, 1, 1, False, C.ACT_RELU)
            ref = C.conv_fwd(d, w, b, dev)[0].clone()
            out = capture(lambda: C.conv_fwd(d, w, b, dev)[0])
            print('max diff', float((out - ref).abs().max()))
        elif what == 'head':
            from dvs_of_training_framework_amd import conv as C
            x = torch.randn(8, 64, 64, 64, device=dev)
            w = torch.randn(2, 64, device=dev)
            b = torch.zeros(2, device=dev)
            ref = C.head_fwd(x, w, b, 8, 64, 64, 64).clone()
            out = capture(lambda: C.head_fwd(x, w, b, 8, 64, 64, 64))
            print('max diff', float((out - ref).abs().max()))
        elif what == 'voxel':
            from dvs_of_training_framework_amd import synthetic
            from dvs_of_training_framework_amd.voxel import voxelize
            ev = {k: torch.from_numpy(v).to(dev) for k, v in
                  synthetic.make_events(np.random.default_rng(1), 1, 256, 256, 65536).items()}
            t0, t1 = torch.zeros(1, device=dev), torch.full((1,), 0.04, device=dev)
            ref = voxelize(ev, t0, t1, 1, 5, 256, 256).clone()
            out = capture(lambda: voxelize(ev, t0, t1, 1, 5, 256, 256))
            print('max diff', float((out - ref).abs().max()))
        elif what in ('model', 'model_nocache'):
            from dvs_of_training_framework_amd import synthetic
            from dvs_of_training_framework_amd.net import Model
            m = Model(dev, event_representation_depth=5).eval()
            ev = {k: torch.from_numpy(v).to(dev) for k, v in
                  synthetic.make_events(np.random.default_rng(1), 1, 256, 256, 65536).items()}
            ts = torch.tensor([0.0, 0.04], device=dev)
            sidx = torch.tensor([0, 0], device=dev)
            m(ev, ts, sidx, (256, 256), batch_size=1)        # strict, eager
            m.strict = False
            if what == 'model':
                m(ev, ts, sidx, (256, 256), batch_size=1)    # builds the layout cache eagerly
            ref = [f.clone() for f in m(ev, ts, sidx, (256, 256), batch_size=1)[0]] \
                if what == 'model' else None
            out = capture(lambda: m(ev, ts, sidx, (256, 256), batch_size=1)[0]) \
                if what == 'model' else None
            if what == 'model_nocache':      # cache tensors created INSIDE the capture
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    out = m(ev, ts, sidx, (256, 256), batch_size=1)[0]
                torch.cuda.synchronize()
                print('captured', flush=True)
                g.replay()
                torch.cuda.synchronize()
                print('replayed', flush=True)
            if ref is not None:
                print('max diff', max(float((a - b).abs().max()) for a, b in zip(out, ref)))
        elif what == 'of':
            from dvs_of_training_framework_amd.of import OpticalFlow
            rng = np.random.default_rng(0)
            n, H, W = 65536, 256, 256
            evs = [(rng.integers(0, W, n), rng.integers(0, H, n), np.sort(rng.random(n) * 0.04),
                    rng.integers(0, 2, n) * 2 - 1)]
            of = OpticalFlow((H, W), model=None, graph=True, event_representation_depth=5)
            a = of(evs, [0.0], [0.04])
            print('first call ok', flush=True)
            b = of(evs, [0.0], [0.04])
            print('second call ok', float(np.abs(a - b).max()), flush=True)
        elif what == 'predictor':
            from dvs_of_training_framework_amd.predictor import Predictor
            net = Predictor(5).cuda().eval()
            x = torch.randn(1, 5, 256, 256, device=dev)
            ref = [f.clone() for f in net(x)]
            out = capture(lambda: net(x))
            print('max diff', max(float((a - b).abs().max()) for a, b in zip(out, ref)))


if __name__ == '__main__':
    main()
