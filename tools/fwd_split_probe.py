"""Would the forward gain from running the two halves of the batch on two streams?
Graph-captured forward of the predictor: batch 8 on one stream against 2 x batch 4 on two
streams (same weights; inference forward, no gradient forms).  python tools/fwd_split_probe.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch  # noqa: E402

from dvs_of_training_framework_amd.predictor import Predictor  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
net = Predictor(5, torch.nn.ReLU()).to(dev)
x8 = torch.randn(8, 5, 256, 256, device=dev)
xa, xb = x8[:4].contiguous(), x8[4:].contiguous()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def one():
    return net(x8)


def two():
    cur = torch.cuda.current_stream()
    s2.wait_stream(cur)
    ya = net(xa)
    with torch.cuda.stream(s2):
        yb = net(xb)
    cur.wait_stream(s2)
    return ya, yb


def bench(fn, name):
    with torch.no_grad():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s1):
            with torch.cuda.graph(g, stream=s1, capture_error_mode='thread_local'):
                out = fn()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        print(f'{name}: {e0.elapsed_time(e1) / 30 * 1e3:.1f} us per forward')
    return out


o1 = bench(one, 'batch 8, one stream      ')
o2 = bench(two, '2 x batch 4, two streams ')
err = max(float((torch.cat([a, b]) - c).abs().max()) for a, b, c in zip(o2[0], o2[1], o1))
print('max |difference| of the flows:', err)
