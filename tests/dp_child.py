"""Child process of tests/test_gpu_configs.py::test_rccl_bucket_path_...:
runs the predictor backward at configs[2]'s per-GPU workload (B=8,
256x256x5) (a) plain, one stream, no reducer; (b) with the RCCL bucket
reducer of a 1-rank group and both backward streams; (c) additionally with
the optimizer fused into the bucket hooks.  Prints one JSON line."""
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from dvs_of_training_framework_amd import parallel  # noqa: E402
from dvs_of_training_framework_amd.optim import FusedAdamW  # noqa: E402
from dvs_of_training_framework_amd.predictor import Predictor  # noqa: E402


def main():
    rank, local, world = parallel.init_distributed('cuda')
    dev = torch.device('cuda', local)
    B, C, H, W = 8, 5, 256, 256
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(B, C, H, W, device=dev, generator=g)
    seeds = [torch.randn(B, 2, H // s, W // s, device=dev, generator=g) * (0.5 / s)
             for s in (8, 4, 2, 1)]

    def fresh():
        torch.manual_seed(21)
        return Predictor(C).to(dev)

    def run(reducer, two_streams, fused, steps=1):
        os.environ['DVSOF_WGRAD_STREAM'] = '1' if two_streams else '0'
        net = fresh()
        net.reducer = reducer
        opt = FusedAdamW(net.parameters(), lr=1e-3, weight_decay=1e-4, amsgrad=True)
        if fused:
            opt.fuse_into_backward(net, flush_at=None)
        grads = None
        for _ in range(steps):
            flows = net(x)
            torch.autograd.backward(flows, seeds)
            if reducer is not None:
                reducer.wait()
            if not fused:
                grads = [p.grad.detach().clone() for p in net.parameters()]
            opt.step()
            opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        return grads, [p.detach().clone() for p in net.parameters()]

    g_plain, w_plain = run(None, False, False, steps=2)
    red = parallel.GradReducer()
    assert red.active()
    calls = []
    orig = red.bucket_ready

    def counted(flat, after=None):
        calls.append(flat.numel())
        return orig(flat, after)
    red.bucket_ready = counted
    g_red, w_red = run(red, True, False, steps=2)
    g_red2, w_red2 = run(red, True, False, steps=2)
    _, w_fused = run(red, True, True, steps=2)
    # the same exchange through the C ABI's own communicator
    direct = parallel.GradReducer(direct=True)
    g_dir, w_dir = run(direct, True, False, steps=2)
    _, w_dir_fused = run(direct, True, True, steps=2)
    direct_bytes = direct.bytes_reduced
    direct.close()
    same = lambda a, b: all(torch.equal(u, v) for u, v in zip(a, b))  # noqa: E731
    nparam = sum(p.numel() for p in fresh().parameters())
    print(json.dumps({
        'backend': dist.get_backend(), 'world': world,
        'buckets_reduced': len(calls), 'reduced_runs': 6,
        'bytes_reduced': red.bytes_reduced, 'grad_bytes': 4 * nparam,
        'grads_bit_identical': same(g_plain, g_red) and same(w_plain, w_red),
        'repeat_bit_identical': same(g_red, g_red2) and same(w_red, w_red2),
        'fused_weights_bit_identical': same(w_plain, w_fused),
        'direct_bit_identical': same(g_plain, g_dir) and same(w_plain, w_dir) and
        same(w_plain, w_dir_fused),
        'direct_bytes': direct_bytes}), flush=True)
    torch.cuda.synchronize()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
