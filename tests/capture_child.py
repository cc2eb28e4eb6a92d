"""Child process of tests/test_gpu_capture.py (a GPU fault must not take the
test runner down): runs one scenario and prints one JSON line.

  train   N training steps eagerly and N as replays of the captured step by
          the step executor (no host sync between steps), same seeds / batches
          / LR schedule: bitwise equality
  train_graph   the same through hipGraphLaunch (executor=False)
  bind    two captured steps bound to resident batches, replayed alternately
  loop    training.train(capture=False) vs train(capture=True): logged scalars
  infer   OpticalFlow(graph=True): 30 replays back to back WITHOUT reading the
          result in between, then compared with the eager wrapper
  big:<dtype>   executor vs eager at the BENCHMARK shape (B=8, 256x256x5,
          65 536 events per sample): the Winograd / K-split / twins kernels
          of that plan; with DVSOF_FORCE_DIST=1 in the environment both sides
          run the data-parallel exchange (eager: torch.distributed on the
          exchange stream; executor: marks -> ncclAllReduce from the C call)
  accum[:dp]  train(accumulation_steps=3) eager vs capture=True (roles first /
          middle / last), optionally under the 1-rank reducer
  compact   train(capture=True) on compact (9 B/event) batches
  grow    a batch with more events than the captured capacity: re-recorded
  feed[:k]  train() fed by feed.DeviceFeeder (k = accumulation steps)
  fail    recording raises: training continues eagerly, same results
  audit   pointer audit of a captured step (every kernel argument pointer in
          the graph pool or in an object the step keeps alive)
"""
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import torch  # noqa: E402

from dvs_of_training_framework_amd import synthetic  # noqa: E402


def unique_pixel_batch(seed, B, H, W, n):
    """Every event of a sample on its own pixel: each voxel then receives
    contributions of ONE event, so the voxeliser's float atomics cannot
    reorder anything and whole steps are bitwise reproducible."""
    b = synthetic.make_batch(seed, B, H, W, n)
    rng = np.random.default_rng(seed + 99)
    ev = b['events']
    for s in range(B):
        m = ev['sample_index'] == s
        pix = rng.permutation(H * W)[:int(m.sum())]
        ev['x'][m], ev['y'][m] = pix % W, pix // W
    return b


def make(seed=5, C=5, dtype='f32'):
    from dvs_of_training_framework_amd.loss import init_losses
    from dvs_of_training_framework_amd.net import Model
    from dvs_of_training_framework_amd.optim import FusedAdamW
    torch.manual_seed(seed)
    model = Model('cuda', event_representation_depth=C, compute_dtype=dtype)
    model.train()
    opt = FusedAdamW(model.predictor.parameters(), lr=1e-3, weight_decay=1e-4, amsgrad=True)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 0.5 ** (s / 3) if s > 1 else (s + 1) / 2)
    return model, opt, sched, init_losses


def scenario_train(executor=True):
    from dvs_of_training_framework_amd.capture import CapturedTrainStep
    from dvs_of_training_framework_amd.loss import unit_backward
    from dvs_of_training_framework_amd.timer import FakeTimer
    from dvs_of_training_framework_amd.training import process_minibatch
    B, H, W, steps = 2, 64, 64, 12
    counts = [4096, 3500, 4096, 2800]
    batches = [synthetic.to_torch(unique_pixel_batch(70 + i, B, H, W, counts[i % 4]), 'cuda')
               for i in range(4)]

    def eager():
        model, opt, sched, init_losses = make()
        ev = init_losses((H, W), B, model, 'cuda', sequence_length=1)
        losses = []
        for i in range(steps):
            opt.zero_grad(set_to_none=True)
            loss, _, _ = process_minibatch(model, batches[i % 4], FakeTimer(), 'cuda', True, ev,
                                           [0.5, 1, 1])
            unit_backward(loss)
            model.strict = False
            opt.step()
            sched.step()
            losses.append(loss.detach().clone())
        torch.cuda.synchronize()
        return [float(v) for v in losses], [p.detach().clone() for p in model.parameters()]

    info = {}

    def graphed():
        model, opt, sched, init_losses = make()
        ev = init_losses((H, W), B, model, 'cuda', sequence_length=1)
        step = CapturedTrainStep(model, ev, opt, [0.5, 1, 1], 'cuda', batches[0],
                                 event_capacity=8192, executor=executor)
        info.update(kernels=step.executor.kernels, lanes=step.executor.lanes,
                    lane_kernels=step.executor.lane_kernels, events=step.executor.events,
                    waits=step.executor.waits) if executor else None
        sched.step()
        losses = [step.first_loss]
        terms = None
        for i in range(1, steps):        # no host sync anywhere in this loop
            loss, terms = step(batches[i % 4])
            sched.step()
            losses.append(loss.clone())
        torch.cuda.synchronize()
        table = terms.host()
        step.close()
        return [float(v) for v in losses], [p.detach().clone() for p in model.parameters()], \
            table, step.replays

    l_e, w_e = eager()
    l_g, w_g, table, replays = graphed()
    l_e2, w_e2 = eager()
    return {'losses_equal': l_e == l_g, 'weights_equal': all(torch.equal(a, b) for a, b in zip(w_e, w_g)),
            'eager_reproducible': l_e == l_e2 and all(torch.equal(a, b) for a, b in zip(w_e, w_e2)),
            'replays': replays, 'loss_first': l_e[0], 'loss_last': l_e[-1], 'graph_last': l_g[-1],
            'terms_finite': bool(np.isfinite(np.array(table)).all()),
            'max_weight_diff': max(float((a - b).abs().max()) for a, b in zip(w_e, w_g)),
            'executor': info}


def scenario_bind():
    """Two captured steps BOUND to two resident batches (no staging copies),
    sharing model and optimizer, replayed alternately by the step executor,
    against the eager loop over the same batches."""
    from dvs_of_training_framework_amd.capture import CapturedTrainStep
    from dvs_of_training_framework_amd.loss import unit_backward
    from dvs_of_training_framework_amd.timer import FakeTimer
    from dvs_of_training_framework_amd.training import process_minibatch
    B, H, W, steps = 2, 64, 64, 10
    batches = [synthetic.to_torch(unique_pixel_batch(170 + i, B, H, W, 4096 - 500 * i), 'cuda')
               for i in range(2)]

    def eager():
        model, opt, sched, init_losses = make()
        ev = init_losses((H, W), B, model, 'cuda', sequence_length=1)
        losses = []
        for i in range(steps):
            opt.zero_grad(set_to_none=True)
            loss, _, _ = process_minibatch(model, batches[i % 2], FakeTimer(), 'cuda', True, ev,
                                           [0.5, 1, 1])
            unit_backward(loss)
            model.strict = False
            opt.step()
            sched.step()
            losses.append(loss.detach().clone())
        torch.cuda.synchronize()
        return [float(v) for v in losses], [p.detach().clone() for p in model.parameters()]

    def bound():
        model, opt, sched, init_losses = make()
        ev = init_losses((H, W), B, model, 'cuda', sequence_length=1)
        caps, losses = {}, []
        for i in range(steps):
            k = i % 2
            if k not in caps:
                caps[k] = CapturedTrainStep(model, ev, opt, [0.5, 1, 1], 'cuda', batches[k], bind=True)
                losses.append(caps[k].first_loss.clone())
            else:
                losses.append(caps[k]()[0].clone())
            sched.step()
        torch.cuda.synchronize()
        for c in caps.values():
            c.close()
        return [float(v) for v in losses], [p.detach().clone() for p in model.parameters()]
    l_e, w_e = eager()
    l_b, w_b = bound()
    return {'losses_equal': l_e == l_b, 'weights_equal': all(torch.equal(a, b) for a, b in zip(w_e, w_b)),
            'eager': l_e, 'bound': l_b}


def scenario_loop():
    from dvs_of_training_framework_amd.timer import FakeTimer
    from dvs_of_training_framework_amd.training import train
    B, H, W = 2, 64, 64
    data = [unique_pixel_batch(200 + i, B, H, W, 4096 if i % 3 else 3000) for i in range(6)]
    data[4] = unique_pixel_batch(204, 1, H, W, 4096)        # another signature: runs eagerly

    def run(capture):
        model, opt, sched, init_losses = make()
        ev = init_losses((H, W), B, model, 'cuda', sequence_length=1)
        rows = []

        class Log:
            def add_scalar(self, t, v, x):
                rows.append((t, float(v), x))
        train(model, 'cuda', (synthetic.to_torch(b) for b in data), opt, 6, sched, Log(), ev,
              timers=FakeTimer(), capture=capture, max_events_per_batch=10 ** 7)
        torch.cuda.synchronize()
        return rows, [p.detach().clone() for p in model.parameters()]
    r0, w0 = run(False)
    r1, w1 = run(True)
    return {'rows_equal': r0 == r1, 'n_rows': len(r0),
            'weights_equal': all(torch.equal(a, b) for a, b in zip(w0, w1)),
            'first_diff': next((a, b) for a, b in zip(r0, r1) if a != b) if r0 != r1 else None}


def scenario_infer():
    from dvs_of_training_framework_amd.of import OpticalFlow
    H = W = 64
    rng = np.random.default_rng(3)

    def events(n):
        return [(rng.integers(0, W, n), rng.integers(0, H, n), np.sort(rng.random(n) * 0.04),
                 rng.integers(0, 2, n) * 2 - 1) for _ in range(2)]
    torch.manual_seed(4)
    eager = OpticalFlow((H, W), model=None, event_representation_depth=5)
    graph = OpticalFlow((H, W), model=None, graph=True, event_representation_depth=5)
    graph.load_state_dict(eager._net.state_dict())
    evs = [events(n) for n in (6000, 5000, 6000)]
    graph(evs[0], [0.0, 0.0], [0.04, 0.04])                 # capture
    with torch.no_grad():
        for i in range(30):                                  # replays back to back, unread
            ev, ts, sidx = graph._collate(evs[i % 3], [0.0, 0.0], [0.04, 0.04])
            flow = graph._replay(ev, ts, sidx, 2)
    got = graph._postprocess(flow, False)
    want = eager(evs[29 % 3], [0.0, 0.0], [0.04, 0.04])
    return {'max_diff': float(np.abs(got - want).max()), 'peak': float(np.abs(want).max()),
            'graphs': len(graph._graphs)}


def _dist():
    """1-rank RCCL group when the test asks for it (DVSOF_FORCE_DIST=1), or the
    loopback communicator (DVSOF_LOOPBACK="world:delay_us": average with
    world - 1 all-zero buckets, delay_us late -- NOT an identity, so a kernel
    on the wrong side of a collective changes the weights)."""
    import os
    from dvs_of_training_framework_amd import parallel
    if os.environ.get('DVSOF_LOOPBACK'):
        parallel.claim_streams('cuda')
        red = parallel.GradReducer()
        assert red.active() and red.comm_info()['loopback']
        return red
    if os.environ.get('DVSOF_FORCE_DIST') != '1':
        return None
    parallel.init_distributed('cuda')
    red = parallel.GradReducer()
    assert red.active() and red.comm_info()['ranks'] == 1
    return red


def scenario_big(dtype):
    from dvs_of_training_framework_amd.capture import CapturedTrainStep
    from dvs_of_training_framework_amd.loss import unit_backward
    from dvs_of_training_framework_amd.timer import FakeTimer
    from dvs_of_training_framework_amd.training import process_minibatch
    import time
    B, H, W, steps = 8, 256, 256, 4
    # "<dtype>:fused": the replayed leg updates every gradient bucket inside the backward
    # (optim.fuse_into_backward; behind the bucket's exchange mark under the reducer), the
    # eager leg in optimizer.step() -- same arithmetic, so still bit-identical
    fused = dtype.endswith(':fused')
    dtype = dtype.split(':')[0]
    red = _dist()
    batches = [synthetic.to_torch(synthetic.make_batch(900 + i, B, H, W, None), 'cuda')
               for i in range(2)]

    def eager():
        model, opt, sched, init_losses = make(dtype=dtype)
        model.predictor.reducer = red
        ev = init_losses((H, W), B, model, 'cuda', sequence_length=1)
        losses = []
        for i in range(steps):
            opt.zero_grad(set_to_none=True)
            loss, _, _ = process_minibatch(model, batches[i % 2], FakeTimer(), 'cuda', True, ev,
                                           [0.5, 1, 1])
            unit_backward(loss)
            model.strict = False
            if red is not None:
                red.wait()
            opt.step()
            sched.step()
            losses.append(loss.detach().clone())
        torch.cuda.synchronize()
        return [float(v) for v in losses], [p.detach().clone() for p in model.parameters()]

    info = {}

    def replayed():
        model, opt, sched, init_losses = make(dtype=dtype)
        model.predictor.reducer = red
        if fused:
            opt.fuse_into_backward(model.predictor)
        ev = init_losses((H, W), B, model, 'cuda', sequence_length=1)
        caps, losses = {}, []
        host = 0.0
        i, timed = -1, 0
        while timed < 4:
            i += 1
            k = i % 2
            if k not in caps:
                caps[k] = CapturedTrainStep(model, ev, opt, [0.5, 1, 1], 'cuda', batches[k],
                                            bind=True, reducer=red)
                loss = caps[k].first_loss
            else:
                # steady state: calibrated, and the executors have settled on a lane plan
                # (the trial steps wait for the device)
                steady = i >= steps and all(c.executor.plan()[1] for c in caps.values())
                t0 = time.perf_counter()
                loss = caps[k]()[0]
                if steady:
                    host += time.perf_counter() - t0
                    timed += 1
            sched.step()
            if i < steps:
                losses.append(loss.clone())
            if i == steps - 1:
                torch.cuda.synchronize()
                weights = [p.detach().clone() for p in model.parameters()]
        torch.cuda.synchronize()
        x = caps[0].executor
        info.update(kernels=x.kernels, lanes=x.lanes, lane_kernels=x.lane_kernels, marks=x.marks,
                    host_ms_per_step=round(host / 4 * 1e3, 3),
                    exchange_audit=getattr(caps[0], 'exchange_audit', None))
        for c in caps.values():
            c.close()
        return [float(v) for v in losses], weights
    calls = []
    l_e, w_e = eager()
    calls.append(red.comm_info()['calls'] if red is not None else 0)
    l_r, w_r = replayed()
    calls.append(red.comm_info()['calls'] - calls[0] if red is not None else 0)
    ci = red.comm_info() if red is not None else None
    if red is not None:
        red.close()
    # the exchange does something: under the loopback communicator the weights differ from a
    # run without any exchange (else "bit-identical" would say nothing about ordering)
    moved = None
    if ci and ci['loopback']:
        red = None
        _, w_plain = eager()
        moved = not all(torch.equal(a, b) for a, b in zip(w_e, w_plain))
    return {'losses_equal': l_e == l_r, 'weights_equal': all(torch.equal(a, b) for a, b in zip(w_e, w_r)),
            'eager': l_e, 'replayed': l_r, 'executor': info, 'dist': ci is not None,
            'comm': ci, 'calls': calls, 'exchange_changes_weights': moved,
            'max_weight_diff': max(float((a - b).abs().max()) for a, b in zip(w_e, w_r))}


def _train_rows(data, capture, accum=1, red=None, steps=None, patch=None, compact=False, feed=False):
    from dvs_of_training_framework_amd import capture as cap_mod
    from dvs_of_training_framework_amd.timer import FakeTimer
    from dvs_of_training_framework_amd.training import train
    B, H, W = data[0]['size'], data[0]['images'].shape[-2], data[0]['images'].shape[-1]
    model, opt, sched, init_losses = make()
    model.predictor.reducer = red
    ev = init_losses((H, W), B, model, 'cuda', sequence_length=1)
    rows = []

    class Log:
        def add_scalar(self, t, v, x):
            rows.append((t, float(v), x))
    undo = patch(cap_mod) if patch and capture else None
    info = {}
    orig_close = cap_mod.CapturedLoop.close

    def spy(self):
        info.setdefault('roles', sorted(self.steps))
        info['recaptures'] = self.recaptures
        info['failed'] = str(self.failed) if self.failed else None
        info['replays'] = info.get('replays', 0) + sum(
            s.replays for s in list(self.steps.values()) + list(self.bound.values()))
        return orig_close(self)
    cap_mod.CapturedLoop.close = spy
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            loader = (synthetic.to_torch(b) for b in data)
            if feed:
                from dvs_of_training_framework_amd.feed import DeviceFeeder

                def pinned_every_other(it):     # pinned sources go as they are, pageable ones
                    for i, b in enumerate(it):  # through the slot's staging buffers
                        if i % 2:
                            b = {k: ({c: t.pin_memory() for c, t in v.items()} if isinstance(v, dict)
                                     else v.pin_memory() if torch.is_tensor(v) else v) for k, v in b.items()}
                        yield b
                loader = DeviceFeeder(pinned_every_other(loader), 'cuda')
                info['feeder'] = loader
            train(model, 'cuda', loader, opt,
                  steps or len(data) // accum, sched, Log(), ev, timers=FakeTimer(), capture=capture,
                  max_events_per_batch=10 ** 7, accumulation_steps=accum, reducer=red)
    finally:
        cap_mod.CapturedLoop.close = orig_close
        if undo:
            undo()
    torch.cuda.synchronize()
    if 'feeder' in info:
        f = info.pop('feeder')
        info['fed_batches'], info['fed_bytes'] = f.batches, f.bytes_moved
    return rows, [p.detach().clone() for p in model.parameters()], info


def scenario_feed(accum):
    """train() fed by feed.DeviceFeeder (copy stream, two slots, captured steps
    bound to the slots) against the plain loop with .to(device) per batch."""
    B, H, W = 2, 64, 64
    counts = [2100, 4000, 3000, 4096, 2500, 4096, 7000, 2200]    # 7000: the slots grow
    data = [synthetic.make_batch(800 + i, B, H, W, c) for i, c in enumerate(counts)]
    r0, w0, _ = _train_rows(data, False, accum=accum)
    r1, w1, i1 = _train_rows(data, False, accum=accum, feed=True)
    r2, w2, i2 = _train_rows(data, True, accum=accum, feed=True)
    same = lambda a, b: all(torch.equal(u, v) for u, v in zip(a, b))  # noqa: E731
    return {'n_rows': len(r0), 'eager_feed_equal': r0 == r1 and same(w0, w1),
            'capture_feed_equal': r0 == r2 and same(w0, w2), 'eager': i1, 'capture': i2}


def _compare(data, **kw):
    r0, w0, _ = _train_rows(data, False, **kw)
    r1, w1, info = _train_rows(data, True, **kw)
    return {'rows_equal': r0 == r1, 'n_rows': len(r0),
            'weights_equal': all(torch.equal(a, b) for a, b in zip(w0, w1)), 'info': info,
            'first_diff': next((a, b) for a, b in zip(r0, r1) if a != b) if r0 != r1 else None}


def scenario_accum(dp=False):
    B, H, W = 2, 64, 64
    red = _dist() if dp else None
    data = [unique_pixel_batch(300 + i, B, H, W, 4096 if i % 4 else 3100) for i in range(12)]
    data[7] = unique_pixel_batch(307, 1, H, W, 4000)     # another signature, as a 'middle'
    out = _compare(data, accum=3, red=red)
    if red is not None:
        red.close()
    return out


def scenario_compact():
    from dvs_of_training_framework_amd import encoding
    B, H, W = 2, 64, 64
    data = []
    for i in range(6):
        b = unique_pixel_batch(400 + i, B, H, W, 4096 if i % 2 else 3300)
        ev = b['events']
        counts = np.bincount(ev['sample_index'], minlength=B)
        b['events'] = {'x': ev['x'].astype(np.int16), 'y': ev['y'].astype(np.int16),
                       'timestamp': ev['timestamp'].astype(np.float32),
                       'polarity': (ev['polarity'] > 0),
                       'sample_event_offsets': np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)}
        data.append(b)
    return _compare(data)


def scenario_grow(dp=False):
    """dp: under the reducer.  A rank whose batch outgrows its event capacity
    re-records (an eager micro-batch + a calibration step) while its peers go
    on replaying -- a rank-local decision.  What keeps the ranks' collectives
    paired is that EVERY mode issues the same bucket all-reduces on the same
    communicator: the call count per optimizer step is the same for the eager
    loop and for the capture loop with its two re-recordings."""
    B, H, W = 2, 64, 64
    # events per sample; capacities 8192 -> 16384 -> 32768 (all on the tiled
    # voxeliser, which is bitwise reproducible whatever the event order)
    counts = [2100, 4000, 6000, 5000, 2100, 12000]
    data = [synthetic.make_batch(500 + i, B, H, W, c) for i, c in enumerate(counts)]
    red = _dist() if dp else None
    if red is None:
        return _compare(data)
    r0, w0, _ = _train_rows(data, False, red=red)
    c0 = red.comm_info()
    r1, w1, info = _train_rows(data, True, red=red)
    c1 = red.comm_info()
    red.close()
    return {'rows_equal': r0 == r1, 'n_rows': len(r0),
            'weights_equal': all(torch.equal(a, b) for a, b in zip(w0, w1)), 'info': info,
            'calls_eager': c0['calls'], 'calls_capture': c1['calls'] - c0['calls'],
            'elements_eager': c0['elements'], 'elements_capture': c1['elements'] - c0['elements'],
            'steps': len(data)}


def scenario_fail():
    B, H, W = 2, 64, 64
    data = [unique_pixel_batch(600 + i, B, H, W, 4096) for i in range(5)]

    def patch(cap_mod):
        orig = cap_mod.CapturedTrainStep._record

        def boom(self, executor):
            raise RuntimeError('injected: dvsof_exec_create refused the graph')
        cap_mod.CapturedTrainStep._record = boom
        return lambda: setattr(cap_mod.CapturedTrainStep, '_record', orig)
    return _compare(data, patch=patch)


def scenario_audit():
    from dvs_of_training_framework_amd.capture import CapturedTrainStep
    B, H, W = 2, 64, 64
    model, opt, sched, init_losses = make()
    ev = init_losses((H, W), B, model, 'cuda', sequence_length=1)
    batch = synthetic.to_torch(unique_pixel_batch(700, B, H, W, 4096), 'cuda')
    step = CapturedTrainStep(model, ev, opt, [0.5, 1, 1], 'cuda', batch, event_capacity=8192)
    ok = step.audit()
    # the bug class: drop an eager object the graph points at from the keep list
    step._keep = [k for k in step._keep if k is not step.static]
    static_ptrs = step.static
    step.static = None
    bad = step.audit()
    step.static = static_ptrs
    step.close()
    return {'ok': ok, 'bad': {k: v for k, v in bad.items() if k != 'unheld'},
            'bad_unheld': len(bad['unheld'])}


def _shutdown():
    """Orderly end of a child that joined a 1-rank group.  (The aborts these children showed
    now and then -- SIGABRT, "operation not permitted when stream is capturing" from the
    process group's watchdog thread -- came from stream capture in the default GLOBAL error
    mode: capture.py records in thread_local mode now.)"""
    import torch.distributed as dist
    if dist.is_initialized():
        torch.cuda.synchronize()
        dist.destroy_process_group()


if __name__ == '__main__':
    name, _, arg = sys.argv[1].partition(':')
    out = {'train': scenario_train, 'train_graph': lambda: scenario_train(False),
           'loop': scenario_loop, 'bind': scenario_bind, 'infer': scenario_infer,
           'big': lambda: scenario_big(arg or 'f32'), 'accum': lambda: scenario_accum(arg == 'dp'),
           'compact': scenario_compact, 'grow': lambda: scenario_grow(arg == 'dp'), 'feed': lambda: scenario_feed(int(arg or 1)), 'fail': scenario_fail,
           'audit': scenario_audit}[name]()
    print(json.dumps(out), flush=True)
    _shutdown()
