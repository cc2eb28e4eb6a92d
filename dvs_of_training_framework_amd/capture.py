"""One training step as ONE hipGraph launch.

The reference's loop body (utils/training.py:138-167: batch to device, model,
loss, ``backward()``, ``optimizer.step()``) is ~130 kernel launches here; at
batch 8 the host needs ~2 ms to enqueue what the GPU runs in ~3 ms.
``CapturedTrainStep`` records that body once -- voxelise, predictor forward,
fused loss, the two-stream backward, AdamW -- for a fixed batch signature
(batch size, frame size, event capacity) and replays it:

  * inputs live in static device buffers; a batch is copied in, its events
    padded to the capacity with x = y = -1 (the voxeliser drops them);
  * the graph holds KERNELS ONLY: no memset / memcpy nodes (the voxeliser's
    control words clean up after themselves, the loss needs no zero-fill), all
    scratch comes from the graph's private pool, gradient buckets and
    optimizer tables are the persistent ones the eager warm-up step made;
  * what changes per step is read from device memory: the scheduled learning
    rate and Adam's bias corrections (``FusedAdamW.advance`` refreshes a
    3-float table before each replay), so LambdaLR keeps working.

Results are bit-identical to the eager step (same kernels, same order, same
arguments; tests/test_gpu_capture.py).

``executor=True`` (default): the capture is replayed by the step executor of
``csrc/exec.hip`` (``StepExecutor``) instead of ``hipGraphLaunch``: the kernel
nodes of the captured graph as plain launches from one C call, on the main
stream and the predictor's weight-gradient stream with an event per
dependency between them -- the eager schedule's two-stream overlap at the host
cost of a bare launch per kernel.  Not captured: gradient accumulation
over micro-batches and the data-parallel exchange (``train`` keeps those
eager).
"""
import ctypes

import torch

from . import _lib
from .loss import unit_backward
from .timer import FakeTimer
from .training import TermReadback, process_minibatch

EVENT_KEYS = ('x', 'y', 'timestamp', 'polarity', 'element_index',
              'sample_index')


def _pow2_at_least(n, floor=4096):
    return max(floor, 1 << max(int(n) - 1, 1).bit_length())


class StepExecutor:
    """A captured graph (``torch.cuda.CUDAGraph(keep_graph=True)``) replayed
    as plain kernel launches: ``dvsof_exec_*`` (csrc/exec.hip).  ``side`` are
    the torch streams of lanes 1..; lane 0 is the current stream of a call."""

    def __init__(self, graph, side=()):
        self._graph = graph            # owns the nodes the executor reads
        self._side = list(side)
        raw = graph.raw_cuda_graph()
        arr = (ctypes.c_void_p * max(len(self._side), 1))(
            *[s.cuda_stream for s in self._side])
        handle = ctypes.c_void_p()
        _lib.check(_lib.lib().dvsof_exec_create(
            ctypes.c_void_p(int(raw)), arr, len(self._side),
            ctypes.byref(handle)), 'dvsof_exec_create')
        self._handle = handle
        self._launch = _lib.lib().dvsof_exec_launch
        self.calibrated = False
        self._info()

    def _info(self):
        n = [ctypes.c_int() for _ in range(4)]
        lanes = (ctypes.c_int * 8)()
        _lib.check(_lib.lib().dvsof_exec_info(
            self._handle, *[ctypes.byref(v) for v in n], lanes, 8), 'dvsof_exec_info')
        self.kernels, self.lanes, self.events, self.waits = (v.value for v in n)
        self.lane_kernels = list(lanes)[:self.lanes]

    def replay(self):
        """One step.  The FIRST call runs it on the current stream alone, timed
        per kernel, and waits for it (``dvsof_exec_calibrate``): the lanes are
        then planned from measured durations; later calls never synchronise."""
        if not self.calibrated:
            _lib.check(_lib.lib().dvsof_exec_calibrate(self._handle, _lib.stream()),
                       'dvsof_exec_calibrate')
            self.calibrated = True
            self._info()
            return
        rc = self._launch(self._handle, _lib.stream())
        if rc:
            _lib.check(rc, 'dvsof_exec_launch')

    def nodes(self):
        """[(lane, measured us, cross-lane waits, kernel name)] in launch order."""
        out, lane, us, nw = [], ctypes.c_int(), ctypes.c_float(), ctypes.c_int()
        buf = ctypes.create_string_buffer(256)
        i = 0
        while _lib.lib().dvsof_exec_node(self._handle, i, ctypes.byref(lane), ctypes.byref(us),
                                         ctypes.byref(nw), buf, 256) == 0:
            out.append((lane.value, us.value, nw.value, buf.value.decode(errors='replace')))
            i += 1
        return out

    def close(self):
        if self._handle is not None:
            _lib.lib().dvsof_exec_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CapturedTrainStep:
    def __init__(self, model, evaluator, optimizer, weights, device,
                 example_batch, event_capacity=None, executor=True, bind=False):
        """example_batch: a batch of the signature to capture (wire-format
        events; its tensors may live on the host).  The constructor runs ONE
        eager, validated step on it (host-side assertions, lazy allocations,
        optimizer state and tables) -- a real optimizer step -- then captures.

        bind=True: the device tensors of ``example_batch`` ARE the step's input
        buffers (no staging copy; the event capacity is their length).  A
        loader that fills ``step.static`` in place -- or a resident batch, as
        in bench.py -- then replays with ``step()``; several captured steps
        (one per resident buffer set) may share a model and an optimizer."""
        assert hasattr(optimizer, 'begin_capture'), \
            'the captured step needs optim.FusedAdamW (device-resident lr table)'
        self.model, self.evaluator, self.optimizer = model, evaluator, optimizer
        self.weights, self.device = list(weights), torch.device(device)
        if getattr(optimizer, '_use_dyn', False):
            optimizer.end_capture()      # another captured step shares the optimizer
        ev = example_batch['events']
        n = ev['x'].numel()
        dev = self.device
        self.bound = example_batch if bind else None
        if bind:
            assert all(t.is_cuda for t in (ev['x'], example_batch['images'])), \
                'bind=True needs a device-resident batch'
            self.capacity = n
            self.static = {'events': {k: ev[k] for k in EVENT_KEYS if k in ev},
                           'timestamps': example_batch['timestamps'],
                           'sample_idx': example_batch['sample_idx'],
                           'images': example_batch['images'],
                           'size': int(example_batch['size'])}
        else:
            self.capacity = event_capacity or _pow2_at_least(n)
            assert n <= self.capacity
            self.static = {
                'events': {k: torch.zeros(self.capacity, dtype=ev[k].dtype, device=dev)
                           for k in EVENT_KEYS if k in ev},
                'timestamps': torch.zeros_like(example_batch['timestamps'], device=dev),
                'sample_idx': torch.zeros_like(example_batch['sample_idx'], device=dev),
                'images': torch.zeros_like(example_batch['images'], device=dev),
                'size': int(example_batch['size']),
            }
        self.signature = self._signature(example_batch)
        if not bind:
            self._load(example_batch)
        # eager step: validates the layout, allocates buckets / state / tables
        model.train()
        optimizer.zero_grad(set_to_none=True)
        loss, terms, tags = process_minibatch(
            model, self.static, FakeTimer(), dev, True, evaluator, self.weights)
        unit_backward(loss)
        optimizer.step()
        if hasattr(model, 'strict'):
            # the index vectors of the validated layout are built now, eagerly
            # (inside the capture they would become graph-owned ATen launches)
            model.strict = False
            model._select(self.static['timestamps'], self.static['sample_idx'],
                          self.static['size'])
        self.tags = list(tags)
        self.first_loss, self.first_terms = loss.detach().clone(), terms
        terms.host()
        torch.cuda.synchronize(dev)
        # Eager objects the graph's kernels point at must outlive it whatever
        # the caches that made them do later: index vectors of the layout,
        # voxeliser workspace, gradient buckets, optimizer tables.
        from . import voxel
        self._keep = [dict(getattr(model, '_layout_cache', {})),
                      list(voxel._WORKSPACES.values()),
                      list(getattr(model.predictor, '_bucket_flat', [])),
                      dict(optimizer._tables)]
        # p.grad = None: the recorded backward WRITES the gradient buckets (and
        # re-attaches them as .grad, same pointers as the optimizer's tables);
        # with .grad set it would record the accumulate-into-.grad path
        optimizer.zero_grad(set_to_none=True)
        optimizer.begin_capture(dev)
        self.graph = torch.cuda.CUDAGraph(keep_graph=True) if executor \
            else torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            loss, terms, _ = process_minibatch(
                model, self.static, FakeTimer(), dev, True, evaluator,
                self.weights)
            unit_backward(loss)
            optimizer.step()
        self.loss = loss.detach()
        self._terms = terms._terms      # _Terms: .packed is the [3,K] tensor
        self._keep.append(optimizer._dyn)
        self.replays = 0
        self.executor = None
        if executor:
            pred = model.predictor
            side = [pred._wgrad_stream(dev)] + list(pred._extra_streams(dev))
            self.executor = StepExecutor(self.graph, [s for s in side if s is not None])

    @staticmethod
    def _signature(batch):
        return (int(batch['size']), tuple(batch['images'].shape),
                tuple(batch['timestamps'].shape))

    def fits(self, batch):
        return self._signature(batch) == self.signature and \
            batch['events']['x'].numel() <= self.capacity

    def _load(self, batch):
        ev, st = batch['events'], self.static['events']
        n = ev['x'].numel()
        for k, buf in st.items():
            buf[:n].copy_(ev[k], non_blocking=True)
        if n < self.capacity:
            st['x'][n:] = -1
            st['y'][n:] = -1
            st['sample_index'][n:] = 0
        for k in ('timestamps', 'sample_idx', 'images'):
            self.static[k].copy_(batch[k], non_blocking=True)

    def __call__(self, batch=None):
        """-> (loss, terms): 0-dim tensor and a TermReadback, both reading the
        graph's static outputs (valid until the next call).  ``batch=None`` (or
        the bound batch itself): the input buffers already hold the batch."""
        if batch is not None and batch is not self.bound and batch is not self.static:
            assert self.fits(batch), 'batch does not match the captured signature'
            self._load(batch)
        if not getattr(self.optimizer, '_use_dyn', False):
            self.optimizer.begin_capture(self.device)   # a sibling step was built since
        self.optimizer.advance()
        (self.executor or self.graph).replay()
        self.replays += 1
        return self.loss, TermReadback(self._terms)

    def eager_step(self, batch, timers=None):
        """A batch that does not fit the captured signature: the same step,
        eagerly, on the same persistent gradient buckets."""
        opt = self.optimizer
        opt.end_capture()
        opt.zero_grad(set_to_none=True)
        loss, terms, tags = process_minibatch(
            self.model, batch, timers or FakeTimer(), self.device, True,
            self.evaluator, self.weights)
        unit_backward(loss)
        opt.step()          # p.grad stays: the graph's tables point at the buckets
        opt.begin_capture(self.device)
        return loss, terms, tags

    def close(self):
        self.optimizer.end_capture()
        if self.executor is not None:
            self.executor.close()
