"""One training micro-batch as ONE launch call.

The reference's loop body (utils/training.py:138-167: batch to device, model,
loss, ``backward()``, [``optimizer.step()``]) is ~120 kernel launches here; at
batch 8 the host needs ~2 ms to enqueue what the GPU runs in ~3 ms.
``CapturedTrainStep`` records that body once -- voxelise, predictor forward,
fused loss, the two-stream backward, [gradient exchange, AdamW] -- for a fixed
batch signature (batch size, frame size, event capacity) and replays it:

  * inputs live in static device buffers; a batch is copied in, its events
    padded to the capacity with x = y = -1 (the voxeliser drops them).  Wire
    columns (44 B/event) or the compact encoded columns (9 B/event +
    ``sample_event_offsets``, voxel.is_compact) -- whatever the example batch
    carries;
  * the graph holds KERNELS ONLY: no memset / memcpy nodes (the voxeliser's
    control words clean up after themselves, the loss needs no zero-fill), all
    scratch comes from the graph's private pool, gradient buckets and
    optimizer tables are the persistent ones the eager warm-up step made;
  * what changes per step is read from device memory: the scheduled learning
    rate and Adam's bias corrections (``FusedAdamW.advance`` refreshes a
    3-float table before each replay), so LambdaLR keeps working.

ROLES -- gradient accumulation (utils/options.py:318-325: ``bs // mbs``
micro-batches per optimizer step, utils/training.py:156-167).  A micro-batch
is recorded in the role it plays in its optimizer step:

    full    the only micro-batch: write gradients, exchange, update
    first   write the gradient buckets, nothing else
    middle  accumulate into them
    last    accumulate, exchange, update

Every role is its own capture of the same Python body the eager loop runs
(the loss scaled by 1/accumulation_steps exactly as there), over the same
static input buffers (``share=``).

DATA PARALLELISM.  With a ``reducer`` (parallel.GradReducer) the gradient
exchange belongs to the step: while capturing, the reducer leaves MARKS in the
stream where the eager loop issues a bucket's all-reduce / joins the exchange
stream; the step executor turns them into ``ncclAllReduce`` calls on the
exchange stream between its kernel launches (csrc/exec.hip).  One C call per
step under data parallelism too.

Results are bit-identical to the eager step (same kernels, same order, same
arguments; tests/test_gpu_capture.py).

``executor=True`` (default): the capture is replayed by the step executor of
``csrc/exec.hip`` (``StepExecutor``) instead of ``hipGraphLaunch``: the kernel
nodes of the captured graph as plain launches from one C call, on the main
stream and the predictor's weight-gradient stream with an event per
dependency between them -- the eager schedule's two-stream overlap at the host
cost of a bare launch per kernel.
"""
import ctypes

import torch

from . import _lib, voxel
from .loss import unit_backward
from .timer import FakeTimer
from .training import TermReadback, _timed, process_minibatch

EVENT_KEYS = ('x', 'y', 'timestamp', 'polarity', 'element_index',
              'sample_index')
ROLES = ('full', 'first', 'middle', 'last')


def role_of(micro_in_step, accumulation_steps):
    """Role of micro-batch ``micro_in_step`` (0-based) of an optimizer step."""
    if accumulation_steps == 1:
        return 'full'
    if micro_in_step == 0:
        return 'first'
    return 'last' if micro_in_step == accumulation_steps - 1 else 'middle'


def _pow2_at_least(n, floor=4096):
    return max(floor, 1 << max(int(n) - 1, 1).bit_length())


class CaptureFailed(RuntimeError):
    """Recording failed AFTER the constructor's eager micro-batch ran: that
    micro-batch is done (``loss``, ``terms``, ``tags`` are its results); the
    caller goes on eagerly."""

    def __init__(self, cause, loss, terms, tags):
        super().__init__(f'{type(cause).__name__}: {cause}')
        self.loss, self.terms, self.tags = loss, terms, tags


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
        self._xstream = None
        self._info()

    def _info(self):
        n = [ctypes.c_int() for _ in range(4)]
        lanes = (ctypes.c_int * 8)()
        _lib.check(_lib.lib().dvsof_exec_info(
            self._handle, *[ctypes.byref(v) for v in n], lanes, 8), 'dvsof_exec_info')
        self.kernels, self.lanes, self.events, self.waits = (v.value for v in n)
        self.lane_kernels = list(lanes)[:self.lanes]
        m = ctypes.c_int()
        _lib.check(_lib.lib().dvsof_exec_marks(self._handle, ctypes.byref(m)),
                   'dvsof_exec_marks')
        self.marks = m.value

    def set_comm(self, comm, exchange_stream):
        """RCCL communicator (dvsof_comm_create handle) and exchange stream
        the marks of the capture are acted on with; before the first replay."""
        self._xstream = exchange_stream       # keep the torch stream alive
        _lib.check(_lib.lib().dvsof_exec_set_comm(
            self._handle, comm, ctypes.c_void_p(exchange_stream.cuda_stream)),
            'dvsof_exec_set_comm')
        self._info()        # (updates captured behind WAIT marks now have a lane of their own)

    def set_update_stream(self, stream):
        self._ustream = stream
        _lib.check(_lib.lib().dvsof_exec_set_update_stream(
            self._handle, ctypes.c_void_p(stream.cuda_stream)), 'dvsof_exec_set_update_stream')
        self._info()

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

    def plan(self):
        """(name of the lane plan in effect, settled?): after calibration the
        executor tries its plans on the next steps and keeps the fastest."""
        buf, settled = ctypes.create_string_buffer(32), ctypes.c_int()
        _lib.check(_lib.lib().dvsof_exec_plan(self._handle, buf, 32, ctypes.byref(settled)),
                   'dvsof_exec_plan')
        return buf.value.decode(), bool(settled.value)

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

    def node_arg(self, i, arg, nbytes):
        """First ``nbytes`` of kernel argument ``arg`` of node ``i``."""
        buf = ctypes.create_string_buffer(nbytes)
        _lib.check(_lib.lib().dvsof_exec_node_arg(self._handle, i, arg, nbytes, buf),
                   'dvsof_exec_node_arg')
        return buf.raw

    def close(self):
        if self._handle is not None:
            _lib.lib().dvsof_exec_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _event_layout(events):
    """Columns of the example batch that are staged, and the ones that are
    per-event (padded to the capacity)."""
    if voxel.is_compact(events):
        per_event = tuple(k for k in voxel.COMPACT_KEYS
                          if k != 'sample_event_offsets' and k in events)
        return per_event + ('sample_event_offsets',), per_event
    per_event = tuple(k for k in EVENT_KEYS if k in events)
    return per_event, per_event


class CapturedTrainStep:
    def __init__(self, model, evaluator, optimizer, weights, device,
                 example_batch, event_capacity=None, executor=True, bind=False,
                 reducer=None, role='full', accumulation_steps=1, share=None):
        """example_batch: a batch of the signature to capture (wire-format or
        compact events; its tensors may live on the host).  The constructor
        runs ONE eager, validated micro-batch on it in the given role (host-
        side assertions, lazy allocations, optimizer state and tables) -- a
        real one: gradients and, in the roles that close a step, the update --
        then captures.

        bind=True: the device tensors of ``example_batch`` ARE the step's input
        buffers (no staging copy; the event capacity is their length).  A
        loader that fills ``step.static`` in place -- or a resident batch, as
        in bench.py -- then replays with ``step()``; several captured steps
        (one per resident buffer set) may share a model and an optimizer.

        reducer:  parallel.GradReducer -- the data-parallel exchange is part of
                  the step (needs executor=True)
        role, accumulation_steps:  see the module docstring
        share:    another CapturedTrainStep of the same signature whose input
                  buffers this one reads (the roles of one loop)"""
        assert hasattr(optimizer, 'begin_capture'), \
            'the captured step needs optim.FusedAdamW (device-resident lr table)'
        assert role in ROLES and (role == 'full') == (accumulation_steps == 1)
        # (optim.fuse_into_backward: the bucket updates are kernels of the capture like any
        # other; under data parallelism each follows its bucket's exchange mark and the
        # executor makes it wait for that collective)
        self.model, self.evaluator, self.optimizer = model, evaluator, optimizer
        self.weights, self.device = list(weights), torch.device(device)
        self.reducer, self.role, self.accum = reducer, role, int(accumulation_steps)
        self.closes = role in ('full', 'last')
        exchanging = reducer is not None and self.closes and self._would_exchange()
        assert executor or not exchanging, \
            'the gradient exchange of a captured step is issued by the step executor'
        if reducer is not None and self._would_exchange():
            # one communicator for replays AND the eager micro-batches around them (this
            # constructor's own, re-recordings, other signatures, the fall-back after a failed
            # recording); a collective call when the communicator is still to be made, so it
            # happens here -- before anything rank-local can go wrong -- not after the recording
            reducer.adopt_direct()
        if getattr(optimizer, '_use_dyn', False):
            optimizer.end_capture()      # another captured step shares the optimizer
        ev = example_batch['events']
        n = ev['x'].numel()
        dev = self.device
        self.bound = example_batch if bind else None
        self.slot = example_batch.get('slot') if bind else None   # feed.DeviceFeeder buffers
        self.keys, self.per_event = _event_layout(ev)
        if share is not None:
            assert not bind and share.fits(example_batch)
            self.capacity, self.static = share.capacity, share.static
        elif bind:
            assert all(t.is_cuda for t in (ev['x'], example_batch['images'])), \
                'bind=True needs a device-resident batch'
            self.capacity = n
            self.static = {'events': {k: ev[k] for k in self.keys},
                           'timestamps': example_batch['timestamps'],
                           'sample_idx': example_batch['sample_idx'],
                           'images': example_batch['images'],
                           'size': int(example_batch['size'])}
        else:
            self.capacity = event_capacity or _pow2_at_least(n)
            assert n <= self.capacity
            self.static = {
                'events': {k: torch.zeros(self.capacity if k in self.per_event
                                          else ev[k].numel(), dtype=ev[k].dtype, device=dev)
                           for k in self.keys},
                'timestamps': torch.zeros_like(example_batch['timestamps'], device=dev),
                'sample_idx': torch.zeros_like(example_batch['sample_idx'], device=dev),
                'images': torch.zeros_like(example_batch['images'], device=dev),
                'size': int(example_batch['size']),
            }
        self.signature = self._signature(example_batch)
        if not bind:
            self._load(example_batch)
        # eager micro-batch: validates the layout, allocates buckets / state / tables
        model.train()
        loss, terms, tags = self._body(self.static, FakeTimer())
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
        self.graph = self.executor = None
        self.replays = 0
        try:
            self._record(executor)
        except Exception as e:      # noqa: BLE001 -- reported with the eager results
            optimizer.end_capture()
            raise CaptureFailed(e, self.first_loss, self.first_terms, self.tags) from e

    # ------------------------------------------------------------- the body
    def _would_exchange(self):
        was, self.reducer.enabled = self.reducer.enabled, True
        try:
            return self.reducer.active()
        finally:
            self.reducer.enabled = was

    def _body(self, batch, timers):
        """The loop body of utils/training.py:154-167 for one micro-batch in
        this step's role; identical Python for the eager run, the recording
        and a batch that does not fit the captured signature."""
        opt, red = self.optimizer, self.reducer
        if self.role in ('full', 'first'):
            opt.zero_grad(set_to_none=True)     # the backward WRITES the buckets
        else:
            self._attach_grads()                # ... or accumulates into them
        if red is not None:
            red.enabled = self.closes
        loss, terms, tags = process_minibatch(
            self.model, batch, timers, self.device, True, self.evaluator,
            self.weights)
        if self.accum == 1:
            unit_backward(loss)
        else:
            loss /= self.accum
            loss.backward()
        if self.closes:
            if red is not None:
                red.wait()
            opt.step()
        return loss, terms, tags

    def _attach_grads(self):
        """``p.grad`` = the parameter's slice of its gradient bucket wherever
        it is unset (after a replay Python does not know what the graph wrote;
        an eager accumulating micro-batch must find the buckets attached)."""
        self.model.predictor.attach_bucket_grads()

    def _record(self, executor):
        model, optimizer, dev = self.model, self.optimizer, self.device
        # Eager objects the graph's kernels point at must outlive it whatever
        # the caches that made them do later: index vectors of the layout,
        # voxeliser workspace, gradient buckets, optimizer tables.
        self._keep = [dict(getattr(model, '_layout_cache', {})),
                      list(voxel._WORKSPACES.values()),
                      list(getattr(model.predictor, '_bucket_flat', [])),
                      dict(optimizer._tables), self.static]
        optimizer.begin_capture(dev)
        self.graph = torch.cuda.CUDAGraph(keep_graph=True) if executor \
            else torch.cuda.CUDAGraph()
        # thread_local: a process group's watchdog thread polls its work events with
        # hipEventQuery whenever it likes; under the default (global) capture mode such a call
        # from ANOTHER thread invalidates this capture and kills the watchdog ("operation not
        # permitted when stream is capturing" -> SIGABRT, seen in 1 of 3..30 runs of a 1-rank
        # group).  Everything the step itself enqueues comes from this thread.
        with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
            loss, terms, _ = self._body(self.static, FakeTimer())
        self.loss = loss.detach()
        self._terms = terms._terms      # _Terms: .packed is the [3,K] tensor
        self._keep.append(optimizer._dyn)
        if executor:
            pred = model.predictor
            side = [pred._wgrad_stream(dev)] + list(pred._extra_streams(dev))
            self.executor = StepExecutor(self.graph, [s for s in side if s is not None])
            red = self.reducer
            if self.executor.marks:
                assert red is not None
                # kernels the executor lets run beside a bucket's collective must not touch
                # the bucket (csrc/exec.hip, _audit.audit_exchange)
                from ._audit import audit_exchange
                self.exchange_audit = audit_exchange(self)
                if self.exchange_audit['violations']:
                    import sys
                    print('exchange audit:', self.exchange_audit['violations'], file=sys.stderr)
                    raise RuntimeError('a kernel inside the exchange window of a gradient bucket '
                                       f"takes a pointer into it: {self.exchange_audit['violations'][:4]}")
                self.executor.set_comm(red.comm_handle(), red.exchange_stream(dev))
                us = red.update_stream(dev)
                if us is not None:
                    self.executor.set_update_stream(us)

    # ------------------------------------------------------------ the batch
    @staticmethod
    def _signature(batch):
        return (int(batch['size']), tuple(batch['images'].shape),
                tuple(batch['timestamps'].shape),
                voxel.is_compact(batch['events']))

    def same_signature(self, batch):
        return self._signature(batch) == self.signature

    def fits(self, batch):
        return self.same_signature(batch) and \
            batch['events']['x'].numel() <= self.capacity

    def _load(self, batch):
        ev, st = batch['events'], self.static['events']
        n = ev['x'].numel()
        for k in self.keys:
            buf = st[k]
            if k in self.per_event:
                buf[:n].copy_(ev[k], non_blocking=True)
            else:               # sample_event_offsets [B+1]: the last one stays n
                buf.copy_(ev[k], non_blocking=True)
        if n < self.capacity:
            st['x'][n:] = -1
            st['y'][n:] = -1
            if 'sample_index' in st:
                st['sample_index'][n:] = 0
        for k in ('timestamps', 'sample_idx', 'images'):
            self.static[k].copy_(batch[k], non_blocking=True)

    def __call__(self, batch=None):
        """-> (loss, terms): 0-dim tensor and a TermReadback, both reading the
        graph's static outputs (valid until the next call).  ``batch=None`` (or
        the bound batch itself): the input buffers already hold the batch."""
        bound = batch is self.bound or batch is self.static or \
            (self.slot is not None and batch is not None and batch.get('slot') == self.slot)
        if batch is not None and not bound:
            assert self.fits(batch), 'batch does not match the captured signature'
            self._load(batch)
        if not getattr(self.optimizer, '_use_dyn', False):
            self.optimizer.begin_capture(self.device)   # a sibling step was built since
        if self.closes:
            self.optimizer.advance()
        (self.executor or self.graph).replay()
        self.replays += 1
        if not self.closes:
            self._attach_grads()    # Python's view of what the graph just did
        return self.loss, TermReadback(self._terms)

    def eager_step(self, batch, timers=None):
        """A batch that does not fit the captured signature: the same
        micro-batch, eagerly, on the same persistent gradient buckets."""
        opt = self.optimizer
        opt.end_capture()
        out = self._body(batch, timers or FakeTimer())
        opt.begin_capture(self.device)  # p.grad stays: the graph's tables point at the buckets
        return out

    # ------------------------------------------------------------ the audit
    def audit(self):
        """Debug walk over the captured kernel nodes: every pointer a kernel
        argument carries must lie in the graph's private pool or in an object
        this step keeps alive -- the class of bug behind a fault in a replay
        whose eager source tensor had been freed.  -> dict (see _audit.py)."""
        from ._audit import audit_step
        return audit_step(self)

    def close(self):
        self.optimizer.end_capture()
        if self.executor is not None:
            self.executor.close()


class CapturedLoop:
    """What ``training.train(capture=True)`` drives: one CapturedTrainStep per
    role over shared input buffers, created when a role is first met (its
    constructor runs that micro-batch eagerly), re-created with a larger event
    capacity when a batch of the captured signature brings more events, and a
    permanent fall-back to the eager body when recording fails."""

    def __init__(self, model, evaluator, optimizer, weights, device,
                 accumulation_steps=1, reducer=None, executor=True,
                 event_capacity=None):
        self.args = (model, evaluator, optimizer, weights, device)
        self.accum, self.reducer, self.executor = accumulation_steps, reducer, executor
        self.event_capacity = event_capacity
        self.steps = {}         # role -> step over this loop's own input buffers
        self.bound = {}         # (role, slot, generation) -> step bound to a feeder slot
        self.failed = None      # CaptureFailed of the role that could not be recorded
        self.recaptures = 0
        if reducer is not None and torch.device(device).type == 'cuda' and \
                (reducer.world > 1 or reducer.loopback is not None):
            reducer.adopt_direct()      # collective: every rank builds its loop at the same point

    def any(self):
        return next(iter(self.steps.values()), None)

    def _failed(self, e, role):
        import warnings
        warnings.warn(f'captured step ({role}) could not be recorded, '
                      f'training continues eagerly: {e}')
        self.failed = e
        self.close()
        if role in ('full', 'last'):    # the eager loop's zero_grad after its step
            self.args[2].zero_grad(set_to_none=True)
        return e.loss, e.terms, e.tags

    def _run_bound(self, batch, role, timers):
        """A feed.DeviceFeeder batch: its slot's device buffers ARE the input
        buffers of the step (bind=True), one captured step per (role, slot)."""
        index, generation = batch['slot']
        for k in [k for k in self.bound if k[1] == index and k[2] != generation]:
            self.bound.pop(k).close()       # that slot's buffers were re-allocated
        key = (role, index, generation)
        step = self.bound.get(key)
        if step is None and self.failed is None:
            try:
                step = self.bound[key] = CapturedTrainStep(
                    *self.args, batch, executor=self.executor, bind=True,
                    reducer=self.reducer, role=role, accumulation_steps=self.accum)
            except CaptureFailed as e:
                return self._failed(e, role)
            return step.first_loss, step.first_terms, step.tags
        if step is not None:
            with _timed(timers or FakeTimer(), 'forward'):
                loss, terms = step(batch)
            return loss, terms, step.tags
        return self._eager(role, batch, timers)

    def run(self, batch, micro_in_step, timers=None):
        """-> (loss, terms, tags) of this micro-batch."""
        role = role_of(micro_in_step, self.accum)
        if batch.get('slot') is not None:
            return self._run_bound(batch, role, timers)
        step, lead = self.steps.get(role), self.any()
        if lead is not None and lead.same_signature(batch) and not lead.fits(batch):
            # more events than the buffers hold: every role is recorded again
            self.close()
            self.recaptures += 1
            step = lead = None
        if step is None and self.failed is None and \
                (lead is None or lead.fits(batch)):
            n = batch['events']['x'].numel()
            cap = max(self.event_capacity or 0, _pow2_at_least(n))
            try:
                step = self.steps[role] = CapturedTrainStep(
                    *self.args, batch, event_capacity=cap, executor=self.executor,
                    reducer=self.reducer, role=role, accumulation_steps=self.accum,
                    share=lead)
            except CaptureFailed as e:
                return self._failed(e, role)
            return step.first_loss, step.first_terms, step.tags
        if step is not None and step.fits(batch):
            with _timed(timers or FakeTimer(), 'forward'):
                loss, terms = step(batch)
            return loss, terms, step.tags
        return self._eager(role, batch, timers)

    def _eager(self, role, batch, timers):
        """Another signature (or recording is off): the eager body in this
        role, on the same buckets."""
        proto = self.steps.get(role) or next(
            (v for k, v in self.bound.items() if k[0] == role), None)
        if proto is not None:
            return proto.eager_step(batch, timers)
        body = _EagerBody(*self.args, self.reducer, role, self.accum)
        opt = self.args[2]
        was = getattr(opt, '_use_dyn', False)
        if was:
            opt.end_capture()
        out = body._body(batch, timers or FakeTimer())
        if was:
            opt.begin_capture(body.device)
        return out

    def close(self):
        for s in list(self.steps.values()) + list(self.bound.values()):
            s.close()
        self.steps, self.bound = {}, {}


class _EagerBody(CapturedTrainStep):
    """The body of a role without a recording (CapturedLoop fall-back)."""

    def __init__(self, model, evaluator, optimizer, weights, device, reducer,
                 role, accumulation_steps):     # noqa: super().__init__ records
        self.model, self.evaluator, self.optimizer = model, evaluator, optimizer
        self.weights, self.device = list(weights), torch.device(device)
        self.reducer, self.role, self.accum = reducer, role, int(accumulation_steps)
        self.closes = role in ('full', 'last')
        self.executor = None
