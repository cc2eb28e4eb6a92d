"""Data-parallel gradient exchange: one process per GPU, RCCL over xGMI.

The reference is single-process (no torch.distributed
use at all); this is the new capability behind the same train loop.  The
predictor's explicit backward hands over gradient BUCKETS (flat buffers, in
the order the backward produces them: fine decoder stages first, encoder
last) as soon as the last wgrad of a bucket has been enqueued; each bucket is
all-reduced (average) on a side HIP stream gated by an event, so the exchange
overlaps the rest of the backward.  ``wait()`` makes the compute stream wait
for the outstanding collectives right before the optimizer step.
"""
import contextlib
import os
import sys

import torch
import torch.distributed as dist


@contextlib.contextmanager
def _stdout_to_stderr():
    """RCCL prints a five-line version banner on STDOUT when rank 0 creates its
    first communicator; a caller that parses this process's stdout (one JSON
    line from bench.py) must not see it.  File-descriptor level: the banner
    comes from C code."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


_EXCHANGE_STREAMS = {}      # device index -> the stream of the collectives
_UPDATE_STREAMS = {}        # device index -> the stream of the per-bucket optimizer updates


def claim_streams(device):
    """Create AND USE the three streams of a training step -- the current
    stream, the backward's weight-gradient stream, the exchange stream --
    before any RCCL communicator exists.  ROCclr binds a stream to one of
    GPU_MAX_HW_QUEUES hardware queues when it is first used (least-loaded
    queue first); a communicator brings internal streams of its own, and a
    stream first used after two communicators existed was seen sharing ONE
    hardware queue with the weight-gradient stream (rocprofv3 kernel trace,
    profiles/round3): a hardware queue is in order, so the exchange stream's
    wait for "bucket ready" then holds back every weight-gradient kernel
    queued behind it -- 4.8-4.95 instead of 3.25 ms per step at batch 8.
    -> (weight-gradient stream or None, exchange stream)."""
    from .predictor import Predictor
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    side = Predictor._wgrad_stream(None, dev)
    if key not in _EXCHANGE_STREAMS:
        # highest priority the runtime offers (-1 here): a collective's kernels should not queue
        # for CU slots behind the matrix kernels of the two compute lanes, which fill every CU
        # (one workgroup with ~all of its LDS each) -- small kernels of the other lane were seen
        # waiting 20-70 us for a slot (profiles/round4/timeline.txt)
        _EXCHANGE_STREAMS[key] = torch.cuda.Stream(device=dev, priority=-1)
        _UPDATE_STREAMS[key] = torch.cuda.Stream(device=dev)
        for s in (torch.cuda.current_stream(dev), side, _EXCHANGE_STREAMS[key], _UPDATE_STREAMS[key]):
            if s is not None:
                with torch.cuda.stream(s):
                    torch.zeros(64, device=dev).add_(1.0)     # a kernel: the queue is bound now
        torch.cuda.synchronize(dev)
    return side, _EXCHANGE_STREAMS[key]


def init_distributed(device_type='cuda'):
    """Initialise from the torchrun environment (RANK, LOCAL_RANK, WORLD_SIZE,
    MASTER_ADDR, MASTER_PORT).  -> (rank, local_rank, world_size)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    force = os.environ.get('DVSOF_FORCE_DIST') == '1'    # 1-rank group: exercises the RCCL path
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        backend = 'nccl' if device_type == 'cuda' else 'gloo'
        if device_type == 'cuda':
            torch.cuda.set_device(local)
            if os.environ.get('DVSOF_NO_STREAM_CLAIM') != '1':
                claim_streams(torch.device('cuda', local))     # before the communicators' streams
            with _stdout_to_stderr():
                dist.init_process_group(backend, rank=rank, world_size=world,
                                        device_id=torch.device('cuda', local))
                # communicator creation (and its banner) happens here at the latest
                dist.all_reduce(torch.zeros(1, device=torch.device('cuda', local)))
                torch.cuda.synchronize()
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


class GradReducer:
    """Averages gradient buckets across ranks, overlapped with backward.

    ONE communicator carries the exchange on every path of a run (round-3
    advisor finding: captured steps exchanged on the C ABI's communicator, the
    eager micro-batches around them -- re-recordings, other signatures, the
    fall-back after a failed capture -- on torch.distributed's, and those
    decisions are rank-local, so collectives of two communicators interleaved
    differently on different ranks):

    direct=True  (default on a GPU with an 'nccl' process group): the C ABI's
        own RCCL communicator (``dvsof_comm_create`` from a unique id rank 0
        shares through the process group), created HERE -- a collective call,
        at a point every rank passes -- and used by the eager loop
        (``dvsof_allreduce_bucket`` on the exchange stream) and by the step
        executor's marks alike.  Every closing micro-batch issues the same 8
        bucket all-reduces in the same order whatever launch mode its rank is
        in, which is all RCCL needs.
    direct=False  ``torch.distributed.all_reduce`` on the process group (gloo on
        the CPU tests; ``DVSOF_DIRECT_RCCL=0`` on a GPU for comparison runs --
        a captured step then switches the reducer to direct, see
        ``adopt_direct``).
    loopback=(world, delay_us)  or DVSOF_LOOPBACK="world:delay_us": the loopback
        communicator of ``dvsof_comm_create_loopback`` -- no peers, the average
        with world - 1 all-zero buckets, delay_us late: the ordering tests of
        the exchange on one GPU."""

    def __init__(self, group=None, direct=None, loopback=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.enabled = True          # False on non-boundary micro-batches
        self.pending = []
        self._side = None
        self.bytes_reduced = 0
        self._comm = None
        self._events = {}            # bucket address -> its "ready" event, reused every step
        self._marked = 0             # marks left in the capture in progress
        if loopback is None and os.environ.get('DVSOF_LOOPBACK'):
            w, _, us = os.environ['DVSOF_LOOPBACK'].partition(':')
            loopback = (int(w), int(us or 50))
        self.loopback = tuple(loopback) if loopback else None
        if direct is None:
            direct = self.loopback is not None or (
                dist.is_initialized() and dist.get_backend(group) == 'nccl'
                and os.environ.get('DVSOF_DIRECT_RCCL', '1') != '0')
        self._direct = bool(direct)
        if self._direct:
            self._comm = self._make_comm()

    def comm_handle(self):
        """The C ABI's communicator.  Made in the constructor of a direct
        reducer; on first use otherwise -- a COLLECTIVE call then: every rank
        must get here at the same point (capture.py calls ``adopt_direct``
        before it runs or records anything)."""
        if self._comm is None:
            self._comm = self._make_comm()
        return self._comm

    def adopt_direct(self):
        """From now on the eager path exchanges on the C ABI's communicator too
        (called by the captured step / loop before its first micro-batch, so
        that replays and the eager micro-batches around them share one
        communicator).  Collective when the communicator does not exist yet."""
        if not self._direct:
            assert not self.pending, 'adopt_direct between steps only'
            self.comm_handle()
            self._direct = True

    def comm_info(self):
        """{'ranks', 'loopback', 'calls', 'elements'} of the C ABI's
        communicator (ranks = ncclCommCount), or None without one."""
        if self._comm is None:
            return None
        import ctypes
        from . import _lib
        r, lb = ctypes.c_int(), ctypes.c_int()
        c, e = ctypes.c_ulonglong(), ctypes.c_ulonglong()
        _lib.check(_lib.lib().dvsof_comm_info(self._comm, ctypes.byref(r), ctypes.byref(lb),
                                              ctypes.byref(c), ctypes.byref(e)), 'dvsof_comm_info')
        return {'ranks': r.value, 'loopback': bool(lb.value), 'calls': c.value,
                'elements': e.value}

    def exchange_stream(self, device):
        """Stream of the collectives.  DVSOF_EXCHANGE_ON_WGRAD_STREAM=1: the
        backward's second stream (no third stream, the collectives queue up
        between the weight-gradient kernels) -- a measurement switch."""
        if os.environ.get('DVSOF_EXCHANGE_ON_WGRAD_STREAM') == '1':
            from .predictor import _SIDE_STREAMS
            key = torch.device(device).index
            key = torch.cuda.current_device() if key is None else key
            if key in _SIDE_STREAMS:
                return _SIDE_STREAMS[key]
        return self._side_stream(device)

    def update_stream(self, device):
        """Stream of the update lane (dvsof_exec_set_update_stream), claimed with the others;
        None: the updates stay on the exchange stream (DVSOF_UPDATE_ON_XSTREAM=1, or the streams
        were not claimed)."""
        if os.environ.get('DVSOF_UPDATE_ON_XSTREAM') == '1':
            return None
        key = torch.device(device).index
        key = torch.cuda.current_device() if key is None else key
        return _UPDATE_STREAMS.get(key)

    def _make_comm(self):
        import ctypes
        from . import _lib
        lib = _lib.lib()
        comm = ctypes.c_void_p()
        if self.loopback is not None:
            _lib.check(lib.dvsof_comm_create_loopback(ctypes.byref(comm), *self.loopback),
                       'dvsof_comm_create_loopback')
            return comm
        rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        ident = ctypes.create_string_buffer(128)
        if rank == 0:
            _lib.check(lib.dvsof_comm_unique_id(ident), 'dvsof_comm_unique_id')
        if self.world > 1:
            box = [ident.raw]
            dist.broadcast_object_list(box, src=0, group=self.group)
            ident = ctypes.create_string_buffer(box[0], 128)
        with _stdout_to_stderr():        # RCCL's banner, as in init_distributed
            _lib.check(lib.dvsof_comm_create(ctypes.byref(comm), self.world, rank,
                                             ident), 'dvsof_comm_create')
        return comm

    def close(self):
        if self._comm is not None:
            from . import _lib
            torch.cuda.synchronize()
            _lib.lib().dvsof_comm_destroy(self._comm)
            self._comm = None

    def _side_stream(self, device):
        if self._side is None:
            if os.environ.get('DVSOF_NO_STREAM_CLAIM') == '1':
                self._side = torch.cuda.Stream(device=device)
            else:
                self._side = claim_streams(device)[1]
        return self._side

    def active(self):
        """Does bucket_ready exchange anything?"""
        return self.enabled and (self.world > 1 or self.loopback is not None or
                                 os.environ.get('DVSOF_FORCE_DIST') == '1')

    def bucket_ready(self, flat, after=None):
        """Average ``flat`` across ranks (asynchronously).  ``after``: callable
        run once the average is enqueued -- on the exchange stream, behind the
        collective (the fused optimizer update of this bucket)."""
        if not self.active():
            if after is not None:
                after()
            return
        if flat.is_cuda and torch.cuda.is_current_stream_capturing():
            # a captured step (capture.py): leave a MARK where the collective
            # goes; the step executor issues it on every replay (csrc/exec.hip)
            from . import _lib
            _lib.check(_lib.lib().dvsof_exec_mark(
                1, self._marked, flat.data_ptr(), flat.numel(), _lib.stream()),
                'dvsof_exec_mark')
            if after is not None and os.environ.get('DVSOF_UPDATE_ON_LANE') == '1':
                # (before round 4's second half: this bucket's update behind a WAIT mark on the
                # CURRENT stream -- the executor makes that lane wait for the collective there:
                # 3.18 against 2.65 ms per step under the loopback exchange, the lane stalls)
                _lib.check(_lib.lib().dvsof_exec_mark(
                    3, self._marked, flat.data_ptr(), flat.numel(), _lib.stream()),
                    'dvsof_exec_mark')
                self._marked += 1
                after()
                return
            if after is not None:
                # optim.fuse_into_backward: this bucket's update is captured ON THE EXCHANGE
                # STREAM, behind a WAIT mark there (the branch joins the capture again in
                # wait()).  The executor recognises kernels that follow nothing but WAIT marks
                # and launches them on the exchange stream behind the collective: the update
                # overlaps the rest of the backward and no compute lane waits for anything
                # before the JOIN mark (csrc/exec.hip, XNode.xlane).
                cur = torch.cuda.current_stream(flat.device)
                side = self._side_stream(flat.device)
                ev = torch.cuda.Event()
                ev.record(cur)
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    _lib.check(_lib.lib().dvsof_exec_mark(
                        3, self._marked, flat.data_ptr(), flat.numel(), _lib.stream()),
                        'dvsof_exec_mark')
                    after()
                self._capture_branch = side
            self._marked += 1
            return
        self.bytes_reduced += flat.numel() * flat.element_size()
        if flat.is_cuda and self._comm is not None and self._direct:
            from . import _lib
            ready = self._ready_event(flat)
            ready.record()
            side = self._side_stream(flat.device)
            with torch.cuda.stream(side):
                side.wait_event(ready)
                _lib.check(_lib.lib().dvsof_allreduce_bucket(
                    self._comm, flat.data_ptr(), flat.numel(), _lib.stream()),
                    'dvsof_allreduce_bucket')
                if after is not None:    # stream-ordered behind the collective
                    after()
            self._keep = getattr(self, '_keep', [])
            self._keep.append(flat)
            self._touched = True
        elif flat.is_cuda:
            ready = self._ready_event(flat)
            ready.record()
            side = self._side_stream(flat.device)
            with torch.cuda.stream(side):
                side.wait_event(ready)
                # (one rank: SUM is the same average, and RCCL then launches
                # no scaled-copy kernel -- see csrc/comm.hip)
                work = dist.all_reduce(flat, op=dist.ReduceOp.AVG if self.world > 1
                                       else dist.ReduceOp.SUM,
                                       group=self.group, async_op=True)
                if after is not None:
                    work.wait()          # the exchange stream waits for the collective
                    after()
            self.pending.append((work, flat, False))
            self._touched = True
        else:   # gloo (CPU tests): SUM then scale
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM,
                                   group=self.group, async_op=True)
            self.pending.append((work, flat, True, after))

    def _ready_event(self, flat):
        """One event per bucket, recorded again every step (a bucket's
        previous exchange has been joined before its gradients are rewritten)."""
        ev = self._events.get(flat.data_ptr())
        if ev is None:
            ev = self._events[flat.data_ptr()] = torch.cuda.Event()
        return ev

    def wait(self):
        if self._marked and torch.cuda.is_current_stream_capturing():
            from . import _lib      # the optimizer kernels come behind the exchange
            branch, self._capture_branch = getattr(self, '_capture_branch', None), None
            if branch is not None:      # the updates captured on the exchange stream join here
                torch.cuda.current_stream().wait_stream(branch)
            _lib.check(_lib.lib().dvsof_exec_mark(2, 0, None, 0, _lib.stream()),
                       'dvsof_exec_mark')
            self._marked = 0
            return
        for item in self.pending:
            work, flat, scale = item[:3]
            work.wait()          # NCCL: the current stream waits, not the host
            if scale:
                flat.div_(self.world)
                if len(item) > 3 and item[3] is not None:
                    item[3]()
        self.pending = []
        self._keep = []
        if getattr(self, '_touched', False) and self._side is not None:
            # updates enqueued behind the collectives on the exchange stream
            torch.cuda.current_stream().wait_stream(self._side)
            self._touched = False


def broadcast_parameters(module, src=0, group=None):
    """Replicas start from rank ``src``'s weights."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def all_reduce_scalars(t, group=None):
    """Average a small tensor of logged scalars across ranks (optional)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t.div_(dist.get_world_size(group))
    return t
