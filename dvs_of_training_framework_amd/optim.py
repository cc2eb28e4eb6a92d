"""Fused optimizers on the HIP path (csrc/optim.hip).

``FusedAdamW`` is what ``construct_optimizer`` builds for ``--optimizer ADAM``
(reference train_flownet.py:57-75: ``torch.optim.AdamW(amsgrad=True)``).  It
keeps torch's state-dict schema (step, exp_avg, exp_avg_sq, max_exp_avg_sq)
so checkpoints interchange with the reference's (utils/serializer.py:60-110).
One kernel launch per parameter group updates every tensor of the group.
"""
import ctypes

import torch

from . import _lib

_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
_lib.register('dvsof_adamw_chunk_elems', _i, [])
_lib.register('dvsof_adamw_step', _i, [_vp, _vp, _vp, _i, _f, _f, _f, _f, _f,
                                       _i, _i, _vp])
_lib.register('dvsof_adamw_dynamic', None, [_f, _f, _f, _i, ctypes.POINTER(_f)])
_lib.register('dvsof_adamw_set_dynamic', _i, [_vp, ctypes.POINTER(_f), _i, _vp])
_lib.register('dvsof_adamw_step_dyn', _i, [_vp, _vp, _vp, _i, _vp, _f, _f, _f,
                                           _f, _i, _vp])
_lib.register('dvsof_radam_step', _i, [_vp, _vp, _vp, _i, _f, _f, _f, _f, _f,
                                       _i, _f, _i, _i, _f, _vp])
_lib.register('dvsof_grad_centralize', _i, [_vp, _i, _i, _vp])


class _FusedBase(torch.optim.Optimizer):
    """Shared machinery: per-group device tables of {param, grad, state...}
    pointers, element counts and (tensor, chunk) work items."""
    STATE = ()          # names of the 3 state tensors after param and grad

    def __init__(self, params, defaults):
        super().__init__(params, defaults)
        self._tables = {}

    def _init_state(self, p, st):
        for name in self.STATE:
            st[name] = torch.zeros_like(p)

    def _state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st['step'] = 0
            self._init_state(p, st)
        return st

    def load_state_dict(self, state_dict):
        """torch's loader keeps the SAVED strides of state tensors; a state
        dict written by ``torch.optim.AdamW`` (the reference's optimizer, or a
        CPU run) holds contiguous moments while the conv weights here are
        channels_last.  The kernels index parameter, gradient and state with
        one offset, so state is re-laid to its parameter's strides."""
        super().load_state_dict(state_dict)
        for group in self.param_groups:
            for p in group['params']:
                st = self.state.get(p)
                if not st:
                    continue
                shared = {}     # RAdam aliases slow_buffer to exp_avg
                for name in self.STATE:
                    v = st.get(name)
                    if not torch.is_tensor(v):
                        continue
                    if id(v) in shared:
                        st[name] = shared[id(v)]
                        continue
                    if v.stride() != p.stride() or v.dtype != p.dtype or \
                            v.device != p.device:
                        st[name] = torch.empty_like(p).copy_(v)
                    shared[id(v)] = st[name]
                if torch.is_tensor(st.get('step')):
                    st['step'] = int(st['step'])
        self._tables = {}

    def _table(self, gi, plist):
        """Device tables for group gi; rebuilt only when a pointer moved."""
        key = tuple((p.data_ptr(), p.grad.data_ptr()) for p in plist)
        cached = self._tables.get(gi)
        if cached is not None and cached[0] == key:
            return cached[1:]
        chunk = _lib.lib().dvsof_adamw_chunk_elems()
        ptrs, sizes, chunks = [], [], []
        for t, p in enumerate(plist):
            st = self._state(p)
            for q in (p, p.grad) + tuple(st[n] for n in self.STATE):
                assert q.dtype == torch.float32 and q.is_cuda
                assert q.stride() == p.stride(), \
                    'parameter, gradient and state must share one layout'
                ptrs.append(q.data_ptr())
            sizes.append(p.numel())
            chunks += [(t, c) for c in range((p.numel() + chunk - 1) // chunk)]
        dev = plist[0].device
        # uint64 pointers travel as int64 bit patterns
        t_ptrs = torch.tensor([x - (1 << 64) if x >= (1 << 63) else x
                               for x in ptrs], dtype=torch.int64, device=dev)
        t_sizes = torch.tensor(sizes, dtype=torch.int64, device=dev)
        t_chunks = torch.tensor(chunks, dtype=torch.int32, device=dev)
        self._tables[gi] = (key, t_ptrs, t_sizes, t_chunks, len(chunks))
        return self._tables[gi][1:]

    def _launch(self, group, tables, step, plist):
        raise NotImplementedError

    def _step_params(self, key, group, plist):
        _lib.require_cuda(*plist)
        for p in plist:
            assert _dense(p) and not p.grad.is_sparse
        steps = set()
        frozen = getattr(self, '_use_dyn', False)   # captured step: advance() counts
        for p in plist:
            st = self._state(p)
            if not frozen:
                st['step'] = int(st['step']) + 1
            steps.add(st['step'])
        assert len(steps) == 1, 'tensors of one group step together'
        self._launch(group, self._table(key, plist), max(steps.pop(), 1), plist)

    # ---- update fused into the backward ------------------------------------
    def fuse_into_backward(self, predictor, flush_at=None):
        """Update the gradient buckets of ``predictor`` as soon as their
        gradients are final (after the all-reduce under data parallelism)
        instead of in ``step()``: the HBM-bound update then runs beside the
        MFMA-bound backward kernels.  Same arithmetic, same result; ``step()``
        still has to be called and updates whatever is left.  Set
        ``fused_active = False`` on micro-batches that only accumulate.

        flush_at: bucket indices at which everything collected so far is
        updated in ONE launch (None: one launch per bucket).  ``(5,)`` updates
        the decoder and residual parameters (82 % of the bytes) when the last
        residual weight gradient is done, beside the encoder's backward, and
        leaves the encoder buckets to ``step()``."""
        self.fused_active = True
        self._flush_at = None if flush_at is None else set(flush_at)
        self._pending = []
        self._done = set()
        self._group_of = {id(p): (gi, g) for gi, g in enumerate(self.param_groups)
                          for p in g['params']}
        predictor.bucket_hook = self._on_bucket

    @torch.no_grad()
    def _on_bucket(self, b, params):
        if not getattr(self, 'fused_active', False):
            return
        if self._flush_at is not None:      # collect, update at the flush points
            self._pending += params
            if b not in self._flush_at:
                return
            params, self._pending = self._pending, []
        by_group = {}
        for p in params:
            if id(p) in self._group_of and p.grad is not None:
                gi, g = self._group_of[id(p)]
                by_group.setdefault(gi, (g, []))[1].append(p)
        for gi, (g, plist) in by_group.items():
            self._step_params((gi, b), g, plist)
            self._done.update(id(p) for p in plist)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        done = getattr(self, '_done', None)
        if getattr(self, '_pending', None):
            self._pending = []          # never flushed this step: step() takes them
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group['params'] if p.grad is not None and
                     not (done and id(p) in done)]
            if not plist:
                continue
            self._step_params(gi if not done else (gi, 'rest'), group, plist)
        if done:
            done.clear()
        return loss


class FusedAdamW(_FusedBase):
    STATE = ('exp_avg', 'exp_avg_sq', 'max_exp_avg_sq')

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=1e-2, amsgrad=False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps,
                                      weight_decay=weight_decay,
                                      amsgrad=amsgrad))

    def _launch(self, group, tables, step, plist):
        t_ptrs, t_sizes, t_chunks, n = tables
        b1, b2 = group['betas']
        if getattr(self, '_use_dyn', False):     # captured step: lr and bias corrections from the device table
            gi = next(i for i, g in enumerate(self.param_groups) if g is group)
            _lib.check(_lib.lib().dvsof_adamw_step_dyn(
                t_ptrs.data_ptr(), t_sizes.data_ptr(), t_chunks.data_ptr(), n,
                self._dyn[gi].data_ptr(), float(b1), float(b2),
                float(group['eps']), float(group['weight_decay']),
                1 if group['amsgrad'] else 0, _lib.stream()),
                'dvsof_adamw_step_dyn')
            return
        _lib.check(_lib.lib().dvsof_adamw_step(
            t_ptrs.data_ptr(), t_sizes.data_ptr(), t_chunks.data_ptr(), n,
            float(group['lr']), float(b1), float(b2), float(group['eps']),
            float(group['weight_decay']), step,
            1 if group['amsgrad'] else 0, _lib.stream()), 'dvsof_adamw_step')

    # ---- a step captured in a hipGraph (capture.CapturedTrainStep) ---------
    def begin_capture(self, device):
        """From now on ``step()`` enqueues the dyn-table kernel and leaves the
        step counters alone (capturing enqueues nothing; ``advance`` counts).
        The table is allocated once: a captured graph keeps its address."""
        if getattr(self, '_dyn', None) is None:
            ng = len(self.param_groups)
            self._dyn = torch.zeros(ng, 4, dtype=torch.float32, device=device)
        self._use_dyn = True

    def advance(self):
        """Before every replay: count the step and refresh {lr, lr/bc1,
        sqrt(bc2)} of every group (one tiny kernel carrying the values as
        arguments: dvsof_adamw_set_dynamic)."""
        ng = len(self.param_groups)
        buf, rows = (ctypes.c_float * 3)(), (ctypes.c_float * (4 * ng))()
        for gi, group in enumerate(self.param_groups):
            steps = set()
            for p in group['params']:
                st = self._state(p)
                st['step'] = int(st['step']) + 1
                steps.add(st['step'])
            assert len(steps) == 1
            b1, b2 = group['betas']
            _lib.lib().dvsof_adamw_dynamic(float(group['lr']), float(b1),
                                           float(b2), steps.pop(), buf)
            rows[4 * gi], rows[4 * gi + 1], rows[4 * gi + 2] = buf[0], buf[1], buf[2]
        _lib.check(_lib.lib().dvsof_adamw_set_dynamic(
            self._dyn.data_ptr(), rows, 4 * ng, _lib.stream()), 'dvsof_adamw_set_dynamic')

    def end_capture(self):
        """Back to eager steps (the table stays: a graph may still use it)."""
        self._use_dyn = False


class FusedRAdam(_FusedBase):
    """Rectified Adam (Liu et al., ICLR 2020) with the defaults of the
    ``RAdam.radam.RAdam`` class the reference builds for ``--optimizer RADAM``
    (train_flownet.py:62-64; upstream submodule absent: parity unpinned,
    oracle/ref_optim.py restates the published algorithm)."""
    STATE = ('exp_avg', 'exp_avg_sq', 'slow_buffer')

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0, degenerated_to_sgd=True):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps,
                                      weight_decay=weight_decay,
                                      degenerated_to_sgd=degenerated_to_sgd))

    def _init_state(self, p, st):
        st['exp_avg'] = torch.zeros_like(p)
        st['exp_avg_sq'] = torch.zeros_like(p)
        st['slow_buffer'] = st['exp_avg']      # unused by the kernel

    def _launch(self, group, tables, step, plist):
        t_ptrs, t_sizes, t_chunks, n = tables
        b1, b2 = group['betas']
        _lib.check(_lib.lib().dvsof_radam_step(
            t_ptrs.data_ptr(), t_sizes.data_ptr(), t_chunks.data_ptr(), n,
            float(group['lr']), float(b1), float(b2), float(group['eps']),
            float(group['weight_decay']), step, 5.0,
            2 | (1 if group['degenerated_to_sgd'] else 0), 0, 0.0,
            _lib.stream()), 'dvsof_radam_step')


class FusedRanger(_FusedBase):
    """Ranger = RAdam + Lookahead (k, alpha) + gradient centralisation, with
    the defaults of lessw2020's ``ranger.Ranger`` which the reference builds
    for its DEFAULT ``--optimizer RANGER`` (train_flownet.py:65-71,
    utils/options.py:254-257; upstream submodule absent: parity unpinned)."""
    STATE = ('exp_avg', 'exp_avg_sq', 'slow_buffer')

    def __init__(self, params, lr=1e-3, alpha=0.5, k=6, N_sma_threshhold=5,
                 betas=(.95, 0.999), eps=1e-5, weight_decay=0, use_gc=True,
                 gc_conv_only=False):
        super().__init__(params, dict(
            lr=lr, alpha=alpha, k=k, N_sma_threshhold=N_sma_threshhold,
            betas=betas, eps=eps, weight_decay=weight_decay, use_gc=use_gc,
            gc_conv_only=gc_conv_only))

    def _init_state(self, p, st):
        st['exp_avg'] = torch.zeros_like(p)
        st['exp_avg_sq'] = torch.zeros_like(p)
        st['slow_buffer'] = p.detach().clone()

    def _launch(self, group, tables, step, plist):
        lib = _lib.lib()
        if group['use_gc']:
            gc_dim = 3 if group['gc_conv_only'] else 1
            for p in plist:
                if p.dim() > gc_dim:     # mean over everything but dim 0
                    assert p.grad.stride() == p.stride()
                    _lib.check(lib.dvsof_grad_centralize(
                        p.grad.data_ptr(), p.shape[0], p.numel() // p.shape[0],
                        _lib.stream()), 'dvsof_grad_centralize')
        t_ptrs, t_sizes, t_chunks, n = tables
        b1, b2 = group['betas']
        _lib.check(lib.dvsof_radam_step(
            t_ptrs.data_ptr(), t_sizes.data_ptr(), t_chunks.data_ptr(), n,
            float(group['lr']), float(b1), float(b2), float(group['eps']),
            float(group['weight_decay']), step,
            float(group['N_sma_threshhold']), 1,
            1 if step % group['k'] == 0 else 0, float(group['alpha']),
            _lib.stream()), 'dvsof_radam_step')


def _dense(t):
    """Dense, non-overlapping storage (any permutation of strides)."""
    return t.is_contiguous() or \
        t.is_contiguous(memory_format=torch.channels_last) or \
        t.numel() == t.untyped_storage().nbytes() // t.element_size()
