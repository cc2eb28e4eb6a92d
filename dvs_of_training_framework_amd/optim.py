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


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=1e-2, amsgrad=False):
        defaults = dict(lr=lr, betas=betas, eps=eps,
                        weight_decay=weight_decay, amsgrad=amsgrad)
        super().__init__(params, defaults)
        self._tables = {}

    def _state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st['step'] = 0
            st['exp_avg'] = torch.zeros_like(p)
            st['exp_avg_sq'] = torch.zeros_like(p)
            st['max_exp_avg_sq'] = torch.zeros_like(p)
        return st

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
            for q in (p, p.grad, st['exp_avg'], st['exp_avg_sq'],
                      st['max_exp_avg_sq']):
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

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group['params'] if p.grad is not None]
            if not plist:
                continue
            _lib.require_cuda(*plist)
            for p in plist:
                assert _dense(p) and not p.grad.is_sparse
            steps = set()
            for p in plist:
                st = self._state(p)
                st['step'] = int(st['step']) + 1
                steps.add(st['step'])
            assert len(steps) == 1, 'tensors of one group step together'
            t_ptrs, t_sizes, t_chunks, n = self._table(gi, plist)
            b1, b2 = group['betas']
            _lib.check(_lib.lib().dvsof_adamw_step(
                t_ptrs.data_ptr(), t_sizes.data_ptr(), t_chunks.data_ptr(), n,
                float(group['lr']), float(b1), float(b2), float(group['eps']),
                float(group['weight_decay']), steps.pop(),
                1 if group['amsgrad'] else 0, _lib.stream()),
                'dvsof_adamw_step')
        return loss


def _dense(t):
    """Dense, non-overlapping storage (any permutation of strides)."""
    return t.is_contiguous() or \
        t.is_contiguous(memory_format=torch.channels_last) or \
        t.numel() == t.untyped_storage().nbytes() // t.element_size()
