"""Thin Python wrappers over the conv-stack entry points of the C ABI
(include/dvsof.h: dvsof_conv2d_*, dvsof_flow_head_*, dvsof_act_bwd,
dvsof_weight_flip_transpose).  Tensors here are raw device buffers with an
explicit layout tag; predictor.py composes them into the network."""
import ctypes

import torch

from . import _lib

_vp, _i, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t

ACT_NONE, ACT_RELU, ACT_MISH = 0, 1, 2
NHWC, NCHW = 0, 1


class Src(ctypes.Structure):
    """dvsof_src_t"""
    _fields_ = [('p', _vp), ('C', _i), ('layout', _i), ('p16', _vp)]


class ConvDesc(ctypes.Structure):
    """dvsof_conv_desc_t"""
    _fields_ = [('src', Src * 3), ('nsrc', _i), ('B', _i), ('H', _i),
                ('W', _i), ('upsample', _i), ('ksize', _i), ('stride', _i),
                ('pad', _i), ('Cout', _i), ('act', _i), ('mfma', _i),
                ('scratch', _vp), ('scratch_bytes', _sz),
                ('winograd_input', _vp), ('y16', _vp), ('w16', _vp),
                ('gout16', _vp), ('flags', _i), ('bias_cls', _vp),
                ('winograd_pre', _vp), ('winograd_next', _vp),
                ('winograd_next_gout', _vp), ('winograd_gout', _vp)]


class GradDst(ctypes.Structure):
    """dvsof_grad_dst_t"""
    _fields_ = [('p', _vp), ('addend', _vp), ('addend2', _vp),
                ('actsrc', _vp), ('p16', _vp), ('head_w', _vp),
                ('head_gflow', _vp), ('head_x', _vp), ('head_part', _vp)]


_P = ctypes.POINTER
_lib.register('dvsof_conv2d_fwd', _i, [_P(ConvDesc), _vp, _vp, _vp, _vp, _vp,
                                       _vp])
_lib.register('dvsof_conv2d_dgrad', _i, [_P(ConvDesc), _vp, _vp, _P(GradDst),
                                         _i, _vp])
_lib.register('dvsof_conv2d_wgrad_workspace_bytes', _sz, [_P(ConvDesc)])
_lib.register('dvsof_conv2d_dgrad_fuses_head', _i, [_P(ConvDesc)])
_lib.register('dvsof_conv2d_dgrad_head_rows', _i, [_P(ConvDesc)])
_lib.register('dvsof_flow_head_reduce', _i, [_vp, _i, _i, _vp, _vp, _vp])
_lib.register('dvsof_conv2d_wgrad', _i, [_P(ConvDesc), _vp, _vp, _vp, _vp,
                                         _sz, _vp])
_lib.register('dvsof_weight_flip_transpose', _i, [_vp, _vp, _i, _i, _i, _vp])
_lib.register('dvsof_flow_head_fwd', _i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i,
                                          _vp])
_lib.register('dvsof_flow_head_bwd_workspace_bytes', _sz, [_i, _i, _i, _i])
_lib.register('dvsof_flow_head_bwd', _i, [_vp, _vp, _vp, _vp, _vp, _i, _vp,
                                          _vp, _vp, _i, _i, _i, _i, _vp, _sz,
                                          _vp, _vp])
_lib.register('dvsof_to_bf16', _i, [_vp, _vp, _sz, _vp])
_lib.register('dvsof_act_bwd', _i, [_vp, _vp, _i, _vp, _sz, _vp])
_lib.register('dvsof_conv2d_tile_id', _i, [_P(ConvDesc), _i])
_lib.register('dvsof_conv2d_kernel_generation', _i, [_P(ConvDesc), _i])
_lib.register('dvsof_conv2d_last_patch', _i, [_i])
_lib.register('dvsof_conv2d_fwd_weight_elems', _sz, [_P(ConvDesc)])
_lib.register('dvsof_conv2d_dgrad_weight_elems', _sz, [_P(ConvDesc)])
_lib.register('dvsof_conv2d_prepare', _i, [_P(ConvDesc), _vp, _vp, _vp, _vp])
_lib.register('dvsof_conv2d_prepare16', _i, [_P(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp])
_lib.register('dvsof_to_bf16_many', _i, [_P(_vp), _P(_vp), _P(ctypes.c_size_t), _i, _vp])
_lib.register('dvsof_conv2d_scratch_bytes', _sz, [_P(ConvDesc)])
_lib.register('dvsof_flow_fold_weights', _i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp])
_lib.register('dvsof_flow_fold_workspace_bytes', _sz, [_i, _i])
_lib.register('dvsof_flow_fold_bias', _i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp])
_lib.register('dvsof_flow_fold_grads', _i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp,
                                            _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp])
WGRAD_SKIP_FLAT = 1
_lib.register('dvsof_conv2d_winograd_tile', _i, [_P(ConvDesc), _i])
_lib.register('dvsof_conv2d_winograd_chain', _i, [_P(ConvDesc), _i])


MFMA_F32, MFMA_BF16, MFMA_BF16X3, MFMA_BF16_TWINS = 0, 1, 2, 3


def twin(t):
    """Uninitialised bf16 twin buffer of an f32 tensor (mode 3: written by
    the producing kernel next to the f32 tensor)."""
    return torch.empty(t.shape, dtype=torch.bfloat16, device=t.device)


def to_bf16(src):
    """bf16 twin of a prepared weight form (dvsof_to_bf16)."""
    dst = torch.empty(src.numel(), dtype=torch.bfloat16, device=src.device)
    _lib.check(_lib.lib().dvsof_to_bf16(src.data_ptr(), dst.data_ptr(),
                                         src.numel(), _lib.stream()),
               'dvsof_to_bf16')
    return dst


def to_bf16_many(tensors):
    """bf16 twins of up to 16 tensors in ONE launch (dvsof_to_bf16_many)."""
    tensors = list(tensors)
    out = [torch.empty(t.numel(), dtype=torch.bfloat16, device=t.device)
           for t in tensors]
    for i in range(0, len(tensors), 16):
        chunk, dst = tensors[i:i + 16], out[i:i + 16]
        n = len(chunk)
        src_p = (_vp * n)(*[t.data_ptr() for t in chunk])
        dst_p = (_vp * n)(*[t.data_ptr() for t in dst])
        sizes = (ctypes.c_size_t * n)(*[t.numel() for t in chunk])
        _lib.check(_lib.lib().dvsof_to_bf16_many(src_p, dst_p, sizes, n,
                                                  _lib.stream()),
                   'dvsof_to_bf16_many')
    return out


UP_NONE, UP_NEAREST, UP_ZERO = 0, 1, 2     # dvsof_conv_desc_t.upsample


def make_desc(srcs, B, H, W, Cout, ksize=3, stride=1, pad=1, upsample=False,
              act=ACT_NONE, mfma=MFMA_F32):
    """srcs: list of (tensor, C, layout[, bf16 twin]).  upsample: False / True
    (2x nearest) or UP_ZERO (2x zero insertion: the layer is a transposed
    convolution with stride 2 -- one NHWC source, 3x3, pad 1)."""
    d = ConvDesc()
    d.nsrc = len(srcs)
    for i, src in enumerate(srcs):
        t, C, layout = src[:3]
        d.src[i].p = t.data_ptr()
        d.src[i].C = C
        d.src[i].layout = layout
        d.src[i].p16 = _lib.ptr(src[3]) if len(src) > 3 else None
    d.B, d.H, d.W = B, H, W
    if upsample is True or upsample is False:
        upsample = UP_NEAREST if upsample else UP_NONE
    assert upsample in (UP_NONE, UP_NEAREST, UP_ZERO)
    d.upsample = upsample
    d.ksize, d.stride, d.pad = ksize, stride, pad
    d.Cout, d.act, d.mfma = Cout, act, mfma
    return d


def out_size(desc):
    up = 2 if desc.upsample else 1
    ho = (desc.H * up + 2 * desc.pad - desc.ksize) // desc.stride + 1
    wo = (desc.W * up + 2 * desc.pad - desc.ksize) // desc.stride + 1
    return ho, wo


_PLAN = {}


def _plan(desc):
    """Shape-only facts of a layer, cached: (raw weight elems, forward-form
    elems, data-gradient-form elems, forward / data-gradient scratch bytes,
    weight-gradient workspace bytes)."""
    key = (desc.nsrc, tuple((desc.src[i].C, desc.src[i].layout)
                            for i in range(desc.nsrc)), desc.B, desc.H, desc.W,
           desc.upsample, desc.ksize, desc.stride, desc.pad, desc.Cout,
           desc.mfma)
    p = _PLAN.get(key)
    if p is None:
        lib, ref = _lib.lib(), ctypes.byref(desc)
        raw = desc.Cout * desc.ksize ** 2 * sum(desc.src[i].C
                                                for i in range(desc.nsrc))
        p = _PLAN[key] = (raw, lib.dvsof_conv2d_fwd_weight_elems(ref),
                          lib.dvsof_conv2d_dgrad_weight_elems(ref),
                          lib.dvsof_conv2d_scratch_bytes(ref),
                          lib.dvsof_conv2d_wgrad_workspace_bytes(ref))
    return p


def _scratch(desc, device):
    """Attach the scratch a Winograd-evaluated layer needs for this call
    (dvsof_conv2d_scratch_bytes; stream-ordered reuse by torch's allocator).
    -> the tensor, to be kept alive until the call is enqueued."""
    n = _plan(desc)[3]
    if n == 0:
        desc.scratch, desc.scratch_bytes = None, 0
        return None
    t = torch.empty(n // 4, dtype=torch.float32, device=device)
    desc.scratch, desc.scratch_bytes = t.data_ptr(), n
    return t


def winograd_chain(desc, kind):
    """Can this layer's forward (kind 0) / data gradient (kind 1) also write
    the Winograd forms of its consumer (``wino_next`` / ``wino_next_gout``)?"""
    return bool(_lib.lib().dvsof_conv2d_winograd_chain(ctypes.byref(desc), kind))


def winograd_tile(desc, kind):
    """0 | 2 | 4: Winograd output tile of the forward (kind 0), data gradient
    (1) or weight gradient (2) of this layer (dvsof_conv2d_winograd_tile)."""
    return _lib.lib().dvsof_conv2d_winograd_tile(ctypes.byref(desc), kind)


def winograd_form(desc, channels, device):
    """Buffer of one F(4x4) form [36][tiles][channels] of this layer's frame."""
    tiles = desc.B * (desc.H // 4) * (desc.W // 4)
    return torch.empty(36 * tiles * channels, dtype=torch.float32, device=device)


def _chain(desc, pre=None, nxt=None, nxt_gout=None, gout=None):
    desc.winograd_pre = _lib.ptr(pre)
    desc.winograd_next = _lib.ptr(nxt)
    desc.winograd_next_gout = _lib.ptr(nxt_gout)
    desc.winograd_gout = _lib.ptr(gout)


def conv_fwd(desc, weight, bias, device, residual=None, want_z=False,
             keep_input_transform=False, weight16=None, bias_cls=None,
             wino_pre=None, wino_next=None):
    """-> y [B,Ho,Wo,Cout] (NHWC buffer), z or None.  In mode 3
    (MFMA_BF16_TWINS) ``desc._y16`` is y's bf16 twin afterwards.
    keep_input_transform: a Winograd layer's scratch (it starts with the
    transformed input) stays attached to ``desc`` for conv_wgrad.
    wino_pre / wino_next: forms shared along a chain of Winograd layers
    (dvsof_conv_desc_t.winograd_pre / winograd_next; winograd_form buffers)."""
    ho, wo = out_size(desc)
    y = torch.empty(desc.B, ho, wo, desc.Cout, dtype=torch.float32,
                    device=device)
    z = torch.empty_like(y) if want_z else None
    desc._y16 = twin(y) if desc.mfma == MFMA_BF16_TWINS else None
    desc.y16 = _lib.ptr(desc._y16)
    desc.w16 = _lib.ptr(weight16)
    desc.bias_cls = _lib.ptr(bias_cls)
    ws = _scratch(desc, device)     # noqa: F841  (alive across the call)
    _chain(desc, pre=wino_pre, nxt=wino_next)
    _lib.check(_lib.lib().dvsof_conv2d_fwd(
        ctypes.byref(desc), weight.data_ptr(), _lib.ptr(bias),
        _lib.ptr(residual), y.data_ptr(), _lib.ptr(z), _lib.stream()),
        'dvsof_conv2d_fwd')
    _chain(desc)
    desc._wino_input = (wino_pre if wino_pre is not None else ws) \
        if keep_input_transform else None
    return y, z


def prepare(desc, weight, want_dgrad, phase_weights=None, want16=False):
    """-> (w_fwd, w_dgrad): prepared weights (dvsof_conv2d_prepare).  w_fwd is
    the raw weight itself unless the layer runs as sub-pixel phases.
    ``phase_weights``: the w_fwd of an earlier call -- only the data-gradient
    form is made (from it, for a sub-pixel layer).
    ``want16``: -> (w_fwd, w_dgrad, w_fwd16, w_dgrad16), the bf16 twins written
    by the kernels that make the forms (dvsof_conv2d_prepare16).  w_fwd16 is
    None when nothing in this call touched the forward form (``phase_weights``
    given, or a raw-weight layer without ``want_dgrad``: convert those with
    ``to_bf16_many``)."""
    lib = _lib.lib()
    raw, nf, ndg, nscratch, _ = _plan(desc)
    dg_only = phase_weights is not None
    if dg_only:
        w_fwd = phase_weights
    else:
        w_fwd = weight if nf == raw else torch.empty(
            nf, dtype=torch.float32, device=weight.device)
    w_dg = torch.empty(ndg, dtype=torch.float32, device=weight.device) \
        if want_dgrad else None
    make_fwd = not dg_only and nf != raw
    w_fwd16 = w_dg16 = None
    if make_fwd or want_dgrad:
        # Winograd layer: both forms come from the raw weights
        wino = nscratch > 0
        # dgrad only: w_fwd = NULL, the data-gradient form comes from the raw weights (the
        # forward form of a sub-pixel layer may be the nine-product one: csrc/fwd_min.hip)
        wp = weight.data_ptr()
        fp = w_fwd.data_ptr() if (nf != raw) else None
        if dg_only:
            fp = None
        if want16 and not wino:
            dev = weight.device
            if make_fwd or (nf == raw and want_dgrad and not dg_only):
                w_fwd16 = torch.empty(nf, dtype=torch.bfloat16, device=dev)
            if want_dgrad:
                w_dg16 = torch.empty(ndg, dtype=torch.bfloat16, device=dev)
            _lib.check(lib.dvsof_conv2d_prepare16(
                ctypes.byref(desc), wp, fp, _lib.ptr(w_dg), _lib.ptr(w_fwd16),
                _lib.ptr(w_dg16), _lib.stream()), 'dvsof_conv2d_prepare16')
        else:
            _lib.check(lib.dvsof_conv2d_prepare(
                ctypes.byref(desc), wp, fp, _lib.ptr(w_dg), _lib.stream()),
                'dvsof_conv2d_prepare')
            if want16:      # Winograd forms (not used by the twins mode)
                w_fwd16 = to_bf16(w_fwd) if make_fwd else None
                w_dg16 = to_bf16(w_dg) if w_dg is not None else None
    if want16:
        return w_fwd, w_dg, w_fwd16, w_dg16
    return w_fwd, w_dg


def flip_transpose(weight, Cout, ksize, Ctot):
    wt = torch.empty(Ctot * ksize * ksize * Cout, dtype=torch.float32,
                     device=weight.device)
    _lib.check(_lib.lib().dvsof_weight_flip_transpose(
        weight.data_ptr(), wt.data_ptr(), Cout, ksize, Ctot, _lib.stream()),
        'dvsof_weight_flip_transpose')
    return wt


def conv_dgrad(desc, weight_t, gout, dsts, bwd_act=ACT_NONE, weight16=None,
               gout16=None, wino_pre=None, wino_next=None, wino_next_gout=None):
    """dsts: list of dict(p=, addend=, addend2=, actsrc=[, p16=]) per source.
    wino_*: forms shared along a chain of Winograd layers (conv_fwd)."""
    arr = (GradDst * len(dsts))()
    for i, d in enumerate(dsts):
        arr[i].p = d['p'].data_ptr()
        arr[i].addend = _lib.ptr(d.get('addend'))
        arr[i].addend2 = _lib.ptr(d.get('addend2'))
        arr[i].actsrc = _lib.ptr(d.get('actsrc'))
        arr[i].p16 = _lib.ptr(d.get('p16'))
        arr[i].head_w = _lib.ptr(d.get('head_w'))
        arr[i].head_gflow = _lib.ptr(d.get('head_gflow'))
        arr[i].head_x = _lib.ptr(d.get('head_x'))
        arr[i].head_part = _lib.ptr(d.get('head_part'))
    desc.w16 = _lib.ptr(weight16)
    desc.gout16 = _lib.ptr(gout16)
    ws = _scratch(desc, gout.device)     # noqa: F841
    _chain(desc, pre=wino_pre, nxt=wino_next, nxt_gout=wino_next_gout)
    _lib.check(_lib.lib().dvsof_conv2d_dgrad(
        ctypes.byref(desc), weight_t.data_ptr(), gout.data_ptr(), arr,
        bwd_act, _lib.stream()), 'dvsof_conv2d_dgrad')
    _chain(desc)


def flow_fold_weights(w, Cout, Ctot, cx_off, Cx, cf_off, wh):
    """-> w_eff [Cout][9][Ctot-2] (dvsof_flow_fold_weights)."""
    out = torch.empty(Cout * 9 * (Ctot - 2), dtype=torch.float32, device=w.device)
    _lib.check(_lib.lib().dvsof_flow_fold_weights(
        w.data_ptr(), Cout, Ctot, cx_off, Cx, cf_off, wh.data_ptr(),
        out.data_ptr(), _lib.stream()), 'dvsof_flow_fold_weights')
    return out


def flow_fold_bias(w, Cout, Ctot, cf_off, bh, bias):
    """-> (bias_eff [Cout], bias_cls [9][Cout]) (dvsof_flow_fold_bias)."""
    be = torch.empty(Cout, dtype=torch.float32, device=w.device)
    bc = torch.empty(9 * Cout, dtype=torch.float32, device=w.device)
    _lib.check(_lib.lib().dvsof_flow_fold_bias(
        w.data_ptr(), Cout, Ctot, cf_off, bh.data_ptr(), _lib.ptr(bias),
        be.data_ptr(), bc.data_ptr(), _lib.stream()), 'dvsof_flow_fold_bias')
    return be, bc


def flow_fold_grads(dW, w, Cout, Ctot, cx_off, Cx, cf_off, wh, bh, db_conv, g,
                    B, H, W, dwh, dbh):
    """Flow columns of dW and the head's gradient additions
    (dvsof_flow_fold_grads)."""
    n = _lib.lib().dvsof_flow_fold_workspace_bytes(B, Cout)
    ws = torch.empty(n // 4 + 1, dtype=torch.float32, device=g.device)
    _lib.check(_lib.lib().dvsof_flow_fold_grads(
        dW.data_ptr(), w.data_ptr(), Cout, Ctot, cx_off, Cx, cf_off,
        wh.data_ptr(), bh.data_ptr(), db_conv.data_ptr(), g.data_ptr(), B, H,
        W, dwh.data_ptr(), dbh.data_ptr(), ws.data_ptr(), ws.numel() * 4,
        _lib.stream()), 'dvsof_flow_fold_grads')


def conv_wgrad(desc, gout, dweight, dbias, gout16=None, skip_flat=False,
               wino_gout=None):
    """gout16: bf16 twin of gout (mode 3): with the sources' twins the
    vector members' weight gradient streams bf16 through LDS.  skip_flat:
    the narrow planar members' columns are left to flow_fold_grads."""
    desc.gout16 = _lib.ptr(gout16)
    desc.flags = WGRAD_SKIP_FLAT if skip_flat else 0
    nbytes = _plan(desc)[4]
    ws = torch.empty(max(nbytes // 4, 4), dtype=torch.float32,
                     device=gout.device)
    kept = getattr(desc, '_wino_input', None)
    desc.winograd_input = kept.data_ptr() if kept is not None else None
    _chain(desc, gout=wino_gout)
    _lib.check(_lib.lib().dvsof_conv2d_wgrad(
        ctypes.byref(desc), gout.data_ptr(), dweight.data_ptr(),
        _lib.ptr(dbias), ws.data_ptr(), ws.numel() * 4, _lib.stream()),
        'dvsof_conv2d_wgrad')
    _chain(desc)


def head_fwd(x, w, bias, B, H, W, C):
    flow = torch.empty(B, 2, H, W, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().dvsof_flow_head_fwd(
        x.data_ptr(), w.data_ptr(), _lib.ptr(bias), flow.data_ptr(), B, H, W,
        C, _lib.stream()), 'dvsof_flow_head_fwd')
    return flow


_lib.register('dvsof_flow_heads_fwd', _i, [_i, _P(_vp), _P(_vp), _P(_vp), _P(_vp), _i,
                                           _P(_i), _P(_i), _P(_i), _vp])


def heads_fwd(items, B, out=None):
    """items: [(x NHWC, w, bias, H, W, C)] (at most 4) -> [flow NCHW] from ONE
    launch (dvsof_flow_heads_fwd); ``out``: preallocated flow tensors."""
    n = len(items)
    flows = out if out is not None else [
        torch.empty(B, 2, H, W, dtype=torch.float32, device=x.device)
        for x, _, _, H, W, _ in items]
    arr = lambda ts: (_vp * n)(*[_lib.ptr(t) for t in ts])        # noqa: E731
    ints = lambda k: (_i * n)(*[it[k] for it in items])             # noqa: E731
    _lib.check(_lib.lib().dvsof_flow_heads_fwd(
        n, arr([it[0] for it in items]), arr([it[1] for it in items]),
        arr([it[2] for it in items]), arr(flows), B, ints(3), ints(4), ints(5),
        _lib.stream()), 'dvsof_flow_heads_fwd')
    return flows


def dgrad_fuses_head(desc):
    """Does this layer's data gradient accept dsts[i]['head_w'] / ['head_gflow']
    (the flow head on a source folded into the epilogue)?"""
    return bool(_lib.lib().dvsof_conv2d_dgrad_fuses_head(ctypes.byref(desc)))


def dgrad_head_rows(desc):
    """> 0: the layer's data gradient is the nine-product kernel (rows of its
    head_part buffer, dvsof_conv2d_dgrad_head_rows)."""
    return _lib.lib().dvsof_conv2d_dgrad_head_rows(ctypes.byref(desc))


def dgrad_head_part(desc, C, device):
    """Partial-sum buffer for dsts[0]['head_part'] (None: not available)."""
    rows = _lib.lib().dvsof_conv2d_dgrad_head_rows(ctypes.byref(desc))
    if rows <= 0:
        return None
    return torch.empty(rows, 2 * C + 2, dtype=torch.float32, device=device)


def head_reduce(part, C, dw, dbias):
    """dw, dbias of a flow head from the partials a data gradient left
    (dsts[0]['head_part']; dvsof_flow_head_reduce)."""
    _lib.check(_lib.lib().dvsof_flow_head_reduce(
        part.data_ptr(), part.shape[0], C, dw.data_ptr(), _lib.ptr(dbias),
        _lib.stream()), 'dvsof_flow_head_reduce')


def head_bwd(x, w, gflow, gx_in, actsrc, act, gx, dw, dbias, B, H, W, C,
             gx16=None):
    """gx=None: the head's weight / bias gradient only (its data part went
    into the epilogue of the data gradient that produced the tensor)."""
    nbytes = _lib.lib().dvsof_flow_head_bwd_workspace_bytes(B, H, W, C)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().dvsof_flow_head_bwd(
        x.data_ptr(), w.data_ptr(), gflow.data_ptr(), _lib.ptr(gx_in),
        _lib.ptr(actsrc), act, _lib.ptr(gx), dw.data_ptr(),
        _lib.ptr(dbias), B, H, W, C, ws.data_ptr(), nbytes, _lib.ptr(gx16),
        _lib.stream()), 'dvsof_flow_head_bwd')


def act_bwd(dy, actsrc, act, out=None):
    out = dy if out is None else out
    _lib.check(_lib.lib().dvsof_act_bwd(
        dy.data_ptr(), actsrc.data_ptr(), act, out.data_ptr(), dy.numel(),
        _lib.stream()), 'dvsof_act_bwd')
    return out


# ---- timing experiment (DVSOF_STALE_FORMS=1; results are WRONG after the first step): every
# weight form is made once and reused -- what the step would cost if the forms were free
import os as _os
if _os.environ.get('DVSOF_STALE_FORMS') == '1':
    def _memo(fn, key):
        cache = {}

        def inner(*a, **k):
            kk = key(*a, **k)
            if kk not in cache:
                cache[kk] = fn(*a, **k)
            return cache[kk]
        return inner

    def _dkey(d):
        return (d.nsrc, tuple(d.src[i].C for i in range(d.nsrc)), d.B, d.H, d.W, d.upsample,
                d.stride, d.Cout, d.mfma)
    prepare = _memo(prepare, lambda d, w, dg, phase_weights=None, want16=False:
                    (_dkey(d), w.data_ptr(), dg, phase_weights is not None and phase_weights.data_ptr(), want16))
    flow_fold_weights = _memo(flow_fold_weights, lambda w, *a: (w.data_ptr(),) + tuple(
        x if isinstance(x, int) else x.data_ptr() for x in a))
    flow_fold_bias = _memo(flow_fold_bias, lambda w, *a: (w.data_ptr(),) + tuple(
        x if isinstance(x, int) else (x.data_ptr() if x is not None else 0) for x in a))
