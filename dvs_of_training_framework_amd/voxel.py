"""Event stream -> count image / voxel grid (HIP, csrc/voxel.hip).

``get_count_image`` mirrors /root/reference/utils/data.py:120-136;
``voxelize`` is the arithmetic behind ``Model.quantize`` (reference callers
utils/training.py:59-64, scripts/quantize_preprocessed.py:87-91), specified in
docs/VOXEL_SPEC.md.
"""
import numpy as np
import torch

from . import _lib


def get_count_image(events, imsize, device='cuda'):
    """events: [x, y, ...] arrays or tensors; -> uint64 numpy [H,W] like the
    reference (device tensors in, histogram by integer atomics)."""
    x, y = [torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v)
            .to(device=device, dtype=torch.long).contiguous()
            for v in events[:2]]
    H, W = int(imsize[0]), int(imsize[1])
    if x.numel():
        # np.ravel_multi_index raises for out-of-frame coordinates
        bad = ((x < 0) | (x >= W) | (y < 0) | (y >= H)).any()
        if bool(bad):
            raise ValueError('invalid entry in coordinates array')
    out = torch.empty(H, W, dtype=torch.int32, device=x.device)
    _lib.check(_lib.lib().dvsof_count_image(
        x.data_ptr(), y.data_ptr(), x.numel(), H, W, out.data_ptr(),
        _lib.stream()), 'dvsof_count_image')
    return out.cpu().numpy().astype(np.uint32).astype(np.uint64)


WS_CLEAN = 1       # DVSOF_VOX_WS_CLEAN
_WORKSPACES = {}
_WS_ROUND = 1 << 16


def _forget_workspace(ws):
    """A call on ``ws`` failed: its control words may be dirty -- the next
    call of that shape starts from a fresh, zero-filled workspace."""
    for k in [k for k, v in _WORKSPACES.items() if v is ws]:
        del _WORKSPACES[k]


def _checked(rc, what, ws):
    if rc != 0:
        _forget_workspace(ws)
    _lib.check(rc, what)


def _workspace(n, B, C, H, W, device):
    """-> (tensor, nbytes, flags).  The tiled voxeliser's control words clean
    up after themselves (include/dvsof.h), so a workspace is zero-filled ONCE
    and then reused by every call of the same shape on that device: no memset
    launch per call, and a hipGraph capture of the step (capture.py) holds
    kernels only -- the capture reuses the workspace its eager warm-up call
    made.  One voxelisation of a given shape at a time per device: calls on
    one stream are ordered; two streams voxelising the same shape concurrently
    must not share this cache (pass their own workspace through the C ABI)."""
    lib = _lib.lib()
    control = lib.dvsof_voxelize_control_bytes(n, B, C, H, W)
    if control == 0:        # thread-per-event kernel: plain scratch
        nbytes = lib.dvsof_voxelize_workspace_bytes(n, B, C, H, W)
        return (torch.empty(max(nbytes, 16), dtype=torch.uint8, device=device),
                nbytes, 0)
    # sized and keyed by the event count ROUNDED UP (the kernels accept a larger
    # buffer; the control region depends on B, C, H, W only): batches of
    # varying event counts share one workspace instead of allocating and
    # zero-filling one per distinct count
    n_up = (n + _WS_ROUND - 1) // _WS_ROUND * _WS_ROUND
    nbytes = max(lib.dvsof_voxelize_workspace_bytes(n_up, B, C, H, W),
                 lib.dvsof_voxelize_workspace_bytes(n, B, C, H, W))
    key = (n_up, control, B, C, H, W, str(device))
    ws = _WORKSPACES.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            # no warm-up call made one: graph-owned scratch, zero-filled by a
            # kernel node of the capture (flags 0)
            return (torch.empty(nbytes, dtype=torch.uint8, device=device),
                    nbytes, 0)
        if len(_WORKSPACES) >= 16:      # a handful of shapes is the normal case:
            _WORKSPACES.pop(next(iter(_WORKSPACES)))    # forget the oldest
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        ws[:control].zero_()
        _WORKSPACES[key] = ws
    return ws, nbytes, WS_CLEAN


def voxelize(events, t0, t1, B, C, H, W, debug=False):
    """events: dict of device tensors in the wire format; t0/t1: float32[B].
    -> grid float32 [B,C,H,W] (and bin0 int32[n], lin0 int64[n] if debug)."""
    x = events['x'].contiguous()
    y = events['y'].contiguous()
    t = events['timestamp'].contiguous().float()
    p = events['polarity'].contiguous()
    s = events['sample_index'].contiguous()
    _lib.require_cuda(x, y, t, p, s, t0, t1)
    for v in (x, y, p, s):
        assert v.dtype == torch.long, 'event columns are int64 on the wire'
    n = x.numel()
    out = torch.empty(B, C, H, W, dtype=torch.float32, device=t0.device)
    bin0 = lin0 = None
    if debug:
        bin0 = torch.empty(max(n, 1), dtype=torch.int32, device=t0.device)
        lin0 = torch.empty(max(n, 1), dtype=torch.int64, device=t0.device)
    lib = _lib.lib()
    ws, nbytes, flags = _workspace(n, B, C, H, W, t0.device)
    _checked(lib.dvsof_voxelize_tiled(
        x.data_ptr(), y.data_ptr(), t.data_ptr(), p.data_ptr(), s.data_ptr(),
        n, t0.contiguous().data_ptr(), t1.contiguous().data_ptr(), B, C, H, W,
        out.data_ptr(), _lib.ptr(bin0), _lib.ptr(lin0), _lib.ptr(ws), nbytes,
        flags, _lib.stream()), 'dvsof_voxelize_tiled', ws)
    if debug:
        return out, bin0[:n], lin0[:n]
    return out


COMPACT_KEYS = ('x', 'y', 'timestamp', 'polarity', 'sample_event_offsets')


def is_compact(events):
    """Compact event columns = the reference's ENCODED columns
    (utils/dataset.py:286-289: int16 x, int16 y, float32 timestamp, bool
    polarity; 9 bytes/event) plus ``sample_event_offsets`` int64[B+1], the
    first event of every sample (what ``events_per_element`` /
    ``elements_per_sample`` add up to, encoding.sample_event_offsets)."""
    return isinstance(events, dict) and 'sample_event_offsets' in events


def voxelize_compact(events, t0, t1, B, C, H, W, debug=False):
    """Voxel grid [B,C,H,W] straight from the 9 B/event compact columns
    (dvsof_voxelize_encoded): no expansion to the 44 B/event wire format."""
    dev = t0.device
    x = events['x'].to(dev, torch.short).contiguous()
    y = events['y'].to(dev, torch.short).contiguous()
    t = events['timestamp'].to(dev, torch.float32).contiguous()
    p = events['polarity'].to(dev, torch.uint8).contiguous()
    off = events['sample_event_offsets'].to(dev, torch.long).contiguous()
    _lib.require_cuda(x, t0, t1)
    assert off.numel() == B + 1, 'one offset per sample plus the end'
    n = x.numel()
    out = torch.empty(B, C, H, W, dtype=torch.float32, device=dev)
    bin0 = lin0 = None
    if debug:
        bin0 = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        lin0 = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    lib = _lib.lib()
    ws, nbytes, flags = _workspace(n, B, C, H, W, dev)
    _checked(lib.dvsof_voxelize_encoded(
        x.data_ptr(), y.data_ptr(), t.data_ptr(), p.data_ptr(),
        off.data_ptr(), n, t0.contiguous().data_ptr(),
        t1.contiguous().data_ptr(), B, C, H, W, out.data_ptr(),
        _lib.ptr(bin0), _lib.ptr(lin0), ws.data_ptr(), nbytes, flags,
        _lib.stream()), 'dvsof_voxelize_encoded', ws)
    if debug:
        return out, bin0[:n], lin0[:n]
    return out
