"""ctypes binding of libdvsof_hip.so (C ABI: include/dvsof.h).

There is NO fallback: if the library is missing or a call fails the product
raises.  PyTorch is used only for device memory and streams; every entry
point gets raw device pointers and the current HIP stream.
"""
import ctypes
import os
import re
from pathlib import Path

import torch

_PKG = Path(__file__).resolve().parent
# DVSOF_PROBE_LIB=1: the probe build (`make -C csrc probes`; timing probes of
# DVSOF_GCONV_DBG / DVSOF_LOSS_DBG compiled in) -- diagnostics tools only
LIB_PATH = _PKG / ('libdvsof_hip_probes.so' if os.environ.get('DVSOF_PROBE_LIB') == '1'
                   else 'libdvsof_hip.so')
if os.environ.get('DVSOF_LIB_PATH'):    # an experiment's variant build (tools/variant.sh)
    LIB_PATH = Path(os.environ['DVSOF_LIB_PATH']).resolve()
HEADER_PATH = _PKG.parent / 'include' / 'dvsof.h'
MAX_SCALES = 8

_vp = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_f = ctypes.c_float
_sz = ctypes.c_size_t


class LossScale(ctypes.Structure):
    """dvsof_loss_scale_t"""
    _fields_ = [('frames', _vp), ('flow', _vp), ('grad_flow', _vp),
                ('h', _i), ('w', _i)]


_SIGNATURES = {
    'dvsof_version': (_i, []),
    'dvsof_error_string': (ctypes.c_char_p, [_i]),
    'dvsof_comm_unique_id': (_i, [_vp]),
    'dvsof_comm_create': (_i, [ctypes.POINTER(_vp), _i, _i, _vp]),
    'dvsof_comm_destroy': (_i, [_vp]),
    'dvsof_allreduce_bucket': (_i, [_vp, _vp, _sz, _vp]),
    'dvsof_comm_create_loopback': (_i, [ctypes.POINTER(_vp), _i, _i]),
    'dvsof_comm_info': (_i, [_vp, ctypes.POINTER(_i), ctypes.POINTER(_i),
                             ctypes.POINTER(ctypes.c_ulonglong),
                             ctypes.POINTER(ctypes.c_ulonglong)]),
    'dvsof_exec_create': (_i, [_vp, ctypes.POINTER(_vp), _i,
                               ctypes.POINTER(_vp)]),
    'dvsof_exec_info': (_i, [_vp, ctypes.POINTER(_i), ctypes.POINTER(_i),
                             ctypes.POINTER(_i), ctypes.POINTER(_i),
                             ctypes.POINTER(_i), _i]),
    'dvsof_exec_launch': (_i, [_vp, _vp]),
    'dvsof_exec_calibrate': (_i, [_vp, _vp]),
    'dvsof_exec_node': (_i, [_vp, _i, ctypes.POINTER(_i), ctypes.POINTER(_f),
                             ctypes.POINTER(_i), ctypes.c_char_p, _i]),
    'dvsof_exec_plan': (_i, [_vp, ctypes.c_char_p, _i, ctypes.POINTER(_i)]),
    'dvsof_exec_destroy': (_i, [_vp]),
    'dvsof_exec_mark': (_i, [_i, _i, _vp, _sz, _vp]),
    'dvsof_exec_set_comm': (_i, [_vp, _vp, _vp]),
    'dvsof_exec_set_update_stream': (_i, [_vp, _vp]),
    'dvsof_exec_marks': (_i, [_vp, ctypes.POINTER(_i)]),
    'dvsof_exec_node_arg': (_i, [_vp, _i, _i, _sz, _vp]),
    'dvsof_exec_mark_window': (_i, [_vp, _i, ctypes.POINTER(_vp),
                                    ctypes.POINTER(_sz), ctypes.POINTER(_i),
                                    ctypes.POINTER(_i), _i,
                                    ctypes.POINTER(_i)]),
    'dvsof_count_image': (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp]),
    'dvsof_voxelize_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp,
                                _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'dvsof_voxelize_workspace_bytes': (_sz, [_i64, _i, _i, _i, _i]),
    'dvsof_voxelize_control_bytes': (_sz, [_i64, _i, _i, _i, _i]),
    'dvsof_voxelize_tiled': (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp,
                                  _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz,
                                  _i, _vp]),
    'dvsof_voxelize_encoded': (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp,
                                    _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz,
                                    _i, _vp]),
    'dvsof_resize_bilinear_ac': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'dvsof_loss_workspace_bytes': (_sz, [ctypes.POINTER(LossScale), _i, _i]),
    'dvsof_loss_fwd': (_i, [ctypes.POINTER(LossScale), _i, _i, _vp, _vp, _vp,
                            _vp, _vp, _sz, _vp]),
    'dvsof_loss_bwd': (_i, [ctypes.POINTER(LossScale), _i, _i, _vp, _vp, _vp,
                            _vp, _vp]),
    'dvsof_loss_fused': (_i, [ctypes.POINTER(LossScale), _i, _i, _vp, _vp,
                              ctypes.POINTER(_f), _f, _vp, _vp, _vp, _vp,
                              _sz, _vp]),
    'dvsof_augment_lut': (_i, [_vp, _i, _i, _i, _vp, _vp]),
    'dvsof_augment_frames': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i,
                                  _i, _vp, _vp]),
    'dvsof_augment_events': (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i, _i,
                                  _i, _vp, _vp, _vp]),
    'dvsof_loss_pyramid': (_i, [_vp, _i, _i, _i, ctypes.POINTER(_vp),
                                ctypes.POINTER(_i), ctypes.POINTER(_i), _i,
                                _vp]),
    'dvsof_loss_fused_pyramid': (_i, [_vp, _i, _i, _i,
                                      ctypes.POINTER(LossScale), _i, _i, _vp,
                                      _vp, ctypes.POINTER(_f), _f, _vp, _vp,
                                      _vp, _vp, _sz, _vp]),
}

_lib = None
LOADED_STAT = None


def declared_symbols():
    """Names of all entry points include/dvsof.h declares."""
    text = HEADER_PATH.read_text()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(dvsof_[a-z0-9_]+)\s*\(', text)))


def register(name, restype, argtypes):
    _SIGNATURES[name] = (restype, argtypes)
    if _lib is not None:
        fn = getattr(_lib, name)
        fn.restype, fn.argtypes = restype, argtypes


def lib():
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(
                f'{LIB_PATH} is missing: build it with '
                '`python -c "import __graft_entry__ as g; g.build()"` or '
                '`make -C dvs_of_training_framework_amd/csrc`. '
                'There is no CPU fallback for the HIP hot path.')
        _lib = ctypes.CDLL(str(LIB_PATH))
        st = LIB_PATH.stat()
        global LOADED_STAT      # the file these code objects came from (_audit.py)
        LOADED_STAT = (st.st_ino, st.st_size, st.st_mtime_ns)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.restype, fn.argtypes = res, args
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().dvsof_error_string(rc).decode()
        raise RuntimeError(f'{what} failed: {msg} (code {rc})')


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream():
    """Current torch HIP stream as a raw hipStream_t (one C call: this runs
    before every kernel launch, ~120 times per training step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                'the dvsof HIP path needs device tensors (got a CPU tensor); '
                'there is no CPU implementation in this package')
