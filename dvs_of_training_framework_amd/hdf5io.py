"""Minimal HDF5 reader / writer over ``libhdf5`` through ctypes.

The reference stores its preprocessed (encoded / quantized) datasets with
h5py (utils/dataset.py:376-426 write, :505-548 read); h5py is not installed
here, the HDF5 C library is.  This module is the thin file backend those
functions need -- groups, dense datasets of the integer / float / bool dtypes
the encoded batch uses, row-range reads (``dataset[begin:end]``) -- with an
h5py-like surface (``File(path, mode)``, ``group[name]``, ``create_group``,
``create_dataset``, ``len``, slicing), so ``preprocessed.py`` reads like the
reference's code.  Files are interchangeable with h5py's: bools are written
as the enum(FALSE=0, TRUE=1) over int8 that h5py uses and read back from it.

Host-side I/O only; nothing here touches the GPU.
"""
import ctypes
import os
from pathlib import Path

import numpy as np
import torch

hid_t = ctypes.c_int64
herr_t = ctypes.c_int
hsize_t = ctypes.c_uint64
_c = ctypes

_CANDIDATES = ('libhdf5.so', 'libhdf5_serial.so', '/opt/conda/lib/libhdf5.so',
               '/usr/lib/x86_64-linux-gnu/libhdf5_serial.so',
               '/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so')
_lib = None

H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT = 0
H5S_ALL = 0
H5S_SELECT_SET = 0
H5T_INTEGER, H5T_FLOAT, H5T_ENUM = 0, 1, 8
H5T_SGN_NONE = 0
H5O_TYPE_GROUP, H5O_TYPE_DATASET = 0, 1
H5_INDEX_NAME, H5_ITER_INC = 0, 0


def available():
    try:
        lib()
        return True
    except OSError:
        return False


def lib():
    """The HDF5 C library (>= 1.10: 64-bit hid_t).  OSError if absent."""
    global _lib
    if _lib is not None:
        return _lib
    names = [os.environ['DVSOF_LIBHDF5']] if 'DVSOF_LIBHDF5' in os.environ \
        else list(_CANDIDATES)
    last = None
    for name in names:
        try:
            h = ctypes.CDLL(name)
            break
        except OSError as e:
            last = e
    else:
        raise OSError('libhdf5 not found (tried %s; set DVSOF_LIBHDF5): %s'
                      % (', '.join(names), last))
    sig = {
        'H5open': (herr_t, []),
        'H5get_libversion': (herr_t, [_c.POINTER(_c.c_uint)] * 3),
        'H5Fcreate': (hid_t, [_c.c_char_p, _c.c_uint, hid_t, hid_t]),
        'H5Fopen': (hid_t, [_c.c_char_p, _c.c_uint, hid_t]),
        'H5Fclose': (herr_t, [hid_t]),
        'H5Gcreate2': (hid_t, [hid_t, _c.c_char_p, hid_t, hid_t, hid_t]),
        'H5Gopen2': (hid_t, [hid_t, _c.c_char_p, hid_t]),
        'H5Gclose': (herr_t, [hid_t]),
        'H5Gget_num_objs': (herr_t, [hid_t, _c.POINTER(hsize_t)]),
        'H5Gget_objname_by_idx': (_c.c_ssize_t, [hid_t, hsize_t, _c.c_char_p,
                                                 _c.c_size_t]),
        'H5Gget_objtype_by_idx': (_c.c_int, [hid_t, hsize_t]),
        'H5Lexists': (_c.c_int, [hid_t, _c.c_char_p, hid_t]),
        'H5Dcreate2': (hid_t, [hid_t, _c.c_char_p, hid_t, hid_t, hid_t, hid_t,
                               hid_t]),
        'H5Dopen2': (hid_t, [hid_t, _c.c_char_p, hid_t]),
        'H5Dclose': (herr_t, [hid_t]),
        'H5Dget_space': (hid_t, [hid_t]),
        'H5Dget_type': (hid_t, [hid_t]),
        'H5Dwrite': (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, _c.c_void_p]),
        'H5Dread': (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, _c.c_void_p]),
        'H5Screate_simple': (hid_t, [_c.c_int, _c.POINTER(hsize_t),
                                     _c.POINTER(hsize_t)]),
        'H5Sget_simple_extent_ndims': (_c.c_int, [hid_t]),
        'H5Sget_simple_extent_dims': (_c.c_int, [hid_t, _c.POINTER(hsize_t),
                                                 _c.POINTER(hsize_t)]),
        'H5Sselect_hyperslab': (herr_t, [hid_t, _c.c_int, _c.POINTER(hsize_t),
                                         _c.POINTER(hsize_t),
                                         _c.POINTER(hsize_t),
                                         _c.POINTER(hsize_t)]),
        'H5Sclose': (herr_t, [hid_t]),
        'H5Tget_class': (_c.c_int, [hid_t]),
        'H5Tget_size': (_c.c_size_t, [hid_t]),
        'H5Tget_sign': (_c.c_int, [hid_t]),
        'H5Tget_super': (hid_t, [hid_t]),
        'H5Tenum_create': (hid_t, [hid_t]),
        'H5Tenum_insert': (herr_t, [hid_t, _c.c_char_p, _c.c_void_p]),
        'H5Tclose': (herr_t, [hid_t]),
        'H5Eset_auto2': (herr_t, [hid_t, _c.c_void_p, _c.c_void_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(h, name)
        fn.restype, fn.argtypes = res, args
    h.H5open()
    h.H5Eset_auto2(0, None, None)   # errors come back as return codes: we raise
    _lib = h
    return h


def _native(name):
    return hid_t.in_dll(lib(), f'H5T_NATIVE_{name}_g').value


_NP2H5 = {np.dtype(np.int8): 'INT8', np.dtype(np.uint8): 'UINT8',
          np.dtype(np.int16): 'INT16', np.dtype(np.uint16): 'UINT16',
          np.dtype(np.int32): 'INT32', np.dtype(np.uint32): 'UINT32',
          np.dtype(np.int64): 'INT64', np.dtype(np.uint64): 'UINT64',
          np.dtype(np.float32): 'FLOAT', np.dtype(np.float64): 'DOUBLE'}


def _check(rc, what):
    if rc < 0:
        raise OSError(f'HDF5: {what} failed')
    return rc


def _bool_type():
    """h5py's representation of numpy bool: enum over int8, FALSE=0, TRUE=1."""
    h = lib()
    t = _check(h.H5Tenum_create(_native('INT8')), 'H5Tenum_create')
    for name, v in ((b'FALSE', 0), (b'TRUE', 1)):
        val = ctypes.c_int8(v)
        _check(h.H5Tenum_insert(t, name, ctypes.byref(val)), 'H5Tenum_insert')
    return t


def _to_numpy(data):
    if isinstance(data, torch.Tensor):
        return data.detach().cpu().contiguous().numpy()
    return np.ascontiguousarray(data)


class Dataset:
    """An open dataset: ``len``, ``shape``, ``dtype``, ``ds[a:b]``, ``ds[...]``
    and conversion by ``torch.tensor(ds)`` / ``np.asarray(ds)``."""

    def __init__(self, did, name):
        self._id, self.name = did, name
        h = lib()
        sp = _check(h.H5Dget_space(did), 'H5Dget_space')
        nd = _check(h.H5Sget_simple_extent_ndims(sp), 'ndims')
        dims = (hsize_t * max(nd, 1))()
        if nd:
            h.H5Sget_simple_extent_dims(sp, dims, None)
        h.H5Sclose(sp)
        self.shape = tuple(int(dims[i]) for i in range(nd))
        t = _check(h.H5Dget_type(did), 'H5Dget_type')
        self.dtype, self._is_bool = self._numpy_type(t)
        h.H5Tclose(t)

    @staticmethod
    def _numpy_type(t):
        h = lib()
        cls, size = h.H5Tget_class(t), h.H5Tget_size(t)
        if cls == H5T_ENUM:         # h5py bool
            base = h.H5Tget_super(t)
            ok = h.H5Tget_class(base) == H5T_INTEGER and h.H5Tget_size(base) == 1
            h.H5Tclose(base)
            if not ok:
                raise TypeError('only the 1-byte bool enum is supported')
            return np.dtype(np.bool_), True
        if cls == H5T_FLOAT:
            return np.dtype({4: np.float32, 8: np.float64}[size]), False
        if cls == H5T_INTEGER:
            unsigned = h.H5Tget_sign(t) == H5T_SGN_NONE
            return np.dtype(f'{"u" if unsigned else "i"}{size}'), False
        raise TypeError(f'unsupported HDF5 type class {cls}')

    def __len__(self):
        if not self.shape:
            raise TypeError('scalar dataset has no len()')
        return self.shape[0]

    def _read(self, begin, end):
        h = lib()
        mem_dtype = np.dtype(np.int8) if self._is_bool else self.dtype
        if not self.shape:
            out = np.empty((), mem_dtype)
            _check(h.H5Dread(self._id, _native(_NP2H5[mem_dtype]), H5S_ALL,
                             H5S_ALL, H5P_DEFAULT, out.ctypes.data), 'H5Dread')
            return out.astype(np.bool_) if self._is_bool else out
        n = max(end - begin, 0)
        shape = (n,) + self.shape[1:]
        out = np.empty(shape, mem_dtype)
        if out.size:
            nd = len(self.shape)
            start = (hsize_t * nd)(begin, *([0] * (nd - 1)))
            count = (hsize_t * nd)(*shape)
            fs = _check(h.H5Dget_space(self._id), 'H5Dget_space')
            _check(h.H5Sselect_hyperslab(fs, H5S_SELECT_SET, start, None, count,
                                         None), 'H5Sselect_hyperslab')
            ms = _check(h.H5Screate_simple(nd, count, None), 'H5Screate_simple')
            rc = h.H5Dread(self._id, _native(_NP2H5[mem_dtype]), ms, fs,
                           H5P_DEFAULT, out.ctypes.data)
            h.H5Sclose(ms)
            h.H5Sclose(fs)
            _check(rc, 'H5Dread')
        return out.astype(np.bool_) if self._is_bool else out

    def __getitem__(self, key):
        if key is Ellipsis or key == ():
            return self._read(0, self.shape[0] if self.shape else 0)
        if isinstance(key, slice):
            begin, end, step = key.indices(self.shape[0])
            assert step == 1, 'only contiguous row ranges'
            return self._read(begin, end)
        if isinstance(key, (int, np.integer)):
            i = int(key) + (self.shape[0] if key < 0 else 0)
            return self._read(i, i + 1)[0]
        raise TypeError(f'unsupported index {key!r}')

    def __array__(self, dtype=None, copy=None):
        a = self[...]
        return a if dtype is None else a.astype(dtype)

    def close(self):
        if self._id is not None:
            lib().H5Dclose(self._id)
            self._id = None

    __del__ = close


class Group:
    def __init__(self, gid, name, owns=True):
        self._id, self.name, self._owns = gid, name, owns
        self._children = []

    def _track(self, obj):
        self._children.append(obj)
        return obj

    def __contains__(self, name):
        return lib().H5Lexists(self._id, name.encode(), H5P_DEFAULT) > 0

    def keys(self):
        h = lib()
        n = hsize_t()
        _check(h.H5Gget_num_objs(self._id, ctypes.byref(n)), 'H5Gget_num_objs')
        out = []
        for i in range(n.value):
            size = h.H5Gget_objname_by_idx(self._id, i, None, 0)
            buf = ctypes.create_string_buffer(size + 1)
            h.H5Gget_objname_by_idx(self._id, i, buf, size + 1)
            out.append(buf.value.decode())
        return out

    def __iter__(self):
        return iter(self.keys())

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def __getitem__(self, name):
        h = lib()
        if name not in self:
            raise KeyError(f'{self.name}/{name}')
        idx = self.keys().index(name)
        if h.H5Gget_objtype_by_idx(self._id, idx) == H5O_TYPE_GROUP:
            gid = _check(h.H5Gopen2(self._id, name.encode(), H5P_DEFAULT),
                         'H5Gopen2')
            return self._track(Group(gid, f'{self.name}/{name}'))
        did = _check(h.H5Dopen2(self._id, name.encode(), H5P_DEFAULT),
                     'H5Dopen2')
        return self._track(Dataset(did, f'{self.name}/{name}'))

    def create_group(self, name):
        gid = _check(lib().H5Gcreate2(self._id, name.encode(), H5P_DEFAULT,
                                      H5P_DEFAULT, H5P_DEFAULT), 'H5Gcreate2')
        return self._track(Group(gid, f'{self.name}/{name}'))

    def create_dataset(self, name, data):
        h = lib()
        a = _to_numpy(data)
        is_bool = a.dtype == np.bool_
        if is_bool:
            ftype, mtype, a = _bool_type(), None, a.astype(np.int8)
        elif a.dtype in _NP2H5:
            ftype = mtype = _native(_NP2H5[a.dtype])
        else:
            raise TypeError(f'unsupported dtype {a.dtype} for {name}')
        if is_bool:
            mtype = ftype
        dims = (hsize_t * max(a.ndim, 1))(*a.shape)
        sp = _check(h.H5Screate_simple(a.ndim, dims, None), 'H5Screate_simple')
        did = h.H5Dcreate2(self._id, name.encode(), ftype, sp, H5P_DEFAULT,
                           H5P_DEFAULT, H5P_DEFAULT)
        rc = 0
        if did >= 0 and a.size:
            rc = h.H5Dwrite(did, mtype, H5S_ALL, H5S_ALL, H5P_DEFAULT,
                            a.ctypes.data)
        h.H5Sclose(sp)
        if is_bool:
            h.H5Tclose(ftype)
        _check(did, f'H5Dcreate2({name})')
        h.H5Dclose(did)
        _check(rc, f'H5Dwrite({name})')

    def close(self):
        for c in self._children:
            c.close()
        self._children = []
        if self._id is not None and self._owns:
            lib().H5Gclose(self._id)
        self._id = None


class File(Group):
    """``with File(path, 'r' | 'w') as f:`` (h5py.File's two basic modes)."""

    def __init__(self, path, mode='r'):
        h = lib()
        p = str(Path(path)).encode()
        if mode == 'r':
            fid = h.H5Fopen(p, H5F_ACC_RDONLY, H5P_DEFAULT)
        elif mode == 'w':
            fid = h.H5Fcreate(p, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        else:
            raise ValueError("mode must be 'r' or 'w'")
        if fid < 0:
            raise OSError(f'cannot open {path} (mode {mode})')
        super().__init__(fid, '', owns=False)
        self._fid = fid

    def close(self):
        super().close()
        if self._fid is not None:
            lib().H5Fclose(self._fid)
            self._fid = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    __del__ = close
