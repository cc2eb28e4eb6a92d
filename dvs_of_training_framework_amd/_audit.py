"""Pointer audit of a captured step (debug tool; capture.CapturedTrainStep.audit).

A captured graph replays kernels with the ARGUMENT VALUES of the recording.
A device pointer among them that belongs to an eager torch tensor which nobody
holds any more is a fault waiting for the allocator to reuse the block (round
2: ``Model._layout_cache`` dropped index vectors a graph still read).  The
rule the capture relies on: every pointer a kernel argument carries lies

  * in the graph's private memory pool (tensors made while recording), or
  * in a tensor the step keeps alive: ``step._keep``, the static input
    buffers, model parameters / buffers, optimizer state, the step's outputs,
    module-level constants (the unit seed of loss.unit_backward).

This walk checks it node by node.  Argument layouts come from the AMDGPU code
object metadata of libdvsof_hip.so (``llvm-readelf --notes`` on the gfx950
code objects of the fat binary: per kernel the explicit arguments with offset,
size and kind); argument bytes come from the captured nodes
(``dvsof_exec_node_arg``).  ``global_buffer`` arguments are pointers;
by-value structs (descriptors with embedded pointers) are scanned as aligned
8-byte words, a word counting as a pointer when it falls inside a segment of
torch's caching allocator.  Kernels of other libraries (two ATen gathers in
the step) have no metadata here and are listed as ``foreign``.
"""
import ctypes
import re
import struct
import subprocess
from pathlib import Path

import torch

from . import _lib

_READELF = '/opt/rocm/lib/llvm/bin/llvm-readelf'
_MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'
_LAYOUTS = None


def _code_objects(path):
    data = Path(path).read_bytes()
    for m in re.finditer(re.escape(_MAGIC), data):
        p = m.start()
        n = struct.unpack_from('<Q', data, p + 24)[0]
        off = p + 32
        for _ in range(n):
            o, sz, ts = struct.unpack_from('<QQQ', data, off)
            triple = data[off + 24:off + 24 + ts].decode(errors='replace')
            off += 24 + ts
            if sz and 'amdgcn' in triple:
                yield data[p + o:p + o + sz]


def kernel_layouts():
    """{mangled kernel name: [(size, is_pointer)] of the explicit arguments}."""
    global _LAYOUTS
    if _LAYOUTS is not None:
        return _LAYOUTS
    import tempfile
    import yaml
    # dvsof_exec_node_arg is unchecked: the layouts must describe the library that is
    # LOADED (a rebuilt libdvsof_hip.so on disk under a running process would make the
    # argument indices / sizes below an out-of-bounds host read)
    _lib.lib()
    st = Path(_lib.LIB_PATH).stat()
    if (st.st_ino, st.st_size, st.st_mtime_ns) != _lib.LOADED_STAT:
        raise RuntimeError(f'{_lib.LIB_PATH} changed on disk since it was loaded: its '
                           'code-object metadata no longer describes the loaded kernels')
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for i, blob in enumerate(_code_objects(_lib.LIB_PATH)):
            f = Path(tmp) / f'{i}.co'
            f.write_bytes(blob)
            text = subprocess.run([_READELF, '--notes', str(f)], capture_output=True,
                                  text=True, check=True).stdout
            a = text.find('---')
            b = text.find('...', a)
            if a < 0:
                continue
            meta = yaml.safe_load(text[a + 3:b if b > 0 else None])
            for k in (meta or {}).get('amdhsa.kernels', []):
                args = [(int(x['.size']), x['.value_kind'] == 'global_buffer')
                        for x in k.get('.args', [])
                        if not str(x['.value_kind']).startswith('hidden_')]
                out[k['.name']] = args
    _LAYOUTS = out
    return out


def _tensors(obj, seen, out):
    if id(obj) in seen:
        return
    seen.add(id(obj))
    if torch.is_tensor(obj):
        if obj.is_cuda:
            out.append(obj)
    elif isinstance(obj, dict):
        for v in obj.values():
            _tensors(v, seen, out)
    elif isinstance(obj, (list, tuple, set)):
        for v in obj:
            _tensors(v, seen, out)


def held_ranges(step):
    """[(lo, hi)] address ranges of the storages the step keeps alive."""
    from . import loss as loss_mod
    roots = [step._keep, step.static, step.loss, getattr(step._terms, 'packed', None),
             list(step.model.parameters()), list(step.model.buffers()),
             [v for st in step.optimizer.state.values() for v in st.values()],
             [p.grad for p in step.model.parameters()],
             list(loss_mod._UNIT.values()), getattr(step.optimizer, '_dyn', None)]
    ts = []
    _tensors(roots, set(), ts)
    ranges = []
    for t in ts:
        s = t.untyped_storage()
        ranges.append((s.data_ptr(), s.data_ptr() + s.nbytes()))
    return ranges


def audit_step(step):
    assert step.executor is not None, 'the audit reads the nodes through the step executor'
    layouts = kernel_layouts()
    pool = tuple(step.graph.pool())
    segs = []
    for seg in torch.cuda.memory_snapshot():
        segs.append((seg['address'], seg['address'] + seg['total_size'],
                     tuple(seg.get('segment_pool_id', (0, 0))) == pool))
    held = held_ranges(step)

    def where(v):
        for lo, hi, in_pool in segs:
            if lo <= v < hi:
                if in_pool:
                    return 'pool'
                return 'held' if any(a <= v < b for a, b in held) else 'unheld'
        return None         # not a device address of this allocator

    x = step.executor
    res = {'nodes': 0, 'audited': 0, 'foreign': [], 'pointers': 0, 'pool': 0, 'held': 0,
           'unheld': []}
    i = 0
    lane, us, nw = (ctypes.c_int(), ctypes.c_float(), ctypes.c_int())
    buf = ctypes.create_string_buffer(2048)
    while _lib.lib().dvsof_exec_node(x._handle, i, ctypes.byref(lane), ctypes.byref(us),
                                     ctypes.byref(nw), buf, 2048) == 0:
        name = buf.value.decode(errors='replace')
        i += 1
        if name in ('(empty)', '?'):
            continue
        res['nodes'] += 1
        args = layouts.get(name)
        if args is None:
            res['foreign'].append(name[:80])
            continue
        res['audited'] += 1
        for k, (size, is_ptr) in enumerate(args):
            if size < 8:
                continue
            raw = x.node_arg(i - 1, k, size)
            words = struct.unpack_from(f'<{size // 8}Q', raw)
            for w in (words[:1] if is_ptr else words):
                kind = where(w)
                if kind is None:
                    continue
                res['pointers'] += 1
                if kind == 'unheld':
                    res['unheld'].append((i - 1, name[:60], k, hex(w)))
                else:
                    res[kind] += 1
    return res


def audit_exchange(step):
    """The rule the executor's BUCKET marks rely on (csrc/exec.hip): a kernel
    captured behind a bucket's exchange mark, but not behind its WAIT / the
    JOIN mark, inherits the mark's dependencies and may run BESIDE the
    collective -- so it must not carry a pointer into that bucket.  True today
    by construction (``Predictor._grad_targets`` closes a bucket behind its
    last writer, the next reader is the optimizer behind the WAIT / JOIN
    mark); checked at every recording that has marks, because at one rank the
    collective is an identity and a violation would change nothing there.
    -> {'marks', 'window_kernels', 'checked_pointers', 'violations': [...],
    'foreign': [...]}; capture.py refuses the recording on a violation.
    Every 8-byte word of a by-value parameter struct is read as a possible
    pointer (the code object does not describe struct members): the launch
    sites zero-initialise those structs, so that an unused member slot cannot
    hold stack residue.  (Round 4: ONE recording of tests/test_gpu_capture.py's
    accumulation scenario was refused for a word 0x704c00000000 -- not seen
    again in 5 reruns; three launch sites then still passed `WGradParams P;`
    with uninitialised unused slots, the likely source: a stale gradient
    pointer left on the stack by an earlier layer's call.  Unproven: the
    refusal's full text was lost; capture.py now prints every violation.)"""
    x = step.executor
    assert x is not None
    layouts = kernel_layouts()
    lib = _lib.lib()
    res = {'marks': 0, 'window_kernels': 0, 'checked_pointers': 0, 'violations': [],
           'foreign': [], 'update_ranges': 0}
    names = {}
    # With the optimizer inside the backward (optim.fuse_into_backward) a bucket's UPDATE runs on
    # a lane of its own behind the collective (csrc/exec.hip, XNode.xlane): the kernels of the
    # window then also run beside IT, and must not touch the bucket's parameters or optimizer
    # state either.  bucket pointer -> [(lo, hi)] of those tensors
    update_ranges = {}
    pred = getattr(step.model, 'predictor', step.model)
    if getattr(pred, 'bucket_hook', None) is not None and hasattr(pred, '_buckets'):
        params = pred.param_list()
        flats, _ = pred._buckets(params)
        for b, units in enumerate(pred.BUCKETS):
            rs = []
            for u in units:
                for i in pred.UNIT_PARAMS[u]:
                    ts = [params[i].data] + [v for v in step.optimizer.state.get(params[i], {}).values()
                                             if torch.is_tensor(v) and v.is_cuda]
                    for t in ts:
                        st_ = t.untyped_storage()
                        rs.append((st_.data_ptr(), st_.data_ptr() + st_.nbytes()))
            update_ranges[flats[b].data_ptr()] = rs

    def name_of(i):
        if i not in names:
            buf = ctypes.create_string_buffer(2048)
            _lib.check(lib.dvsof_exec_node(x._handle, i, None, None, None, buf, 2048),
                       'dvsof_exec_node')
            names[i] = buf.value.decode(errors='replace')
        return names[i]
    k = 0
    while True:
        ptr, n, index, count = ctypes.c_void_p(), ctypes.c_size_t(), ctypes.c_int(), ctypes.c_int()
        cap = 4096
        nodes = (ctypes.c_int * cap)()
        if lib.dvsof_exec_mark_window(x._handle, k, ctypes.byref(ptr), ctypes.byref(n),
                                      ctypes.byref(index), nodes, cap, ctypes.byref(count)) != 0:
            break
        k += 1
        res['marks'] += 1
        lo = ptr.value or 0
        hi = lo + 4 * n.value
        extra = update_ranges.get(lo, [])
        res['update_ranges'] += len(extra)
        for i in list(nodes)[:min(count.value, cap)]:
            name = name_of(i)
            res['window_kernels'] += 1
            args = layouts.get(name)
            if args is None:        # not one of ours (ATen): never handed a gradient bucket
                if name[:80] not in res['foreign']:
                    res['foreign'].append(name[:80])
                continue
            for a, (size, is_ptr) in enumerate(args):
                if size < 8:
                    continue
                raw = x.node_arg(i, a, size)
                words = struct.unpack_from(f'<{size // 8}Q', raw)
                for w in (words[:1] if is_ptr else words):
                    res['checked_pointers'] += 1
                    if lo <= w < hi:
                        res['violations'].append((index.value, i, name[:60], a, hex(w)))
                    elif any(a_ <= w < b_ for a_, b_ in extra):
                        res['violations'].append((index.value, i, name[:60], a, hex(w), 'update'))
    return res
