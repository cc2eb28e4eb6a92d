#!/usr/bin/env python3
"""Extract the LITERAL golden data of the reference's encoded/quantized batch
format tests into tests/golden/encoding.pt (container-only; needs
/root/reference).  The reference test modules cannot be imported (h5py etc.
are missing), so only the literal-building statements of their
``setup_class`` methods and the ``begin/end/gt`` literals of
``test_batch_selection_indices`` are evaluated; nothing but tensors, numbers
and dict/list structure is stored (a fixture is data)."""
import ast
import sys
from pathlib import Path
from types import SimpleNamespace

import torch

REPO = Path(__file__).resolve().parent.parent
REF = Path('/root/reference/tests/dataset')
OUT = REPO / 'tests' / 'golden' / 'encoding.pt'


def class_method_src(path, cls, method):
    src = path.read_text()
    tree = ast.parse(src)
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for m in node.body:
                if isinstance(m, ast.FunctionDef) and m.name == method:
                    return m, src
    raise KeyError((cls, method))


def run_setup(path, cls):
    m, src = class_method_src(path, cls, 'setup_class')
    body = ast.Module(body=m.body, type_ignores=[])
    ns = {'torch': torch, 'self': SimpleNamespace()}
    exec(compile(body, str(path), 'exec'), ns)
    return vars(ns['self'])


def range_cases(path):
    m, _ = class_method_src(path, 'TestDatasetEncoding', 'test_batch_selection_indices')
    cases, cur = [], {}
    for st in m.body:
        if isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Name):
            name = st.targets[0].id
            if name in ('begin', 'end', 'gt'):
                cur[name] = ast.literal_eval(st.value)
                if name == 'gt':
                    cases.append(dict(cur))
    return cases


def main():
    enc = run_setup(REF / 'test_encoding.py', 'TestDatasetEncoding')
    qua = run_setup(REF / 'test_quantization.py', 'TestQuantized')
    data = {'encoding': {k: enc[k] for k in ('decoded', 'encoded', 'encoded_parts')},
            'ranges': range_cases(REF / 'test_encoding.py'),
            'quantized': {k: qua[k] for k in ('decoded_batch', 'encoded_batch',
                                              'decoded_batches', 'encoded_batches')}}
    torch.save(data, OUT)
    print('written', OUT, OUT.stat().st_size, 'bytes;', len(data['ranges']), 'range cases')


if __name__ == '__main__':
    main()
