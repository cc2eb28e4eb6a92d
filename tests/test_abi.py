"""CPU checks of the drop-in boundary: the C-ABI library loads without a GPU
and exports every symbol include/dvsof.h declares; the Python side refuses to
run without device tensors (no CPU fallback)."""
import ctypes

import pytest
import torch

from dvs_of_training_framework_amd import _lib


def test_library_loads_and_exports_every_declared_symbol():
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    declared = _lib.declared_symbols()
    assert 'dvsof_loss_fwd' in declared and len(declared) >= 9
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in dvsof.h, not exported'


def test_every_declared_symbol_has_a_python_signature():
    from dvs_of_training_framework_amd import conv, optim  # noqa: F401 (register)
    for name in _lib.declared_symbols():
        assert name in _lib._SIGNATURES, name


def test_version_and_error_strings():
    lib = _lib.lib()
    assert lib.dvsof_version() == 100
    assert b'workspace' in lib.dvsof_error_string(-2)
    assert b'invalid' in lib.dvsof_error_string(-1)


def test_no_cpu_fallback():
    from dvs_of_training_framework_amd.loss import Losses
    ev = Losses([(4, 4)], 1, 'cpu')
    z = torch.zeros
    with pytest.raises(RuntimeError, match='no CPU implementation'):
        ev([z(1, 2, 4, 4)], z(1, 2), z(1, dtype=torch.long), z(2, 1, 4, 4),
           z(2), z(2, dtype=torch.long))


def test_product_package_never_imports_the_oracle():
    import pathlib
    pkg = pathlib.Path(_lib.__file__).parent
    for py in pkg.rglob('*.py'):
        text = py.read_text()
        assert 'import oracle' not in text and 'from oracle' not in text, py
