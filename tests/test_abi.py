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


def _desc(B, H, W, cin, cout, layout=0, stride=1, up=False, mfma=0, nsrc=1):
    """Descriptor with dummy (never dereferenced) source pointers: the
    planning entry points below only read the shape fields."""
    from dvs_of_training_framework_amd import conv as C
    d = C.ConvDesc()
    d.nsrc = nsrc
    for i in range(nsrc):
        d.src[i].p, d.src[i].C, d.src[i].layout = 4096, cin, layout
    d.B, d.H, d.W = B, H, W
    d.upsample, d.ksize, d.stride, d.pad = int(up), 3, stride, 1
    d.Cout, d.act, d.mfma = cout, C.ACT_RELU, mfma
    return d


def test_winograd_planning_is_a_pure_function_of_the_shape():
    """Host-side dispatch of the wide 3x3 layers (no GPU needed): F(4x4,3x3)
    from 64 tiles on 4-aligned images, F(2x2,3x3) from 128 tiles, the direct
    kernel below (batch-1 inference), for narrow / strided / up-sampling /
    multi-source layers and for bf16-rounded operands; prepared-weight and
    scratch sizes follow the form."""
    from dvs_of_training_framework_amd import conv as C
    lib = C._lib.lib()
    ref = ctypes.byref

    def plan(d):
        return (lib.dvsof_conv2d_winograd_tile(ref(d), 0),
                lib.dvsof_conv2d_winograd_tile(ref(d), 2),
                lib.dvsof_conv2d_scratch_bytes(ref(d)),
                lib.dvsof_conv2d_fwd_weight_elems(ref(d)),
                lib.dvsof_conv2d_dgrad_weight_elems(ref(d)))
    raw = 512 * 512 * 9
    # the headline residual layer: batch 8, 16x16, 512 -> 512: 128 tiles of 4x4
    f, w, scratch, nf, ndg = plan(_desc(8, 16, 16, 512, 512))
    assert (f, w) == (4, 4) and nf == ndg == 36 * 512 * 512
    assert scratch == 36 * 128 * (512 + 512) * 4
    # batch 4: 64 4x4 tiles forward, weight gradient falls back to the 2x2 form (< 128 tiles)
    assert plan(_desc(4, 16, 16, 512, 512))[:2] == (4, 2)
    # batch 2: 32 4x4 tiles -> 2x2 form (128 tiles); batch 1: direct
    assert plan(_desc(2, 16, 16, 512, 512))[:2] == (2, 2)
    assert plan(_desc(1, 16, 16, 512, 512)) == (0, 0, 0, raw, raw)
    # W not a multiple of 4 -> 2x2 form; odd sizes -> direct
    assert plan(_desc(8, 16, 18, 256, 256))[0] == 2
    assert plan(_desc(8, 15, 16, 256, 256))[0] == 0
    # bf16x3 operands keep the 2x2 form, bf16-rounded operands the direct kernel
    assert plan(_desc(8, 16, 16, 512, 512, mfma=2))[:2] == (2, 2)
    assert plan(_desc(8, 16, 16, 512, 512, mfma=1))[0] == 0
    # not a wide 3x3 stride-1 single-source NHWC layer -> direct
    for d in (_desc(8, 16, 16, 128, 512), _desc(8, 16, 16, 512, 128),
              _desc(8, 32, 32, 512, 512, stride=2), _desc(8, 16, 16, 512, 512, up=True),
              _desc(8, 16, 16, 512, 512, layout=1), _desc(8, 16, 16, 256, 256, nsrc=2)):
        assert lib.dvsof_conv2d_winograd_tile(ref(d), 0) == 0
        assert lib.dvsof_conv2d_scratch_bytes(ref(d)) == 0


def test_tile_ids_follow_the_measured_rule():
    """64x64 for every forward / data-gradient problem with more than 32
    output channels, 128x32 for the 32-channel stage (profiles/round1/i_tile_sweep.txt)."""
    from dvs_of_training_framework_amd import conv as C
    lib = C._lib.lib()
    assert lib.dvsof_conv2d_tile_id(ctypes.byref(_desc(8, 64, 64, 64, 128, stride=2)), 0) == 3
    assert lib.dvsof_conv2d_tile_id(ctypes.byref(_desc(8, 128, 128, 64, 32, up=True)), 0) == 5
    assert lib.dvsof_conv2d_tile_id(ctypes.byref(_desc(1, 16, 16, 512, 512)), 0) == 3
