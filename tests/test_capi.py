"""CPU: the C-ABI library loads and exports every symbol include/pwg_kernels.h declares."""
import ctypes
import os
import re

import pytest

from parallelwavegan_amd import _lib, ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "pwg_kernels.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pwg_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 8
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/pwg_kernels.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in parallelwavegan_amd/_lib.py"
    assert set(_lib.SIGNATURES) == set(syms)


def test_version_and_arch():
    l = _lib.lib()
    assert l.pwg_abi_version() == _lib.ABI_VERSION
    assert l.pwg_target_arch() == 950


def test_bad_descriptor_reports_error_without_gpu():
    d = ops.make_conv_desc(1, 6, 8, 16, 16, 3, groups=4)  # 6 % 4 != 0
    assert _lib.lib().pwg_conv1d_packed_weight_floats(ctypes.byref(d)) == 0
    assert b"groups" in _lib.lib().pwg_last_error()


def test_packed_weight_size():
    d = ops.make_conv_desc(1, 80, 512, 32, 32, 7, pad_left=3)
    assert ops.packed_weight_floats(d) == 7 * 80 * 512
    dt = ops.make_conv_desc(1, 512, 256, 32, 256, 16, stride=8, pad_left=4, transposed=True)
    assert ops.packed_weight_floats(dt) == 2 * 512 * (8 * 256)


def test_ops_refuse_cpu_tensors():
    import torch

    d = ops.make_conv_desc(1, 4, 4, 8, 8, 3, pad_left=1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.conv1d_forward(d, torch.zeros(1, 4, 8), torch.zeros(16 * 32 * 3))
