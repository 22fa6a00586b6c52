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


def test_models_and_losses_have_no_cpu_path():
    """The product path never computes on the host: every model family, the losses and the optimizers
    raise on CPU tensors instead of falling back to ATen."""
    import torch

    from parallelwavegan_amd import losses, models, optimizers

    g = models.HiFiGANGenerator(channels=16, upsample_scales=(2, 2), upsample_kernel_sizes=(4, 4),
                                resblock_kernel_sizes=(3,), resblock_dilations=[(1,)])
    cases = [
        lambda: g(torch.zeros(1, 80, 8)),
        lambda: models.ParallelWaveGANDiscriminator(layers=3, conv_channels=8)(torch.zeros(1, 1, 64)),
        lambda: models.MelGANGenerator(channels=32, upsample_scales=[2, 2], stacks=1)(torch.zeros(1, 80, 8)),
        lambda: models.StyleMelGANGenerator(in_channels=8, channels=16, noise_upsample_scales=[2],
                                            upsample_scales=[2, 1])(torch.zeros(1, 80, 2), torch.zeros(1, 8, 1)),
        lambda: models.UHiFiGANGenerator(channels=8, downsample_scales=(2,), downsample_kernel_sizes=(4,),
                                         upsample_scales=(2,), upsample_kernel_sizes=(4,), resblock_kernel_sizes=(3,),
                                         resblock_dilations=[(1,)])(torch.zeros(1, 80, 4), None, torch.zeros(1, 1, 8)),
        lambda: losses.MultiResolutionSTFTLoss()(torch.zeros(1, 4096), torch.zeros(1, 4096)),
        lambda: losses.MelSpectrogramLoss()(torch.zeros(1, 1, 4096), torch.zeros(1, 1, 4096)),
    ]
    for fn in cases:
        with pytest.raises(RuntimeError, match="no CPU fallback|MI355X"):
            fn()
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError, match="no CPU fallback|MI355X"):
        optimizers.Adam([p]).step()


def test_workspace_queries_answer_without_a_gpu():
    """Split-reduction workspaces are sized from the descriptor alone (caller-owned scratch)."""
    import ctypes

    from parallelwavegan_amd import _lib

    big = ops.make_conv_desc(16, 128, 128, 51200, 51200, 11, pad_left=5)            # fills the chip: no slices
    few = ops.make_conv_desc(16, 1024, 1024, 32, 32, 5, pad_left=2)                 # long reduction, 32 columns
    l = _lib.lib()
    assert l.pwg_conv1d_forward_workspace_floats(ctypes.byref(big)) == 0
    n = l.pwg_conv1d_forward_workspace_floats(ctypes.byref(few))
    assert n > 0 and n % (16 * 1024 * 32) == 0
    assert l.pwg_conv1d_backward_data_workspace_floats(ctypes.byref(few)) > 0
    assert l.pwg_conv1d_backward_weight_workspace_floats(ctypes.byref(big)) > 0


def test_resunit_geometry_queries_answer_without_a_gpu():
    """pwg_resunit_supported / _profitable / _packed_weight_floats are pure functions of the descriptor."""
    from parallelwavegan_amd import ops

    ok = [(32, 3, 1), (32, 3, 5), (32, 7, 3), (32, 11, 5), (64, 3, 1), (64, 7, 5), (64, 11, 5)]
    for c, k, d in ok:
        for pair in (True, False):
            assert ops.resunit_supported(ops.make_resunit_desc(2, c, 8192, k, d, pair)), (c, k, d, pair)
    # wider than the resident tile / not a multiple of 4 samples / even kernel / slope outside (0, 1)
    assert not ops.resunit_supported(ops.make_resunit_desc(2, 128, 8192, 3, 1))
    assert not ops.resunit_supported(ops.make_resunit_desc(2, 64, 8192, 11, 7))
    assert not ops.resunit_supported(ops.make_resunit_desc(2, 32, 8190, 3, 1))
    assert not ops.resunit_supported(ops.make_resunit_desc(2, 32, 8192, 4, 1))
    assert not ops.resunit_supported(ops.make_resunit_desc(2, 32, 8192, 3, 1, True, 0.0, 0.1))
    assert not ops.resunit_supported(ops.make_resunit_desc(2, 32, 8192, 3, 1, True, 0.1, 1.5))
    # the measured choice: every C = 32 unit, C = 64 up to k = 7; single-convolution units always
    assert ops.resunit_profitable(ops.make_resunit_desc(2, 32, 8192, 11, 5))
    assert ops.resunit_profitable(ops.make_resunit_desc(2, 64, 8192, 7, 3))
    assert not ops.resunit_profitable(ops.make_resunit_desc(2, 64, 8192, 11, 5))
    assert ops.resunit_profitable(ops.make_resunit_desc(2, 64, 8192, 11, 5, False))
    assert _lib.lib().pwg_resunit_packed_weight_floats(64, 11) == 11 * 64 * 64


def test_fused_optimizers_reject_options_that_change_the_update_rule():
    """``maximize=True`` & co. must not be swallowed (a recipe would silently train with another rule);
    implementation selectors (foreach / fused / capturable) are accepted."""
    import pytest
    import torch

    from parallelwavegan_amd.optimizers import fused

    p = [torch.nn.Parameter(torch.zeros(4))]
    for cls in (fused.Adam, fused.AdamW, fused.RAdam):
        with pytest.raises(TypeError):
            cls(p, lr=1e-3, maximize=True)
        with pytest.raises(TypeError):
            cls(p, lr=1e-3, nesterov=True)
        cls(p, lr=1e-3, foreach=None, capturable=False)
