"""GPU: the reference-side ctypes binding shown in INTEGRATION.md (tools/reference_binding/_pwg_ffi.py -- it
imports nothing from parallelwavegan_amd, only the .so and the header's layout) runs an MRF residual block
and matches the reference's own module semantics (layers/residual_block.py:243-258) on CPU."""
import importlib.util
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_side_binding_runs_an_mrf_block(device):
    spec = importlib.util.spec_from_file_location("_pwg_ffi", os.path.join(ROOT, "tools", "reference_binding", "_pwg_ffi.py"))
    ffi = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ffi)
    g = torch.Generator().manual_seed(0)
    c, t, k, dils = 64, 700, 7, (1, 3, 5)
    x = torch.randn(2, c, t, generator=g)
    params = []
    for _ in dils:
        p = []
        for _ in range(2):
            p += [torch.randn(c, c, k, generator=g) / (c * k) ** 0.5, 1.0 + 0.1 * torch.randn(c, 1, 1, generator=g),
                  0.05 * torch.randn(c, generator=g)]
        params.append(tuple(p))
    # reference semantics on CPU: weight_norm(dim=0), x = convs2(leaky(convs1(leaky(x)))) + x
    ref = x
    for (v1, g1, b1, v2, g2, b2), d in zip(params, dils):
        w1 = g1 * v1 / v1.reshape(c, -1).norm(dim=1).reshape(c, 1, 1)
        w2 = g2 * v2 / v2.reshape(c, -1).norm(dim=1).reshape(c, 1, 1)
        xt = F.conv1d(F.leaky_relu(ref, 0.1), w1, b1, dilation=d, padding=(k - 1) // 2 * d)
        ref = F.conv1d(F.leaky_relu(xt, 0.1), w2, b2, padding=(k - 1) // 2) + ref
    dev_params = [tuple(t_.to(device).contiguous() for t_ in p) for p in params]
    y = ffi.hifigan_residual_block_forward(x.to(device), dev_params, k, dils)
    assert (y.cpu() - ref).abs().max().item() <= 3e-5 * float(ref.abs().max())
    # the same block with one pwg_resunit_forward launch per dilation
    y1 = ffi.hifigan_residual_block_forward_one_launch_per_unit(x.to(device), dev_params, k, dils)
    assert (y1.cpu() - ref).abs().max().item() <= 3e-5 * float(ref.abs().max())
