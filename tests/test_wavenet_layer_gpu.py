"""GPU parity of the one-launch WaveNet residual layer (csrc/wavenet.hip) against ATen on CPU
(layers/residual_block.py:102-140 of the reference) and of its autograd node against the un-fused layer."""
import math

import pytest
import torch
import torch.nn.functional as F

from parallelwavegan_amd import ops
from parallelwavegan_amd.layers.residual_block import WaveNetResidualBlock
from tests.util import poison_empty, poison_lds

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)


@pytest.mark.parametrize("B,T,dil,with_skips,skip_mul", [(2, 1000, 1, False, 1.0), (1, 4096, 2, True, 1.0),
                                                          (3, 777, 64, True, math.sqrt(1 / 30)), (2, 2048, 512, True, 1.0),
                                                          (1, 50, 4, True, 1.0), (1, 640, 512, False, 1.0)])
def test_layer_kernel_matches_aten(B, T, dil, with_skips, skip_mul, device):
    g = torch.Generator().manual_seed(T + dil)
    x, c = torch.randn(B, 64, T, generator=g), torch.randn(B, 80, T, generator=g)
    skips = torch.randn(B, 64, T, generator=g) if with_skips else None
    w_d = torch.randn(128, 64, 3, generator=g) / math.sqrt(192)
    w_a = torch.randn(128, 80, 1, generator=g) / math.sqrt(80)
    w_s, w_o = torch.randn(64, 64, 1, generator=g) / 8, torch.randn(64, 64, 1, generator=g) / 8
    b_d, b_s, b_o = torch.randn(128, generator=g), torch.randn(64, generator=g), torch.randn(64, generator=g)
    s_d = 1.0 + 0.2 * torch.rand(128, generator=g)  # weight-norm style row scales folded into the image
    z = F.conv1d(x, w_d * s_d.view(-1, 1, 1), b_d, padding=dil, dilation=dil) + F.conv1d(c, w_a)
    gt = torch.tanh(z[:, :64]) * torch.sigmoid(z[:, 64:])
    s_ref = (F.conv1d(gt, w_s, b_s) + (skips if with_skips else 0.0)) * skip_mul
    x_ref = (F.conv1d(gt, w_o, b_o) + x) * math.sqrt(0.5)
    desc = ops.make_wavenet_desc(B, T, dil, out_mul=math.sqrt(0.5), skip_mul=skip_mul)
    assert ops.wavenet_layer_supported(desc)
    d = lambda t: None if t is None else t.to(device).contiguous()  # noqa: E731
    with poison_lds(), poison_empty():
        img = ops.wavenet_pack_weights(desc, d(w_d), d(s_d), d(w_a), None, d(w_s), None, d(w_o), None)
        x_out, s_out, z_out, g_out = ops.wavenet_layer_forward(desc, d(x), d(c), d(skips), img, d(b_d), d(b_s), d(b_o),
                                                              save=True)
    for name, got, want in (("z", z_out, z), ("g", g_out, gt), ("skips", s_out, s_ref), ("x", x_out, x_ref)):
        assert torch.isfinite(got).all(), name
        assert _rel(got, want) <= 3e-5, (name, _rel(got, want))


@pytest.mark.parametrize("dil,skip_scale,first", [(1, 1.0, True), (8, 1.0, False), (256, math.sqrt(1 / 30), False)])
def test_fused_layer_autograd_matches_unfused(dil, skip_scale, first, device):
    torch.manual_seed(3)
    blk = WaveNetResidualBlock(dilation=dil).to(device)
    for cv in (blk.conv, blk.conv1x1_aux, blk.conv1x1_skip, blk.conv1x1_out):
        cv.apply_weight_norm()
    with torch.no_grad():
        for p in blk.parameters():
            p.add_(0.05 * torch.randn_like(p))
    B, T = 2, 1500
    x0 = torch.randn(B, 64, T, device=device)
    c0 = torch.randn(B, 80, T, device=device)
    s0 = None if first else torch.randn(B, 64, T, device=device)
    wx, ws = torch.randn(B, 64, T, device=device), torch.randn(B, 64, T, device=device)
    res = {}
    for fused in (True, False):
        blk.fuse_layer = fused
        blk.zero_grad()
        x, c = x0.clone().requires_grad_(), c0.clone().requires_grad_()
        s = None if s0 is None else s0.clone().requires_grad_()
        xo, so = blk(x, c, skips=s, skip_scale=skip_scale)
        ((xo * wx).sum() + (so * ws).sum()).backward()
        res[fused] = dict(xo=xo.detach(), so=so.detach(), dx=x.grad, dc=c.grad, ds=None if s is None else s.grad,
                          **{n: p.grad.clone() for n, p in blk.named_parameters()})
    assert set(res[True]) == set(res[False])
    for k in res[True]:
        if res[True][k] is None:
            assert res[False][k] is None
            continue
        assert _rel(res[True][k], res[False][k]) <= 5e-5, (k, _rel(res[True][k], res[False][k]))
    # inference path (no grad): same values, skip sum accumulated in place
    blk.fuse_layer = True
    with torch.no_grad():
        s_in = None if s0 is None else s0.clone()
        xo, so = blk(x0, c0, skips=s_in, skip_scale=skip_scale)
    assert _rel(xo, res[False]["xo"]) <= 3e-5 and _rel(so, res[False]["so"]) <= 3e-5
    assert s_in is None or so.data_ptr() == s_in.data_ptr()
