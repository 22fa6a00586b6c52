"""GPU parity of the one-launch MelGAN residual stack (csrc/resstack.hip) against ATen on CPU
(layers/residual_stack.py:45-85 of the reference: [LeakyReLU, ReflectionPad1d(d), Conv1d(3, dilation=d), LeakyReLU,
Conv1d(1)](c) + Conv1d(1)(c))."""
import pytest
import torch
import torch.nn.functional as F

from parallelwavegan_amd import ops
from tests.util import poison_empty, poison_lds

pytestmark = pytest.mark.gpu


def _reference(x, w1, b1, w2, b2, ws, bs, d, slope):
    x, w1, w2, ws = (t.detach().cpu().double() for t in (x, w1, w2, ws))
    b1, b2, bs = (None if t is None else t.detach().cpu().double() for t in (b1, b2, bs))
    h = F.conv1d(F.pad(F.leaky_relu(x, slope), (d, d), mode="reflect"), w1, b1, dilation=d)
    y = F.conv1d(F.leaky_relu(h, slope), w2, b2) + F.conv1d(x, ws, bs)
    return y, h


@pytest.mark.parametrize("C,B,T,d", [(48, 2, 4096, 1), (48, 3, 200, 27), (48, 1, 64, 9), (96, 2, 2048, 3), (96, 1, 68, 27),
                                     (96, 3, 320, 9), (192, 2, 512, 27), (192, 1, 128, 1), (192, 2, 76, 3)])
@pytest.mark.parametrize("with_bias,with_scale", [(True, True), (False, False)])
def test_resstack_kernel_matches_aten(C, B, T, d, with_bias, with_scale, device):
    torch.manual_seed(C + T + d)
    slope = 0.2
    x = torch.randn(B, C, T, device=device)
    w1 = torch.randn(C, C, 3, device=device) / (3 * C) ** 0.5
    w2 = torch.randn(C, C, 1, device=device) / C ** 0.5
    ws = torch.randn(C, C, 1, device=device) / C ** 0.5
    b1, b2, bs = ((torch.randn(C, device=device) if with_bias else None) for _ in range(3))
    s1, s2, ss = ((torch.rand(C, device=device) + 0.5 if with_scale else None) for _ in range(3))
    assert ops.resstack_supported(C, T, d)
    with poison_lds(), poison_empty():
        img = ops.resstack_pack_weight(w1, s1, w2, s2, ws, ss)
        y, h = ops.resstack_forward(x, img, d, slope, b1, b2, bs, save_h=True)
        y2, h2 = ops.resstack_forward(x, img, d, slope, b1, b2, bs, save_h=False)
    assert h2 is None and torch.equal(y, y2)

    def eff(w, s):
        return w if s is None else w * s.view(-1, 1, 1)

    want_y, want_h = _reference(x, eff(w1, s1), b1, eff(w2, s2), b2, eff(ws, ss), bs, d, slope)
    eh = (h.cpu().double() - want_h).abs().max().item() / want_h.abs().max().item()
    ey = (y.cpu().double() - want_y).abs().max().item() / want_y.abs().max().item()
    assert eh <= 2e-6 and ey <= 3e-6, (eh, ey)


@pytest.mark.parametrize("C,T,d", [(48, 512, 3), (96, 256, 9), (192, 128, 27)])
def test_resstack_kernels_on_a_base_that_is_only_4_byte_aligned(C, T, d, device):
    """the forward demanded a 16-B aligned x while the data-gradient kernel issued the same 16-B LDS-DMA at
    8-B aligned offsets.  The instruction is legal at 4-byte source alignment (tools/probes/glds_x4.hip): both kernels
    give bit-identical results on the same values placed 4, 8 and 12 bytes past a 16-B boundary."""
    torch.manual_seed(C + d)
    slope, B = 0.2, 2
    w1 = torch.randn(C, C, 3, device=device) / (3 * C) ** 0.5
    w2 = torch.randn(C, C, 1, device=device) / C ** 0.5
    ws = torch.randn(C, C, 1, device=device) / C ** 0.5
    b1 = torch.randn(C, device=device)
    img = ops.resstack_pack_weight(w1, None, w2, None, ws, None)
    imgb = ops.resstack_pack_weight_bwd(w1, None, w2, None, ws, None)
    n = B * C * T
    base_x, base_dy = torch.randn(n + 8, device=device), torch.randn(n + 8, device=device)
    outs = []
    for shift in (0, 1, 2, 3):
        x = torch.empty(n + 8, device=device)[shift:shift + n].view(B, C, T)
        dy = torch.empty(n + 8, device=device)[shift:shift + n].view(B, C, T)
        x.copy_(base_x[:n].view(B, C, T))
        dy.copy_(base_dy[:n].view(B, C, T))
        assert x.data_ptr() % 16 == 4 * shift
        with poison_lds():
            y, h = ops.resstack_forward(x, img, d, slope, b1, None, None, save_h=True)
            dh, dxp = ops.resstack_backward_data(dy, h, x, imgb, d, slope)
        outs.append((y.clone(), h.clone(), dh.clone(), dxp.clone()))
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(o, outs[0]))


def test_resstack_unsupported_geometries():
    assert not ops.resstack_supported(64, 1024, 3)
    assert not ops.resstack_supported(96, 1022, 3)  # T % 4
    assert not ops.resstack_supported(96, 1024, 81)
    assert not ops.resstack_supported(48, 32, 1)


@pytest.mark.parametrize("C,T,d,weight_norm", [(96, 256, 9, True), (48, 512, 3, False), (192, 128, 27, True)])
def test_residual_stack_layer_with_and_without_the_one_launch_unit(C, T, d, weight_norm, device):
    """layers.ResidualStack: the one-launch forward (+ the three layers' own backward through precomputed autograd
    nodes) against the three-launch path: output, input gradient, every parameter gradient; no-grad forward too."""
    from parallelwavegan_amd.layers.residual_stack import ResidualStack

    torch.manual_seed(5)
    blk = ResidualStack(channels=C, dilation=d).to(device)
    if weight_norm:
        for cv in (blk.stack[2], blk.stack[4], blk.skip_layer):
            cv.apply_weight_norm()
    x0 = torch.randn(2, C, T, device=device)
    w = torch.randn(2, C, T, device=device)
    res = {}
    for fused in (False, True, "bwd"):
        blk.fuse_unit = bool(fused)
        blk.fuse_unit_backward = fused == "bwd"  # True: one-launch forward + the three layers' own backward nodes
        blk.zero_grad()
        x = x0.clone().requires_grad_()
        with poison_lds(), poison_empty():
            y = blk(x)
            (y * w).sum().backward()
            with torch.no_grad():
                y_ng = blk(x0)
        res[fused] = dict(y=y.detach(), y_ng=y_ng, dx=x.grad, **{n: p.grad.clone() for n, p in blk.named_parameters()})
    for mode in (True, "bwd"):
        assert set(res[mode]) == set(res[False])
        for k, want in res[False].items():
            got = res[mode][k]
            err = (got - want).abs().max().item() / (want.abs().max().item() + 1e-12)
            assert err <= 2e-5, (mode, k, err)


@pytest.mark.parametrize("C,B,T,d", [(48, 2, 4096, 1), (48, 2, 200, 27), (48, 1, 64, 9), (96, 2, 2048, 3), (96, 1, 68, 27),
                                     (96, 3, 320, 9), (192, 2, 512, 27), (192, 1, 128, 1), (192, 2, 76, 3)])
def test_resstack_data_gradient_kernel_matches_float64(C, B, T, d, device):
    """dh and dx of the one-launch data gradient against float64 ATen on the CPU (adjoint of the dilated convolution
    via conv_transpose1d, adjoint of the reflection via autograd), with the LeakyReLU masks taken from the same h / x."""
    torch.manual_seed(C + T + 3 * d)
    slope = 0.2
    x = torch.randn(B, C, T, device=device)
    dy = torch.randn(B, C, T, device=device)
    w1 = torch.randn(C, C, 3, device=device) / (3 * C) ** 0.5
    w2 = torch.randn(C, C, 1, device=device) / C ** 0.5
    ws = torch.randn(C, C, 1, device=device) / C ** 0.5
    s1, s2, ss = (torch.rand(C, device=device) + 0.5 for _ in range(3))
    b1 = torch.randn(C, device=device)
    with poison_lds(), poison_empty():
        img = ops.resstack_pack_weight(w1, s1, w2, s2, ws, ss)
        _, h = ops.resstack_forward(x, img, d, slope, b1, None, None, save_h=True)
        imgb = ops.resstack_pack_weight_bwd(w1, s1, w2, s2, ws, ss)
        dh, dxp = ops.resstack_backward_data(dy, h, x, imgb, d, slope)
        dh2, dxp2 = ops.resstack_backward_data(dy, h, x, imgb, d, slope)
    assert torch.equal(dh, dh2) and torch.equal(dxp, dxp2)
    e1, e2, es = ((w * s.view(-1, 1, 1)).cpu().double() for w, s in ((w1, s1), (w2, s2), (ws, ss)))
    xc, hc, dyc = x.cpu().double(), h.cpu().double(), dy.cpu().double()
    want_dh = F.conv_transpose1d(dyc, e2) * torch.where(hc > 0, 1.0, slope)
    xv = xc.clone().requires_grad_()
    xp = F.pad(xv, (d, d), mode="reflect")
    want_dxp = F.conv_transpose1d(want_dh, e1, dilation=d) * torch.where(xp.detach() > 0, 1.0, slope) \
        + F.pad(F.conv_transpose1d(dyc, es), (d, d))
    (want_dx,) = torch.autograd.grad(xp, xv, grad_outputs=want_dxp)
    assert tuple(dxp.shape) == (B, C, T + 2 * d)
    got_dxp = dxp.cpu().double()
    xg = x.clone().requires_grad_()
    (got_dx,) = torch.autograd.grad(F.pad(xg, (d, d), mode="reflect"), xg, grad_outputs=dxp)
    for name, got, want in (("dh", dh.cpu().double(), want_dh), ("dxp", got_dxp, want_dxp), ("dx", got_dx.cpu().double(), want_dx)):
        err = (got - want).abs().max().item() / want.abs().max().item()
        assert err <= 3e-6, (name, err)
