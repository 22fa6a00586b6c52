"""GPU parity of the grouped k = 41 kernels of csrc/gconv.hip (4 / 8 input channels per group on
v_mfma_f32_16x16x4_f32: the scale discriminators' strided layers, models/melgan.py:318-336 and
models/hifigan.py:516-540) against ATen on CPU: forward with the fused bias / activations, data gradient with the
pre-activation mask and accumulation, weight + bias gradient (plain and through the weight-norm finish)."""
import pytest
import torch
import torch.nn.functional as F

from parallelwavegan_amd import ops

pytestmark = pytest.mark.gpu
RTOL = 3e-5


def _close(a, b, what):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)
    assert err <= RTOL, f"{what}: rel-to-max error {err:.3e}"


CASES = [
    # B, Cin, Cout, groups, T, stride, pre_slope, post_slope
    (3, 16, 64, 4, 1000, 4, None, 0.2),     # MelGAN D layer 1 pattern (4 -> 16 per group)
    (2, 64, 256, 16, 515, 4, 0.2, None),    # layer 2, ragged length
    (2, 256, 512, 64, 260, 4, None, 0.2),   # layer 3: 8 output channels per group
    (2, 32, 64, 4, 333, 2, 0.1, 0.1),       # HiFi-GAN MSD pattern: 8 -> 16 per group, stride 2
    (1, 16, 16, 2, 77, 2, None, None),      # 8 -> 8 per group
    (2, 128, 256, 16, 600, 4, None, 0.1),   # HiFi-GAN MSD layer 2 as the recipe has it: 8 -> 16 per group, stride 4
    (1, 16, 16, 2, 131, 4, 0.2, None),      # 8 -> 8 per group, stride 4
    (2, 8, 32, 2, 41, 4, None, 0.2),        # T_out = 11: a single partial tile
    (5, 4 * 5, 16 * 5, 5, 4096 + 3, 4, None, 0.2),  # several passes per wave, odd group count
]


@pytest.mark.parametrize("B,Cin,Cout,groups,T,stride,pre,post", CASES)
def test_grouped_k41_forward_and_gradients(B, Cin, Cout, groups, T, stride, pre, post, device):
    K, pad = 41, 20
    g = torch.Generator().manual_seed(Cin * 7 + T)
    x = torch.randn(B, Cin, T, generator=g, requires_grad=True)
    w = (torch.randn(Cout, Cin // groups, K, generator=g) / (Cin // groups * K) ** 0.5).requires_grad_()
    b = torch.randn(Cout, generator=g, requires_grad=True)
    xin = F.leaky_relu(x, pre) if pre is not None else x
    pre_out = F.conv1d(xin, w, b, stride=stride, padding=pad, groups=groups)
    y_ref = F.leaky_relu(pre_out, post) if post is not None else pre_out
    dpre = torch.randn(pre_out.shape, generator=g)  # gradient w.r.t. the pre-post-activation sum (what the kernels take)
    pre_out.backward(dpre)
    t_out = y_ref.shape[-1]
    desc = ops.make_conv_desc(B, Cin, Cout, T, t_out, K, stride, 1, pad, groups,
                              pre_act="leaky_relu" if pre is not None else None, pre_slope=pre or 0.0,
                              post_act="leaky_relu" if post is not None else None, post_slope=post or 0.0)
    xd, wd, bd, gd = (t.detach().to(device).contiguous() for t in (x, w, b, dpre))
    y = ops.conv1d_forward(desc, xd, ops.pack_weight(desc, wd), bd)
    _close(y, y_ref, "forward")
    dx = ops.conv1d_backward_data(desc, gd, ops.pack_weight_bwd(desc, wd), xd)
    _close(dx, x.grad, "backward_data")
    acc = torch.randn(B, Cin, T, generator=g)
    dx2 = ops.conv1d_backward_data(desc, gd, ops.pack_weight_bwd(desc, wd), xd, accum=acc.to(device))
    _close(dx2, x.grad + acc, "backward_data + accum")
    dw, db = ops.conv1d_backward_weight(desc, xd, gd, tuple(w.shape))
    _close(dw, w.grad, "backward_weight")
    _close(db, b.grad, "backward_bias")
    dw_only, none = ops.conv1d_backward_weight(desc, xd, gd, tuple(w.shape), need_db=False)
    assert none is None
    _close(dw_only, w.grad, "backward_weight (no bias)")


def test_grouped_k41_weight_norm_finish(device):
    """dv, dg of a weight-normalised grouped layer (the discriminators' layers are weight-normalised) through
    pwg_conv1d_backward_weight_wn: gconv weight gradient + weight-norm backward."""
    B, Cin, Cout, groups, T, stride, K, pad = 2, 64, 256, 16, 700, 4, 41, 20
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(B, Cin, T, generator=gen)
    v = torch.randn(Cout, Cin // groups, K, generator=gen, requires_grad=True)
    gg = (1.0 + 0.1 * torch.randn(Cout, 1, 1, generator=gen)).requires_grad_()
    b = torch.randn(Cout, generator=gen, requires_grad=True)
    w = gg * v / v.flatten(1).norm(dim=1).view(-1, 1, 1)
    y = F.conv1d(x, w, b, stride=stride, padding=pad, groups=groups)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy)
    desc = ops.make_conv_desc(B, Cin, Cout, T, y.shape[-1], K, stride, 1, pad, groups)
    dv, dg, db = ops.conv1d_backward_weight_wn(desc, x.to(device), dy.to(device), v.detach().to(device),
                                               gg.detach().reshape(-1).to(device))
    _close(dv, v.grad, "dv")
    _close(dg.reshape(-1), gg.grad.reshape(-1), "dg")
    _close(db, b.grad, "db")


def test_grouped_k41_is_deterministic_under_poison(device):
    """Two runs are bit-identical (fixed slices, fixed summation order) with the LDS NaN-poisoned and the
    workspaces NaN-filled: every staged element a wave reads was written by its own DMA."""
    from tests.util import poison_empty, poison_lds

    B, Cin, Cout, groups, T = 4, 16, 64, 4, 2050
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(B, Cin, T, generator=gen).to(device)
    w = torch.randn(Cout, 4, 41, generator=gen).to(device)
    t_out = (T + 40 - 41) // 4 + 1
    dy = torch.randn(B, Cout, t_out, generator=gen).to(device)
    desc = ops.make_conv_desc(B, Cin, Cout, T, t_out, 41, 4, 1, 20, groups)
    outs = []
    with poison_lds(), poison_empty():
        for _ in range(2):
            y = ops.conv1d_forward(desc, x, ops.pack_weight(desc, w), None)
            dx = ops.conv1d_backward_data(desc, dy, ops.pack_weight_bwd(desc, w), None)
            dw, db = ops.conv1d_backward_weight(desc, x, dy, tuple(w.shape))
            outs.append((y, dx, dw, db))
    for a, b in zip(*outs):
        assert torch.isfinite(a).all() and torch.equal(a, b)
