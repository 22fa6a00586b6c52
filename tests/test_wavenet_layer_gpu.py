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
    # inference path (no grad): same values; the caller's skip tensor is left alone unless the caller hands the buffer
    # over (``inplace_skips``, what the generator's own loop does) -- then the running sum is accumulated in place
    blk.fuse_layer = True
    with torch.no_grad():
        s_in = None if s0 is None else s0.clone()
        xo, so = blk(x0, c0, skips=s_in, skip_scale=skip_scale)
        assert s_in is None or (so.data_ptr() != s_in.data_ptr() and torch.equal(s_in, s0))
        assert _rel(xo, res[False]["xo"]) <= 3e-5 and _rel(so, res[False]["so"]) <= 3e-5
        xo2, so2 = blk(x0, c0, skips=s_in, skip_scale=skip_scale, inplace_skips=True)
    assert torch.equal(xo2, xo) and torch.equal(so2, so)
    assert s_in is None or so2.data_ptr() == s_in.data_ptr()


@pytest.mark.parametrize("B,T,dil,with_go", [(3, 4133, 27, True), (1, 64, 1, True), (2, 9000, 512, False), (6, 25600, 4, True)])
def test_weight_backward_kernel_matches_float64(B, T, dil, with_go, device):
    """csrc/wavenet.hip weight path against a float64 contraction of the same operands (every tail: T not a
    multiple of the 64-column chunk, taps reaching over both ends, the residual output unused)."""
    torch.manual_seed(11)
    desc = ops.make_wavenet_desc(B, T, dil)
    dz = torch.randn(B, 128, T, device=device)
    x = torch.randn(B, 64, T, device=device)
    c = torch.randn(B, 80, T, device=device)
    gs = torch.randn(B, 64, T, device=device)
    go = torch.randn(B, 64, T, device=device) if with_go else None
    g = torch.randn(B, 64, T, device=device)
    flat = lambda outs: [t for o in outs for t in o]  # noqa: E731
    with poison_lds(), poison_empty():
        outs = ops.wavenet_weight_backward(desc, dz, x, c, gs, go, g)
        again = ops.wavenet_weight_backward(desc, dz, x, c, gs, go, g)
    (dwd, _, dbd), (dwa, _, dba), (dws, _, dbs), (dwo, _, dbo) = outs
    assert dba is None
    f = lambda t: t.double()  # noqa: E731
    xp = torch.nn.functional.pad(f(x), (dil, dil))
    want_d = torch.stack([torch.einsum("bmt,bct->mc", f(dz), xp[:, :, k * dil:k * dil + T]) for k in range(3)], dim=2)
    want = dict(dwd=want_d, dwa=torch.einsum("bmt,bct->mc", f(dz), f(c))[:, :, None], dbd=f(dz).sum((0, 2)),
                dws=torch.einsum("bmt,bct->mc", f(gs), f(g))[:, :, None], dbs=f(gs).sum((0, 2)))
    got = dict(dwd=dwd, dwa=dwa, dbd=dbd, dws=dws, dbs=dbs)
    if with_go:
        want.update(dwo=torch.einsum("bmt,bct->mc", f(go), f(g))[:, :, None], dbo=f(go).sum((0, 2)))
        got.update(dwo=dwo, dbo=dbo)
    else:
        assert dwo is None and dbo is None
    for k in want:
        assert got[k].shape == want[k].shape, k
        assert torch.isfinite(got[k]).all(), k
        assert _rel(got[k].double(), want[k]) <= 2e-5, (k, _rel(got[k].double(), want[k]))
    for a, b in zip(flat(outs), flat(again)):  # fixed slices, ordered sums: bit-identical
        assert (a is None and b is None) or torch.equal(a, b)
    # weight-normalised form: (dv, dg) = torch's autograd through w = g v / |v| applied to the plain gradients
    vs = [torch.randn(s, device=device) for s in ((128, 64, 3), (128, 80, 1), (64, 64, 1), (64, 64, 1))]
    gm = [torch.rand(v.shape[0], device=device) + 0.5 for v in vs]
    with poison_lds(), poison_empty():
        outs_wn = ops.wavenet_weight_backward(desc, dz, x, c, gs, go, g,
                                              convs=[(v, gg, i != 1) for i, (v, gg) in enumerate(zip(vs, gm))])
    for i, (v, gg) in enumerate(zip(vs, gm)):
        if outs[i][0] is None:
            assert outs_wn[i] == (None, None, None)
            continue
        v64, g64 = f(v).requires_grad_(), f(gg).requires_grad_()
        w = g64[:, None, None] * v64 / v64.flatten(1).norm(dim=1)[:, None, None]
        w.backward(f(outs[i][0]))
        assert _rel(outs_wn[i][0].double(), v64.grad) <= 2e-5, i
        assert _rel(outs_wn[i][1].double(), g64.grad) <= 2e-5, i
        assert (outs_wn[i][2] is None) == (outs[i][2] is None)
        assert outs[i][2] is None or torch.equal(outs_wn[i][2], outs[i][2])


def test_fused_layer_backward_without_residual_output(device):
    """The last layer of the generator: only the skip output reaches the loss (dx_out is None)."""
    torch.manual_seed(5)
    blk = WaveNetResidualBlock(dilation=2).to(device)
    for cv in (blk.conv, blk.conv1x1_aux, blk.conv1x1_skip, blk.conv1x1_out):
        cv.apply_weight_norm()
    x0, c0 = torch.randn(2, 64, 700, device=device), torch.randn(2, 80, 700, device=device)
    ws = torch.randn(2, 64, 700, device=device)
    res = {}
    for fused in (True, False):
        blk.fuse_layer = fused
        blk.zero_grad()
        x = x0.clone().requires_grad_()
        with poison_lds(), poison_empty():
            _, so = blk(x, c0, skips=None, skip_scale=1.0)
            (so * ws).sum().backward()
        res[fused] = dict(dx=x.grad, **{n: (None if p.grad is None else p.grad.clone()) for n, p in blk.named_parameters()})
    for k in res[True]:
        a, b = res[True][k], res[False][k]
        assert (a is None) == (b is None) or (a is None and float(b.abs().max()) == 0.0) or (b is None and float(a.abs().max()) == 0.0), k
        if a is not None and b is not None:
            assert _rel(a, b) <= 5e-5, (k, _rel(a, b))


@pytest.mark.parametrize("c_requires_grad", [True, False])
def test_aux_gradient_chained_through_layers(c_requires_grad, device):
    """Three layers share the aux features: with ``chain_aux`` the gradient w.r.t. c is summed along the chain in the
    data-gradient epilogues (models/parallel_wavegan.py forward); same values as autograd's own accumulation."""
    torch.manual_seed(9)
    blks = [WaveNetResidualBlock(dilation=d).to(device) for d in (1, 4, 16)]
    for blk in blks:
        for cv in (blk.conv, blk.conv1x1_aux, blk.conv1x1_skip, blk.conv1x1_out):
            cv.apply_weight_norm()
    B, T = 2, 900
    x0, c0 = torch.randn(B, 64, T, device=device), torch.randn(B, 80, T, device=device)
    w = torch.randn(B, 64, T, device=device)
    res = {}
    for mode in ("chained", "plain", "unfused"):
        for blk in blks:
            blk.fuse_layer = mode != "unfused"
            blk.zero_grad()
        x = x0.clone().requires_grad_()
        c = c0.clone().requires_grad_(c_requires_grad)
        h, skips, cc = x, None, c
        with poison_lds(), poison_empty():
            for i, blk in enumerate(blks):
                scale = 0.5 if i == len(blks) - 1 else 1.0
                if mode == "chained":
                    h, skips, cc = blk(h, cc, skips=skips, skip_scale=scale, chain_aux=True)
                else:
                    h, skips = blk(h, c, skips=skips, skip_scale=scale)
            (skips * w).sum().backward()
        res[mode] = dict(dx=x.grad, dc=c.grad, out=skips.detach(),
                         **{f"{i}.{n}": p.grad.clone() for i, blk in enumerate(blks) for n, p in blk.named_parameters()
                            if p.grad is not None})
    for mode in ("chained", "plain"):
        assert set(res[mode]) == set(res["unfused"])
        for k, want in res["unfused"].items():
            got = res[mode][k]
            assert (got is None) == (want is None), (mode, k)
            if want is not None:
                assert torch.isfinite(got).all(), (mode, k)
                assert _rel(got, want) <= 5e-5, (mode, k, _rel(got, want))
    assert (res["chained"]["dc"] is not None) == c_requires_grad


def test_weight_gradients_on_the_side_stream_equal_the_in_line_ones(device):
    """The layers' weight-gradient launches run on a side stream (functional._join_wgrad_side_stream): same bits as
    in-line launches, (a) joined by the end-of-pass callback (fresh gradients), (b) joined per layer when the pass
    accumulates into existing ``.grad`` tensors, (c) with a post-accumulate hook that reads the gradient."""
    from parallelwavegan_amd import functional

    torch.manual_seed(11)
    blks = [WaveNetResidualBlock(dilation=d).to(device) for d in (1, 2, 4, 8, 16, 32)]
    for blk in blks:
        for cv in (blk.conv, blk.conv1x1_aux, blk.conv1x1_skip, blk.conv1x1_out):
            cv.apply_weight_norm()
    B, T = 3, 4000
    x0, c0 = torch.randn(B, 64, T, device=device), torch.randn(B, 80, T, device=device)
    w = torch.randn(B, 64, T, device=device)
    params = [p for blk in blks for p in blk.parameters()]

    def run(passes):
        for p in params:
            p.grad = None
        with poison_lds(), poison_empty():
            for _ in range(passes):
                h, skips, cc = x0, None, c0
                for blk in blks:
                    h, skips, cc = blk(h, cc, skips=skips, skip_scale=1.0, chain_aux=True)
                (skips * w).sum().backward()
        # (no synchronize: reading .grad on the current stream must already be ordered after the side stream)
        return [None if p.grad is None else p.grad.clone() for p in params]

    saved = functional.WGRAD_SIDE_STREAM
    try:
        functional.WGRAD_SIDE_STREAM = False
        want1, want2 = run(1), run(2)
        functional.WGRAD_SIDE_STREAM = True
        got1, got2 = run(1), run(2)
        seen = []
        # (torch also fires the hook of a parameter whose incoming gradient is undefined -- the unused residual 1x1)
        handles = [p.register_post_accumulate_grad_hook(lambda p: seen.append(p.grad.clone()) if p.grad is not None else None)
                   for p in params]
        got_hook = run(1)
        for h in handles:
            h.remove()
    finally:
        functional.WGRAD_SIDE_STREAM = saved
    assert sum(t is not None for t in want1) >= len(params) - 3  # (the last layer's residual output is unused)
    for got, want in ((got1, want1), (got2, want2), (got_hook, want1)):
        for a, b in zip(got, want):
            assert (a is None) == (b is None)
            assert a is None or torch.equal(a, b)
    assert len(seen) == sum(t is not None for t in want1)
    want_sorted = sorted(float(t.double().abs().sum()) for t in want1 if t is not None)
    assert sorted(float(t.double().abs().sum()) for t in seen) == want_sorted


def test_backward_rejects_parameters_modified_after_the_forward(device):
    """The fused layer's backward reads the weights through the block: like autograd's version check on saved tensors,
    an update between forward and backward must raise instead of mixing old and new values."""
    torch.manual_seed(3)
    blk = WaveNetResidualBlock(dilation=2).to(device)
    for cv in (blk.conv, blk.conv1x1_aux, blk.conv1x1_skip, blk.conv1x1_out):
        cv.apply_weight_norm()
    x, c = torch.randn(1, 64, 256, device=device), torch.randn(1, 80, 256, device=device)
    h, skips = blk(x, c)
    with torch.no_grad():
        blk.conv1x1_skip.weight_g.mul_(1.5)
    with pytest.raises(RuntimeError, match="modified after the forward"):
        (h.sum() + skips.sum()).backward()
