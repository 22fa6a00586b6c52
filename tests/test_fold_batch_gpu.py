"""GPU: batch folding of weight-heavy, few-column Conv1d layers (layers/conv.py: _ConvNd._fold_batch) -- the tail of
the HiFi-GAN scale discriminator (reference: /root/reference/parallel_wavegan/models/hifigan.py:529-601) run as one
item of width B.  Forward, data gradient and every parameter gradient against the same ATen ops on CPU (fp32, what
the reference executes at these call sites), and against the unfolded launch of the same layer."""
import pytest
import torch
import torch.nn.functional as F

from parallelwavegan_amd.layers.conv import Conv1d, _ConvNd

RTOL = 3e-5  # relative to the largest reference magnitude (fp32 summation order only)


def _close(a, b, what, rtol=RTOL):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)
    assert err <= rtol, f"{what}: rel-to-max error {err:.3e}"


CASES = [
    # B, Cin, Cout, T, K, stride, pad, groups, norm, pre-activation slope
    (16, 1024, 1024, 32, 5, 1, 2, 1, "weight", 0.1),     # k = 5 tail layer, first scale
    (16, 1024, 1024, 9, 5, 1, 2, 1, "weight", 0.1),      # ... third scale (T = 2049 / 256)
    (16, 1024, 1024, 17, 41, 1, 20, 16, "weight", 0.1),  # grouped k = 41, stride 1
    (16, 512, 1024, 65, 41, 4, 20, 16, "weight", 0.1),   # grouped k = 41, stride 4 (T 65 -> 17)
    (16, 1024, 1024, 32, 5, 1, 2, 1, "spectral", 0.1),   # the first scale runs under spectral norm
    (5, 1024, 1024, 12, 5, 1, 2, 1, None, None),         # odd batch, no norm, no activation
]


def _layer(cin, cout, k, stride, pad, groups, norm, device, seed):
    torch.manual_seed(seed)
    m = Conv1d(cin, cout, k, stride=stride, padding=pad, groups=groups)
    if norm == "weight":
        m.apply_weight_norm()
        with torch.no_grad():
            m.weight_g.mul_(torch.rand_like(m.weight_g) + 0.5)
    elif norm == "spectral":
        m.apply_spectral_norm()
    return m.to(device)


@pytest.mark.gpu
@pytest.mark.parametrize("B,Cin,Cout,T,K,stride,pad,groups,norm,slope", CASES)
def test_folded_layer_matches_aten_and_the_unfolded_launch(B, Cin, Cout, T, K, stride, pad, groups, norm, slope,
                                                           device, monkeypatch):
    m = _layer(Cin, Cout, K, stride, pad, groups, norm, device, seed=B + T + K)
    m.eval()  # (spectral norm: no power iteration, so that the two runs see the same weight)
    g = torch.Generator().manual_seed(T * 7 + K)
    x_cpu = torch.randn(B, Cin, T, generator=g)
    fused = dict(pre_act="leaky_relu", pre_slope=slope) if slope is not None else {}
    runs = {}
    for fold in (False, True):
        monkeypatch.setattr(_ConvNd, "fold_batch", fold)
        assert m._fold_batch(x_cpu, None, None, None) == fold
        for p in m.parameters():
            p.grad = None
        x = x_cpu.to(device).requires_grad_()
        y = m(x, **fused)
        if "dy" not in runs:
            runs["dy"] = torch.randn(y.shape, generator=g)
        y.backward(runs["dy"].to(device))
        with torch.no_grad():
            y_nograd = m(x.detach(), **fused)
        assert y.is_contiguous() and x.grad.is_contiguous() and tuple(y.shape) == (B, Cout, m.out_length(T))
        runs[fold] = dict(y=y.detach(), y_nograd=y_nograd, dx=x.grad.detach(),
                          **{n: p.grad.detach().clone() for n, p in m.named_parameters()})
    # the oracle: ATen on CPU with the layer's effective weight
    w = m.effective_weight().cpu().reshape(Cout, Cin // groups, K).requires_grad_()
    b = m.bias.detach().cpu().requires_grad_()
    xr = x_cpu.clone().requires_grad_()
    y_ref = F.conv1d(F.leaky_relu(xr, slope) if slope is not None else xr, w, b, stride=stride, padding=pad, groups=groups)
    y_ref.backward(runs["dy"])
    for fold in (False, True):
        r = runs[fold]
        _close(r["y"], y_ref, f"forward (fold={fold})")
        _close(r["y_nograd"], y_ref, f"no-grad forward (fold={fold})")
        _close(r["dx"], xr.grad, f"data gradient (fold={fold})")
        _close(r["bias"], b.grad, f"bias gradient (fold={fold})")
        if norm is None:
            _close(r["weight"], w.grad.reshape(r["weight"].shape), f"weight gradient (fold={fold})")
    # parameter gradients of the normalised layers: folded == unfolded (their oracle comparison is the unfolded
    # path's own test, tests/test_conv_ops_gpu.py / test_discriminator_gpu.py)
    for name in runs[True]:
        _close(runs[True][name], runs[False][name], f"{name}: folded vs unfolded", rtol=1e-4)


def test_fold_rule_leaves_the_other_layers_alone(monkeypatch):
    monkeypatch.setattr(_ConvNd, "fold_batch", True)
    big = Conv1d(1024, 1024, 5, padding=2)
    assert big._fold_batch(torch.zeros(16, 1024, 32), None, None, None)
    assert not big._fold_batch(torch.zeros(16, 1024, 128), None, None, None)      # enough columns per item
    assert not big._fold_batch(torch.zeros(2, 1024, 32), None, None, None)        # too few items
    assert not big._fold_batch(torch.zeros(16, 1024, 32), torch.zeros(1), None, None)  # fused addend
    small = Conv1d(128, 128, 5, padding=2)
    assert not small._fold_batch(torch.zeros(16, 128, 32), None, None, None)      # small weight
    refl = Conv1d(1024, 1024, 5, padding=2, pad_mode="reflect")
    assert not refl._fold_batch(torch.zeros(16, 1024, 32), None, None, None)      # reflect padding needs width 1


# geometries the rule would also fold in other models: dilation, causal (left-only) padding, stride 2 / 3, groups,
# bias-free layers, post-activation -- run with the weight threshold lowered so that small layers fold
FUZZ = [
    # B, Cin, Cout, T, K, stride, dilation, padding, groups, bias, post_act
    (8, 48, 64, 23, 3, 1, 1, 1, 1, True, None),
    (4, 64, 64, 40, 7, 1, 3, 9, 1, True, "leaky_relu"),       # dilation 3
    (6, 32, 96, 31, 5, 2, 1, 2, 1, False, None),              # stride 2, no bias
    (7, 64, 32, 37, 4, 3, 1, (3, 0), 1, True, None),          # causal padding, stride 3, even kernel
    (16, 64, 64, 9, 9, 1, 2, 8, 4, True, "tanh"),             # groups 4, dilation 2, T < receptive field
    (5, 40, 40, 12, 1, 1, 1, 0, 1, True, None),               # 1 x 1
    (4, 64, 128, 33, 11, 1, 1, (10, 0), 2, True, "leaky_relu"),  # causal, groups 2
    (9, 16, 16, 1, 3, 1, 1, 1, 1, True, None),                # a single column per item
]


@pytest.mark.gpu
@pytest.mark.parametrize("B,Cin,Cout,T,K,stride,dil,pad,groups,bias,post", FUZZ)
def test_folding_is_exact_for_every_conv1d_geometry(B, Cin, Cout, T, K, stride, dil, pad, groups, bias, post, device,
                                                    monkeypatch):
    monkeypatch.setattr(_ConvNd, "fold_min_weight_bytes", 0)
    torch.manual_seed(B * 31 + T)
    m = Conv1d(Cin, Cout, K, stride=stride, padding=pad, dilation=dil, groups=groups, bias=bias).to(device)
    g = torch.Generator().manual_seed(T + K)
    x_cpu = torch.randn(B, Cin, T, generator=g)
    fused = dict(pre_act="leaky_relu", pre_slope=0.2)
    if post is not None:
        fused.update(post_act=post, post_slope=0.1)
    runs = {}
    for fold in (False, True):
        monkeypatch.setattr(_ConvNd, "fold_batch", fold)
        assert m._fold_batch(x_cpu, None, None, None) == fold
        for p in m.parameters():
            p.grad = None
        x = x_cpu.to(device).requires_grad_()
        y = m(x, **fused)
        if "dy" not in runs:
            runs["dy"] = torch.randn(y.shape, generator=g)
        y.backward(runs["dy"].to(device))
        with torch.no_grad():
            y_nograd = m(x.detach(), **fused)
        runs[fold] = dict(y=y.detach(), y_nograd=y_nograd, dx=x.grad.detach(),
                          **{n: p.grad.detach().clone() for n, p in m.named_parameters()})
    # oracle: ATen on CPU
    pl, pr = (pad if isinstance(pad, tuple) else (pad, pad))
    w = m.weight.detach().cpu().requires_grad_()
    b = m.bias.detach().cpu().requires_grad_() if bias else None
    xr = x_cpu.clone().requires_grad_()
    y_ref = F.conv1d(F.pad(F.leaky_relu(xr, 0.2), (pl, pr)), w, b, stride=stride, dilation=dil, groups=groups)
    if post == "leaky_relu":
        y_ref = F.leaky_relu(y_ref, 0.1)
    elif post == "tanh":
        y_ref = torch.tanh(y_ref)
    y_ref.backward(runs["dy"])
    for fold in (False, True):
        r = runs[fold]
        _close(r["y"], y_ref, f"forward (fold={fold})")
        _close(r["y_nograd"], y_ref, f"no-grad forward (fold={fold})")
        _close(r["dx"], xr.grad, f"data gradient (fold={fold})")
        _close(r["weight"], w.grad, f"weight gradient (fold={fold})")
        if bias:
            _close(r["bias"], b.grad, f"bias gradient (fold={fold})")
