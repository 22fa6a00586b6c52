"""GPU parity of the Unet-based HiFi-GAN generator vs the reference golden output (eval
mode), gradients vs autograd on the oracle, and the dropout / channel-concatenation kernels."""
import pytest
import torch

from oracle import torch_cpu
from parallelwavegan_amd import functional as Fn
from parallelwavegan_amd import models
from tests.golden import synth
from tests.util import WAVE_TOL, load_golden, max_abs, synth_for

pytestmark = pytest.mark.gpu


def test_generator_matches_reference_golden_and_oracle_gradients(device):
    gold = load_golden("uhifigan")
    seed = int(gold["meta"][0])
    g = models.UHiFiGANGenerator(**synth.UHIFIGAN_TINY)
    sd = synth_for(g, seed, float(gold["g_scale"]))
    g.load_state_dict(sd)
    g = g.to(device).eval()
    c = synth.synth_input("c", (2, 80, 24), seed=seed)
    e = synth.synth_input("excitation", (2, 1, 24 * 8), seed=seed)
    with torch.no_grad():
        assert max_abs(g(c.to(device), None, e.to(device)), gold["y"]) <= WAVE_TOL
        y_inf = g.inference(excitation=e[0].reshape(-1, 1).numpy(), c=c[0].transpose(0, 1).numpy())
        assert max_abs(y_inf.transpose(0, 1), gold["y"][0]) <= WAVE_TOL
    # gradients (eval mode: dropout off on both sides)
    sd_ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    y_ref = torch_cpu.uhifigan_generator(sd_ref, c, e, **synth.UHIFIGAN_TINY)
    w = synth.synth_input("w", tuple(y_ref.shape), seed=seed + 1)
    (y_ref * w).sum().backward()
    y = g(c.to(device), None, e.to(device))
    (y * w.to(device)).sum().backward()
    n = 0
    for name, p in g.named_parameters():
        ref = sd_ref[name].grad
        assert max_abs(p.grad, ref) <= 2e-4 * (ref.abs().max().item() + 1e-8) + 1e-7, name
        n += 1
    assert n > 60


def test_dropout_and_concat_kernels(device):
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(4, 16, 1000, generator=gen).to(device).requires_grad_()
    y = Fn.DropoutFn.apply(x, 0.3, 12345)
    keep = (y != 0)
    frac = keep.float().mean().item()
    assert abs(frac - 0.7) < 0.01                                  # keep probability 1 - p
    assert torch.allclose(y[keep], x.detach()[keep] / 0.7)          # inverted-dropout scaling
    assert torch.equal(Fn.DropoutFn.apply(x, 0.3, 12345), y)        # same seed, same mask
    assert not torch.equal(Fn.DropoutFn.apply(x, 0.3, 12346) != 0, keep)
    y.backward(torch.ones_like(y))
    assert torch.allclose(x.grad, keep.float() / 0.7)               # backward applies the same mask
    a = torch.randn(2, 5, 33, generator=gen).to(device).requires_grad_()
    b = torch.randn(2, 7, 33, generator=gen).to(device).requires_grad_()
    cat = Fn.ConcatChannelsFn.apply(a, b)
    assert torch.equal(cat, torch.cat((a, b), dim=1))
    wgt = torch.randn(cat.shape, generator=gen).to(device)
    (cat * wgt).sum().backward()
    assert torch.equal(a.grad, wgt[:, :5]) and torch.equal(b.grad, wgt[:, 5:])
    # training mode: dropout active inside the model, output differs from eval and stays finite
    g = models.UHiFiGANGenerator(**synth.UHIFIGAN_TINY).to(device)
    c = torch.randn(1, 80, 16, generator=gen).to(device)
    e = torch.randn(1, 1, 128, generator=gen).to(device)
    with torch.no_grad():
        y_eval = g.eval()(c, None, e)
        y_train = g.train()(c, None, e)
        y_train2 = g(c, None, e)
    assert torch.isfinite(y_train).all() and not torch.equal(y_eval, y_train)
    assert not torch.equal(y_train, y_train2)  # a new mask per call
    # the mask also changes between replays of a captured hipGraph (device-resident seed counter)
    from parallelwavegan_amd.models.uhifigan import _Dropout

    drop = _Dropout(0.5).train()
    xs = torch.ones(4096, device=device)
    drop(xs)  # creates the counter outside the capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ys = drop(xs)
    graph.replay()
    m1 = (ys != 0).clone()
    graph.replay()
    m2 = (ys != 0).clone()
    assert 0.4 < m1.float().mean().item() < 0.6 and not torch.equal(m1, m2)
