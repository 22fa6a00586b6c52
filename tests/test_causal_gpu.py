"""GPU parity of the use_causal_conv=True model variants and the residual PWG discriminator
 against the reference's golden outputs, plus the reference's own causality property
(test_hifigan.py:198-224, test_melgan.py:275-301, test_parallel_wavegan.py:314-358): the output up
to time t must not change when the input after t changes.  Gradients are checked against autograd
through the oracle restatement."""
import pytest
import torch

from oracle import torch_cpu
from parallelwavegan_amd import layers, models
from tests.golden import synth
from tests.util import WAVE_TOL, load_golden, max_abs, synth_for

pytestmark = pytest.mark.gpu


def _load(cls, cfg, seed, g_scale, device, train=False):
    m = cls(**cfg)
    sd = synth_for(m, seed, g_scale)
    m.load_state_dict(sd)
    m = m.to(device)
    return (m.train() if train else m.eval()), sd


def test_causal_generators_match_reference_golden(device):
    gold = load_golden("causal_variants")
    seed = int(gold["meta"][0])
    with torch.no_grad():
        g, _ = _load(models.HiFiGANGenerator, synth.HIFIGAN_CAUSAL, seed, float(gold["g_scale"]), device)
        assert max_abs(g(synth.synth_input("c", (2, 80, 24), seed=seed).to(device)), gold["hifigan"]) <= WAVE_TOL
        m, _ = _load(models.MelGANGenerator, synth.MELGAN_CAUSAL, seed + 1, synth.MELGAN_G_SCALE, device)
        assert max_abs(m(synth.synth_input("c", (2, 80, 20), seed=seed + 1).to(device)), gold["melgan"]) <= WAVE_TOL
        p, _ = _load(models.ParallelWaveGANGenerator, synth.PWG_CAUSAL, seed + 2, 1.0, device)
        z = synth.synth_input("z", (2, 1, 18 * 16), seed=seed + 2)
        c = synth.synth_input("c", (2, 80, 18 + 4), seed=seed + 2)
        assert max_abs(p(z.to(device), c.to(device)), gold["pwg"]) <= WAVE_TOL


@pytest.mark.parametrize("causal", [False, True])
def test_residual_pwg_discriminator_matches_reference_golden(causal, device):
    gold = load_golden("causal_variants")
    seed = int(gold["meta"][0])
    d, _ = _load(models.ResidualParallelWaveGANDiscriminator, dict(use_causal_conv=causal, **synth.RESIDUAL_PWG_D),
                 seed + 3, 1.0, device)
    x = 0.5 * synth.synth_input("wave", (2, 1, 700), seed=seed + 3)
    with torch.no_grad():
        assert max_abs(d(x.to(device)), gold["res_d_causal" if causal else "res_d"]) <= 3e-5


def test_causal_property_exact(device):
    """Changing the second half of the input leaves the first half of the output bit-identical."""
    torch.manual_seed(0)
    with torch.no_grad():
        g = models.HiFiGANGenerator(**synth.HIFIGAN_CAUSAL).to(device).eval()
        c1 = torch.randn(1, 80, 20, device=device)
        c2 = c1.clone()
        c2[..., 10:] = torch.randn(1, 80, 10, device=device)
        up = g.upsample_factor
        y1, y2 = g(c1), g(c2)
        assert torch.equal(y1[..., : 10 * up], y2[..., : 10 * up])
        assert not torch.equal(y1[..., 10 * up:], y2[..., 10 * up:])
        m = models.MelGANGenerator(**synth.MELGAN_CAUSAL).to(device).eval()
        y1, y2 = m(c1), m(c2)
        assert torch.equal(y1[..., : 10 * m.upsample_factor], y2[..., : 10 * m.upsample_factor])
        d = models.ResidualParallelWaveGANDiscriminator(use_causal_conv=True, **synth.RESIDUAL_PWG_D).to(device).eval()
        x1 = torch.randn(2, 1, 400, device=device)
        x2 = x1.clone()
        x2[..., 200:] = torch.randn(2, 1, 200, device=device)
        assert torch.equal(d(x1)[..., :200], d(x2)[..., :200])
        # PWG: noise and conditioning both change after the midpoint
        p = models.ParallelWaveGANGenerator(**synth.PWG_CAUSAL).to(device).eval()
        z1 = torch.randn(1, 1, 16 * 16, device=device)
        a1 = torch.randn(1, 80, 16 + 4, device=device)
        z2, a2 = z1.clone(), a1.clone()
        z2[..., 128:] = torch.randn(1, 1, 128, device=device)
        a2[..., 8 + 2:] = torch.randn(1, 80, 10, device=device)   # frames >= 8 (after the 2 context frames)
        assert torch.equal(p(z1, a1)[..., :128], p(z2, a2)[..., :128])


def test_causal_layers_match_oracle_with_gradients(device):
    """CausalConv1d / CausalConvTranspose1d forward + input/weight gradients vs autograd on the oracle."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 24, 50, generator=g, requires_grad=True)
    conv = layers.CausalConv1d(24, 40, 5, dilation=3)
    w, b = conv.conv.weight.detach().clone().requires_grad_(), conv.conv.bias.detach().clone().requires_grad_()
    y_ref = torch_cpu.causal_conv1d(torch.nn.functional.leaky_relu(x, 0.1), w, b, 3)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    conv = conv.to(device)
    xd = x.detach().to(device).requires_grad_()
    y = conv(xd, pre_act="leaky_relu", pre_slope=0.1)
    y.backward(dy.to(device))
    assert max_abs(y, y_ref) < 3e-5
    assert max_abs(xd.grad, x.grad) < 3e-5
    assert max_abs(conv.conv.weight.grad, w.grad) < 1e-4
    assert max_abs(conv.conv.bias.grad, b.grad) < 1e-4

    x = torch.randn(2, 24, 30, generator=g, requires_grad=True)
    up = layers.CausalConvTranspose1d(24, 12, 8, 4)
    w, b = up.deconv.weight.detach().clone().requires_grad_(), up.deconv.bias.detach().clone().requires_grad_()
    y_ref = torch_cpu.causal_conv_transpose1d(x, w, b, 4)
    assert y_ref.shape[-1] == 30 * 4
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    up = up.to(device)
    xd = x.detach().to(device).requires_grad_()
    y = up(xd)
    y.backward(dy.to(device))
    assert max_abs(y, y_ref) < 3e-5
    assert max_abs(xd.grad, x.grad) < 3e-5
    assert max_abs(up.deconv.weight.grad, w.grad) < 1e-4
