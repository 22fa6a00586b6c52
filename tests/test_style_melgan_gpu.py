"""GPU parity of StyleMelGAN: generator (TADE residual blocks: instance norm, nearest
upsampling, modulation, softmax/sigmoid gates -- HIP kernels of libpwgkernels.so) and the
random-window PQMF discriminator vs the reference's golden outputs; kernel-level forward/backward vs
autograd on the torch definitions."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import torch_cpu
from parallelwavegan_amd import functional as Fn
from parallelwavegan_amd import models
from tests.golden import synth
from tests.util import WAVE_TOL, load_golden, max_abs, synth_for

pytestmark = pytest.mark.gpu


def test_generator_matches_reference_golden(device):
    gold = load_golden("style_melgan")
    seed = int(gold["meta"][0])
    with torch.no_grad():
        for key, cfg in (("tiny", synth.STYLE_MELGAN_TINY), ("tiny_sigmoid", synth.STYLE_MELGAN_TINY_SIGMOID)):
            g = models.StyleMelGANGenerator(**cfg)
            g.load_state_dict(synth_for(g, seed, 1.1))
            g = g.to(device).eval()
            z = synth.synth_input("z", (2, cfg["in_channels"], 5), seed=seed).to(device)
            c = synth.synth_input("c", (2, 80, 20), seed=seed).to(device)
            assert max_abs(g(c, z), gold[key]) <= WAVE_TOL, key
        g = models.StyleMelGANGenerator()
        g.load_state_dict(synth_for(g, seed + 1, 0.8))
        g = g.to(device).eval()
        z = synth.synth_input("z", (1, 128, 1), seed=seed + 1).to(device)
        c = synth.synth_input("c", (1, 80, 88), seed=seed + 1).to(device)
        y = g(c, z)
        assert y.shape == (1, 1, 88 * 256)
        assert max_abs(y[..., :4096], gold["default_head"]) <= WAVE_TOL
        # inference API: (T', C) features, replicate-padded to the noise length, audio cut back
        g.remove_weight_norm()
        c2 = synth.synth_input("c", (1, 80, 60), seed=seed + 5)
        z2 = synth.synth_input("z", (1, 128, 1), seed=seed + 5)
        y_inf = g.inference(c2[0].transpose(0, 1).numpy(), z=z2.to(device))
        sd = {k: v.cpu() for k, v in g.state_dict().items()}
        cpad = F.pad(c2, (0, 88 - 60), mode="replicate")
        ref = torch_cpu.style_melgan_generator(sd, cpad, z2)[..., : 60 * 256]
        assert y_inf.shape == (60 * 256, 1)
        assert max_abs(y_inf.transpose(0, 1).unsqueeze(0), ref) <= WAVE_TOL


def test_discriminator_matches_reference_golden(device):
    gold = load_golden("style_melgan")
    seed = int(gold["meta"][0])
    d = models.StyleMelGANDiscriminator(**synth.STYLE_MELGAN_D)
    d.load_state_dict(synth.synth_state_dict(d.state_dict(), seed=seed + 2, g_scale=1.2, skip=synth.PQMF_BUFFERS),
                      strict=False)
    d = d.to(device).eval()
    x = 0.5 * synth.synth_input("wave", (2, 1, 8192), seed=seed + 2)
    np.random.seed(seed)  # the same np.random.randint draws as the reference
    with torch.no_grad():
        outs = d(x.to(device))
    assert len(outs) == 8 and all(len(o) == 7 for o in outs)
    logits = torch.stack([o[-1].cpu() for o in outs])
    assert max_abs(logits, gold["d_logits"]) <= 3e-5


@pytest.mark.parametrize("use_softmax", [True, False])
def test_tade_kernels_forward_backward(use_softmax, device):
    g = torch.Generator().manual_seed(7)
    # instance norm
    x = torch.randn(3, 16, 301, generator=g, requires_grad=True)
    y_ref = F.instance_norm(x)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    xd = x.detach().to(device).requires_grad_()
    y = Fn.InstanceNormFn.apply(xd, 1e-5)
    y.backward(dy.to(device))
    assert max_abs(y, y_ref) < 1e-5 and max_abs(xd.grad, x.grad) < 2e-5
    # upsample (+ add)
    x = torch.randn(2, 8, 50, generator=g, requires_grad=True)
    a = torch.randn(2, 8, 150, generator=g, requires_grad=True)
    y_ref = F.interpolate(x, scale_factor=3, mode="nearest") + a
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    xd, ad = x.detach().to(device).requires_grad_(), a.detach().to(device).requires_grad_()
    y = Fn.UpsampleNearestFn.apply(xd, 3, ad)
    y.backward(dy.to(device))
    assert max_abs(y, y_ref) == 0.0 and max_abs(xd.grad, x.grad) < 1e-5 and max_abs(ad.grad, a.grad) == 0.0
    # TADE modulation
    xn = torch.randn(2, 8, 40, generator=g, requires_grad=True)
    cg = torch.randn(2, 16, 80, generator=g, requires_grad=True)
    y_ref = cg[:, :8] * F.interpolate(xn, scale_factor=2, mode="nearest") + cg[:, 8:]
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    xnd, cgd = xn.detach().to(device).requires_grad_(), cg.detach().to(device).requires_grad_()
    y = Fn.TadeModulateFn.apply(xnd, cgd, 2)
    y.backward(dy.to(device))
    assert max_abs(y, y_ref) < 1e-6 and max_abs(xnd.grad, xn.grad) < 1e-5 and max_abs(cgd.grad, cg.grad) < 1e-6
    # gate
    z = (2.0 * torch.randn(2, 24, 77, generator=g)).requires_grad_()
    za, zb = z.split(12, dim=1)
    gate = torch.softmax(za, dim=1) if use_softmax else torch.sigmoid(za)
    y_ref = gate * torch.tanh(zb)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    zd = z.detach().to(device).requires_grad_()
    y = Fn.SoftmaxGateFn.apply(zd, use_softmax)
    y.backward(dy.to(device))
    assert max_abs(y, y_ref) < 1e-6 and max_abs(zd.grad, z.grad) < 2e-6


def test_generator_gradients_match_oracle_autograd(device):
    cfg = synth.STYLE_MELGAN_TINY
    g = models.StyleMelGANGenerator(**cfg)
    sd = synth_for(g, 17, 1.0)
    g.load_state_dict(sd)
    g = g.to(device).train()
    z = synth.synth_input("z", (2, cfg["in_channels"], 3), seed=17)
    c = synth.synth_input("c", (2, 80, 12), seed=17)
    sd_ref = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    y_ref = torch_cpu.style_melgan_generator(sd_ref, c, z, **cfg)
    w = synth.synth_input("w", tuple(y_ref.shape), seed=18)
    (y_ref * w).sum().backward()
    y = g(c.to(device), z.to(device))
    (y * w.to(device)).sum().backward()
    assert max_abs(y, y_ref) <= WAVE_TOL
    checked = 0
    for name, p in g.named_parameters():
        ref = sd_ref[name].grad
        if ref is None:
            continue
        scale = ref.abs().max().item() + 1e-8
        assert max_abs(p.grad, ref) <= 2e-4 * scale + 1e-7, name
        checked += 1
    assert checked > 50
