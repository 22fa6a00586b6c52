"""GPU parity: Parallel WaveGAN / Multi-band MelGAN / PQMF vs the reference's golden vectors
and the oracle (configs C1, C2, C4)."""
import numpy as np
import pytest
import torch
import yaml

from oracle import torch_cpu
from parallelwavegan_amd import layers, models
from tests.golden import synth
from tests.util import WAVE_TOL, load_golden, max_abs

pytestmark = pytest.mark.gpu

PWG_G = dict(in_channels=1, out_channels=1, kernel_size=3, layers=30, stacks=3, residual_channels=64,
             gate_channels=128, skip_channels=64, aux_channels=80, aux_context_window=2, dropout=0.0,
             use_weight_norm=True, upsample_net="ConvInUpsampleNetwork",
             upsample_params={"upsample_scales": [4, 4, 4, 4]})
PWG_D = dict(in_channels=1, out_channels=1, kernel_size=3, layers=10, conv_channels=64, bias=True,
             use_weight_norm=True, nonlinear_activation="LeakyReLU",
             nonlinear_activation_params={"negative_slope": 0.2})
MB_G = dict(in_channels=80, out_channels=4, kernel_size=7, channels=384, upsample_scales=[8, 4, 2],
            stack_kernel_size=3, stacks=4, use_weight_norm=True, use_causal_conv=False)
MB_D = dict(in_channels=1, out_channels=1, scales=3, downsample_pooling="AvgPool1d",
            downsample_pooling_params={"kernel_size": 4, "stride": 2, "padding": 1, "count_include_pad": False},
            kernel_sizes=[5, 3], channels=16, max_downsample_channels=512, downsample_scales=[4, 4, 4],
            nonlinear_activation="LeakyReLU", nonlinear_activation_params={"negative_slope": 0.2},
            use_weight_norm=True)


def test_pwg_generator_and_discriminator_match_reference_golden(device):
    gold = load_golden("pwg_v1")
    frames, seed = (int(v) for v in gold["meta"])
    g = models.ParallelWaveGANGenerator(**PWG_G)
    g.load_state_dict(synth.synth_state_dict(g.state_dict(), seed=seed, g_scale=synth.PWG_G_SCALE))
    g = g.to(device).eval()
    c = synth.synth_input("c", (1, 80, frames + 4), seed=seed)
    z = synth.synth_input("z", (1, 1, frames * 256), seed=seed)
    with torch.no_grad():
        y = g(z.to(device), c.to(device))
        assert max_abs(y, gold["g_y"]) <= WAVE_TOL
        g.remove_weight_norm()
        y_inf = g.inference(c=c[0, :, 2:-2].transpose(0, 1).numpy(), x=z[0].transpose(0, 1).numpy())
        assert max_abs(y_inf, gold["g_y_inference"]) <= WAVE_TOL
    d = models.ParallelWaveGANDiscriminator(**PWG_D)
    d.load_state_dict(synth.synth_state_dict(d.state_dict(), seed=seed + 1, g_scale=1.4))
    d = d.to(device).eval()
    x = 0.5 * synth.synth_input("wave", (2, 1, 6000), seed=seed)
    with torch.no_grad():
        assert max_abs(d(x.to(device)), gold["d_y"]) <= 3e-5 * max(1.0, float(np.abs(gold["d_y"]).max()))


def test_multi_band_melgan_and_pqmf_match_reference_golden(device):
    gold = load_golden("mb_melgan_v2")
    seed = int(gold["meta"][0])
    g = models.MelGANGenerator(**MB_G)
    g.load_state_dict(synth.synth_state_dict(g.state_dict(), seed=seed, g_scale=synth.MELGAN_G_SCALE))
    g = g.to(device).eval()
    pqmf = layers.PQMF().to(device)
    c = synth.synth_input("c", (2, 80, 24), seed=seed)
    with torch.no_grad():
        y_mb = g(c.to(device))
        assert max_abs(y_mb, gold["g_y_mb"]) <= WAVE_TOL
        assert max_abs(pqmf.synthesis(y_mb), gold["g_y_full"]) <= WAVE_TOL
        g.pqmf = pqmf
        assert max_abs(g.inference(c[0].transpose(0, 1).numpy()), gold["g_y_inference"]) <= WAVE_TOL
        w = (0.5 * synth.synth_input("wave", (2, 1, 4096), seed=seed)).to(device)
        a = pqmf.analysis(w)
        assert max_abs(a, gold["pqmf_analysis"]) <= 2e-6
        assert max_abs(pqmf.synthesis(a), gold["pqmf_round_trip"]) <= 5e-6
    d = models.MelGANMultiScaleDiscriminator(**MB_D)
    d.load_state_dict(synth.synth_state_dict(d.state_dict(), seed=seed + 1, g_scale=1.2))
    d = d.to(device).eval()
    with torch.no_grad():
        o = d(w)
    logits = np.concatenate([t[-1].reshape(-1).cpu().numpy() for t in o])
    assert np.abs(logits - gold["d_logits"]).max() <= 3e-5 * max(1.0, np.abs(gold["d_logits"]).max())
    st = np.stack([np.array([f.double().abs().sum().item(), (f.double() ** 2).sum().item()]) for t in o for f in t])
    np.testing.assert_allclose(st, gold["d_feat_stats"][:, 1:], rtol=3e-5)


def _grads_match(loss_dev, params_dev, loss_ref, params_ref, rtol=2e-3, names=None):
    gd = torch.autograd.grad(loss_dev, params_dev, allow_unused=True)
    gr = torch.autograd.grad(loss_ref, params_ref, allow_unused=True)
    bad = []
    # gradients that are analytically zero (weight_v of 1-element rows, bias under a zero-sum
    # cotangent) are pure rounding noise on both sides: floor the scale by the global gradient scale
    floor = 1e-3 * max(b.abs().max().item() for b in gr if b is not None)
    for i, (a, b) in enumerate(zip(gd, gr)):
        assert (a is None) == (b is None)
        if a is None:
            continue
        denom = max(b.abs().max().item(), floor)
        err = (a.cpu() - b).abs().max().item() / denom
        if err > rtol:
            bad.append((names[i] if names and i < len(names) else i, tuple(b.shape), round(err, 5)))
    assert not bad, bad[:12]


def test_pwg_training_gradients_match_oracle(device):
    """d(loss)/d(every parameter) of G and D through the HIP backward kernels vs CPU autograd."""
    cfg = dict(PWG_G, layers=6, stacks=3)
    g = models.ParallelWaveGANGenerator(**cfg)
    sd = synth.synth_state_dict(g.state_dict(), seed=9, g_scale=1.0)
    g.load_state_dict(sd)
    g = g.to(device)
    z = synth.synth_input("z", (2, 1, 1024), seed=9)
    c = synth.synth_input("c", (2, 80, 8), seed=9)
    names = [n for n, _ in g.named_parameters()]
    y = g(z.to(device), c.to(device))
    sd_ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    y_ref = torch_cpu.pwg_generator(sd_ref, z, c, **cfg)
    assert max_abs(y, y_ref) <= WAVE_TOL
    w = torch.linspace(-0.5, 1, y_ref.numel()).reshape(y_ref.shape)
    _grads_match((y * w.to(device)).sum(), [p for _, p in g.named_parameters()], (y_ref * w).sum(),
                 [sd_ref[n] for n in names], names=names)
    d = models.ParallelWaveGANDiscriminator(**PWG_D)
    sdd = synth.synth_state_dict(d.state_dict(), seed=10, g_scale=1.4)
    d.load_state_dict(sdd)
    d = d.to(device)
    x = (0.5 * synth.synth_input("wave", (2, 1, 2000), seed=9)).requires_grad_()
    xd = x.detach().to(device).requires_grad_()
    sdd_ref = {k: v.clone().requires_grad_(True) for k, v in sdd.items()}
    o, o_ref = d(xd), torch_cpu.pwg_discriminator(sdd_ref, x, **PWG_D)
    dn = [n for n, _ in d.named_parameters()]
    _grads_match((o ** 2).mean(), [p for _, p in d.named_parameters()] + [xd], (o_ref ** 2).mean(),
                 [sdd_ref[n] for n in dn] + [x])


def test_melgan_training_gradients_match_oracle(device):
    """Reflect-padded convs, transposed convs, PQMF and the MelGAN D under autograd."""
    cfg = dict(MB_G, channels=64, upsample_scales=[4, 2], stacks=2)
    g = models.MelGANGenerator(**cfg)
    sd = synth.synth_state_dict(g.state_dict(), seed=12, g_scale=1.0)
    g.load_state_dict(sd)
    g = g.to(device)
    pqmf = layers.PQMF().to(device)
    c = synth.synth_input("c", (2, 80, 40), seed=12)
    names = [n for n, _ in g.named_parameters()]
    y = pqmf.synthesis(g(c.to(device)))
    sd_ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    y_ref = torch_cpu.pqmf_synthesis(torch_cpu.melgan_generator(sd_ref, c, **cfg))
    assert max_abs(y, y_ref) <= WAVE_TOL
    w = torch.linspace(-0.5, 1, y_ref.numel()).reshape(y_ref.shape)
    _grads_match((y * w.to(device)).sum(), [p for _, p in g.named_parameters()], (y_ref * w).sum(),
                 [sd_ref[n] for n in names])
    d = models.MelGANMultiScaleDiscriminator(**MB_D)
    sdd = synth.synth_state_dict(d.state_dict(), seed=13, g_scale=1.2)
    d.load_state_dict(sdd)
    d = d.to(device)
    x = (0.5 * synth.synth_input("wave", (2, 1, 2048), seed=12)).requires_grad_()
    xd = x.detach().to(device).requires_grad_()
    sdd_ref = {k: v.clone().requires_grad_(True) for k, v in sdd.items()}
    o = d(xd)
    o_ref = torch_cpu.melgan_multi_scale_discriminator(sdd_ref, x, **MB_D)
    loss = sum((t[-1] ** 2).mean() for t in o) + sum(f.abs().mean() for t in o for f in t[:-1])
    loss_ref = sum((t[-1] ** 2).mean() for t in o_ref) + sum(f.abs().mean() for t in o_ref for f in t[:-1])
    dn = [n for n, _ in d.named_parameters()]
    _grads_match(loss, [p for _, p in d.named_parameters()] + [xd], loss_ref, [sdd_ref[n] for n in dn] + [x])
    # sub-band analysis gradient
    a = pqmf.analysis(xd)
    a_ref = torch_cpu.pqmf_analysis(x)
    _grads_match((a ** 2).sum(), [xd], (a_ref ** 2).sum(), [x])


def test_pwg_with_melgan_upsampler_matches_reference_golden(device):
    """ParallelWaveGANGenerator(upsample_net="MelGANGenerator") (models/parallel_wavegan.py:90-98)."""
    import copy

    gold = load_golden("pwg_melgan_upsampler")
    seed = int(gold["meta"][0])
    g = models.ParallelWaveGANGenerator(**copy.deepcopy(synth.PWG_MELGAN_UPSAMPLER))
    g.load_state_dict(synth.synth_state_dict(g.state_dict(), seed=seed, g_scale=synth.PWG_G_SCALE))
    g = g.to(device).eval()
    c = synth.synth_input("c", (2, 80, 9), seed=seed)
    z = synth.synth_input("z", (2, 1, 9 * 256), seed=seed)
    with torch.no_grad():
        assert max_abs(g(z.to(device), c.to(device)), gold["y"]) <= WAVE_TOL
