"""GPU parity: HiFi-GAN MSD+MPD forward vs the reference's golden vectors and the oracle."""
import numpy as np
import pytest
import torch
import yaml

from oracle import torch_cpu
from parallelwavegan_amd.models import HiFiGANMultiScaleMultiPeriodDiscriminator
from tests.golden import synth
from tests.util import load_golden, max_abs

pytestmark = pytest.mark.gpu

D_PARAMS = yaml.safe_load("""
scales: 3
scale_downsample_pooling: "AvgPool1d"
scale_downsample_pooling_params: {kernel_size: 4, stride: 2, padding: 2}
scale_discriminator_params:
    in_channels: 1
    out_channels: 1
    kernel_sizes: [15, 41, 5, 3]
    channels: 128
    max_downsample_channels: 1024
    max_groups: 16
    bias: true
    downsample_scales: [4, 4, 4, 4, 1]
    nonlinear_activation: "LeakyReLU"
    nonlinear_activation_params: {negative_slope: 0.1}
follow_official_norm: true
periods: [2, 3, 5, 7, 11]
period_discriminator_params:
    in_channels: 1
    out_channels: 1
    kernel_sizes: [5, 3]
    channels: 32
    downsample_scales: [3, 3, 3, 3, 1]
    max_downsample_channels: 1024
    bias: true
    nonlinear_activation: "LeakyReLU"
    nonlinear_activation_params: {negative_slope: 0.1}
    use_weight_norm: true
    use_spectral_norm: false
""")


def _stats(t):
    t = t.detach().cpu().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


def test_msmpd_matches_reference_golden(device):
    gold = load_golden("hifigan_v1_d")
    _, t, seed = (int(v) for v in gold["meta"])
    d = HiFiGANMultiScaleMultiPeriodDiscriminator(**D_PARAMS)
    d.load_state_dict(synth.synth_state_dict(d.state_dict(), seed=seed, g_scale=1.0))
    d = d.to(device)
    x = (0.5 * synth.synth_input("wave", (2, 1, t), seed=seed)).to(device)
    with torch.no_grad():
        d.eval()
        o = d(x)
        logits = np.concatenate([t_[-1].reshape(-1).cpu().numpy() for t_ in o])
        scale = np.abs(gold["eval_logits"]).max()
        assert np.abs(logits - gold["eval_logits"]).max() <= 2e-5 * max(scale, 1.0)
        st = np.stack([_stats(f) for t_ in o for f in t_])
        np.testing.assert_allclose(st[:, 1:], gold["eval_feat_stats"][:, 1:], rtol=2e-5)
        # two training-mode calls: spectral-norm power iterations update u, v like the hook does
        d.train()
        d(x)
        o = d(x)
        logits = np.concatenate([t_[-1].reshape(-1).cpu().numpy() for t_ in o])
        assert np.abs(logits - gold["train2_logits"]).max() <= 2e-5 * max(scale, 1.0)
        u0 = d.msd.discriminators[0].layers[1][0].weight_u.cpu().numpy()
        assert np.abs(u0 - gold["train2_u0"]).max() <= 1e-5


@pytest.mark.parametrize("t,batch", [(8192, 2), (2051, 1), (4097, 3)])
def test_msmpd_matches_oracle_feature_maps(t, batch, device):
    d = HiFiGANMultiScaleMultiPeriodDiscriminator(**D_PARAMS)
    sd = synth.synth_state_dict(d.state_dict(), seed=7, g_scale=1.0)
    d.load_state_dict(sd)
    d = d.to(device).eval()
    x = 0.5 * synth.synth_input("wave", (batch, 1, t), seed=t)
    with torch.no_grad():
        mine = d(x.to(device))
        ref = torch_cpu.hifigan_msmpd(dict(sd), x, training=False, **D_PARAMS)
    assert len(mine) == len(ref) == 8
    for dm, dr in zip(mine, ref):
        assert len(dm) == len(dr)
        for a, b in zip(dm, dr):
            assert tuple(a.shape) == tuple(b.shape)
            assert max_abs(a, b) <= 3e-5 * max(1.0, b.abs().max().item())


def test_deferred_activation_form_equals_the_post_activation_form(device):
    """layers.activation.PreActivated: with every LeakyReLU applied by its consumers (the next convolution on load, the
    feature-matching reduction kernel) the logits, the feature-matching loss and every parameter / input gradient equal
    the reference form, and the backward pass launches no activation-gradient kernel."""
    import torch

    from parallelwavegan_amd import losses, ops
    from parallelwavegan_amd.layers.activation import PreActivated, set_deferred_activation
    from parallelwavegan_amd.models import HiFiGANMultiScaleMultiPeriodDiscriminator, MelGANMultiScaleDiscriminator

    torch.manual_seed(5)
    hifi = HiFiGANMultiScaleMultiPeriodDiscriminator(
        scales=2, periods=[2, 3],
        scale_discriminator_params={"in_channels": 1, "out_channels": 1, "kernel_sizes": [15, 41, 5, 3], "channels": 16,
                                    "max_downsample_channels": 64, "max_groups": 4, "bias": True,
                                    "downsample_scales": [4, 4, 1], "nonlinear_activation": "LeakyReLU",
                                    "nonlinear_activation_params": {"negative_slope": 0.1}},
        period_discriminator_params={"in_channels": 1, "out_channels": 1, "kernel_sizes": [5, 3], "channels": 8,
                                     "downsample_scales": [3, 3, 1], "max_downsample_channels": 64, "bias": True,
                                     "nonlinear_activation": "LeakyReLU",
                                     "nonlinear_activation_params": {"negative_slope": 0.1}, "use_weight_norm": True,
                                     "use_spectral_norm": False}).to(device).eval()
    mel = MelGANMultiScaleDiscriminator(scales=2, channels=8, max_downsample_channels=64, downsample_scales=[4, 4]).to(device)
    fm = losses.FeatureMatchLoss(average_by_layers=False, average_by_discriminators=False)
    adv = losses.GeneratorAdversarialLoss(average_by_discriminators=False)
    for name, d in (("hifigan", hifi), ("melgan", mel)):
        x = (0.5 * torch.randn(2, 1, 2400, device=device)).requires_grad_()
        y = 0.5 * torch.randn(2, 1, 2400, device=device)
        res = {}
        for deferred in (False, True):
            assert set_deferred_activation(d, deferred) > 0
            for p in d.parameters():
                p.grad = None
            x.grad = None
            with torch.no_grad():
                target = d(y)
            with ops.profile() as prof:
                out = d(x)
                loss = adv(out) + 2.0 * fm(out, target)
                loss.backward()
            assert all(isinstance(o, PreActivated) == deferred for o in out), name
            n_act = prof.results.get("act_backward_kernel", {}).get("launches", 0)
            res[deferred] = (loss.detach().clone(), [o[-1].detach().clone() for o in out], x.grad.clone(),
                             [p.grad.clone() for p in d.parameters()], n_act)
        set_deferred_activation(d, False)
        (l0, lg0, gx0, gp0, n0), (l1, lg1, gx1, gp1, n1) = res[False], res[True]
        assert n0 > 0 and n1 == 0, (name, n0, n1)
        assert all(torch.equal(a, b) for a, b in zip(lg0, lg1)), name  # the same values reach every convolution
        assert abs(l0.item() - l1.item()) <= 1e-6 * abs(l0.item())
        scale = gx0.abs().max().item()
        assert (gx0 - gx1).abs().max().item() <= 2e-6 * scale, name
        for a, b in zip(gp0, gp1):
            assert (a - b).abs().max().item() <= 3e-6 * max(a.abs().max().item(), 1e-6), name


@pytest.mark.parametrize("rows,cols", [(128, 15), (128, 1312), (1024, 5120), (1, 3072), (1024, 20)])
def test_spectral_norm_iteration_in_fewer_launches_is_bit_identical(rows, cols, device):
    """Round 6: pwg_spectral_norm_forward_saved (normalisation folded into the following matrix-vector product, sigma into
    the second normalisation, the backward pass's u / v copies written by the same kernels) against the 8-launch chain +
    clones, on the weight shapes of HiFi-GAN's spectrally normalised scale discriminator (and one that does not fit the
    workspace split and falls back): w, sigma, u, v and the saved copies are equal bit for bit; float64 check of sigma."""
    import ctypes

    from parallelwavegan_amd import _lib
    from parallelwavegan_amd.ops import _ptr, _stream

    g = torch.Generator().manual_seed(rows + cols)
    w0 = torch.randn(rows, cols, generator=g).to(device)
    u0 = torch.nn.functional.normalize(torch.randn(rows, generator=g), dim=0).to(device)
    v0 = torch.nn.functional.normalize(torch.randn(cols, generator=g), dim=0).to(device)
    L = _lib.lib()

    def old():
        u, v = u0.clone(), v0.clone()
        sigma, w = torch.empty(1, device=device), torch.empty_like(w0)
        tmp = torch.empty(max(rows, 32 * cols), device=device)
        _lib.check(L.pwg_spectral_norm_forward(_ptr(w0), _ptr(u), _ptr(v), _ptr(sigma), _ptr(w), _ptr(tmp), rows, cols, 1,
                                               1e-12, _stream()), "spectral_norm_forward")
        return w, sigma, u, v, u.clone(), v.clone()

    def new():
        u, v = u0.clone(), v0.clone()
        sigma, w = torch.empty(1, device=device), torch.empty_like(w0)
        tmp = torch.full((max(rows, 32 * cols),), float("nan"), device=device)
        us, vs = torch.full_like(u, float("nan")), torch.full_like(v, float("nan"))
        _lib.check(L.pwg_spectral_norm_forward_saved(_ptr(w0), _ptr(u), _ptr(v), _ptr(sigma), _ptr(w), _ptr(tmp), _ptr(us),
                                                     _ptr(vs), rows, cols, 1, 1e-12, _stream()), "spectral_norm_forward_saved")
        return w, sigma, u, v, us, vs

    a, b = old(), new()
    for name, x, y in zip(("w", "sigma", "u", "v", "u_saved", "v_saved"), a, b):
        assert torch.equal(x, y), name
    wd = w0.double().cpu()
    v_ref = torch.nn.functional.normalize(wd.t() @ u0.double().cpu(), dim=0)
    u_ref = torch.nn.functional.normalize(wd @ v_ref, dim=0)
    sigma_ref = float(u_ref @ (wd @ v_ref))
    assert abs(b[1].item() - sigma_ref) <= 1e-5 * abs(sigma_ref)
