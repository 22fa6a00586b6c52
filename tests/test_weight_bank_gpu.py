"""Weight bank (csrc/conv1d.hip ``pwg_weight_bank_*``, parallelwavegan_amd/weight_bank.py): the two-launch weight
preparation of a whole model equals the per-layer weight_norm_scale + pack + pack sequence bit for bit, for every layer
kind of the HiFi-GAN / MelGAN / PWG model families (plain, weight-norm, transposed, grouped, (k,1) Conv2d)."""
import ctypes

import pytest
import torch

from parallelwavegan_amd import _lib, ops
from parallelwavegan_amd import functional as Fn
from parallelwavegan_amd.models import (HiFiGANGenerator, HiFiGANMultiScaleMultiPeriodDiscriminator, MelGANGenerator,
                                        ParallelWaveGANGenerator)
from parallelwavegan_amd.weight_bank import WeightBank

pytestmark = pytest.mark.gpu


def _models(dev):
    torch.manual_seed(7)
    g = HiFiGANGenerator(channels=64, upsample_scales=(5, 4, 3), upsample_kernel_sizes=(10, 8, 6),
                         resblock_kernel_sizes=(3, 7), resblock_dilations=[(1, 3), (1, 2)]).to(dev)
    d = HiFiGANMultiScaleMultiPeriodDiscriminator(
        scales=2, periods=[2, 3],
        scale_discriminator_params={"in_channels": 1, "out_channels": 1, "kernel_sizes": [15, 41, 5, 3], "channels": 16,
                                    "max_downsample_channels": 64, "max_groups": 4, "bias": True,
                                    "downsample_scales": [4, 4, 1], "nonlinear_activation": "LeakyReLU",
                                    "nonlinear_activation_params": {"negative_slope": 0.1}},
        period_discriminator_params={"in_channels": 1, "out_channels": 1, "kernel_sizes": [5, 3], "channels": 8,
                                     "downsample_scales": [3, 3, 1], "max_downsample_channels": 64, "bias": True,
                                     "nonlinear_activation": "LeakyReLU",
                                     "nonlinear_activation_params": {"negative_slope": 0.1}, "use_weight_norm": True,
                                     "use_spectral_norm": False}).to(dev)
    m = MelGANGenerator(channels=64, upsample_scales=[4, 2], stacks=2).to(dev)
    p = ParallelWaveGANGenerator(layers=4, stacks=2).to(dev)
    plain = HiFiGANGenerator(channels=32, upsample_scales=(2, 2), upsample_kernel_sizes=(4, 4), resblock_kernel_sizes=(3,),
                             resblock_dilations=[(1,)], use_weight_norm=False).to(dev)
    return {"hifigan_g": g, "hifigan_d": d, "melgan_g": m, "pwg_g": p, "plain_g": plain}


@pytest.mark.parametrize("with_bwd", [True, False])
def test_bank_images_equal_the_per_layer_kernels(device, with_bwd):
    for name, model in _models(device).items():
        bank = WeightBank(model)
        bank.ensure(with_bwd)
        layers = bank._layers
        assert layers, name
        n_wn = 0
        for m in layers:
            pw = m._cache_packed
            assert isinstance(pw, Fn.PreparedWeights) and m._cache_key == m._params_key()
            assert m.prepared() is pw  # the layer's own cache is warm: no further launches
            desc = m.make_desc(1, m._probe_len())
            w3 = m._w3(m.raw_weight.detach())
            scale = None
            if m.has_weight_norm:
                n_wn += 1
                scale = ops.weight_norm_scale(w3, m.weight_g.detach().reshape(-1).contiguous())
                assert torch.equal(pw.scale, scale), (name, type(m).__name__)
            else:
                assert pw.scale is None
            if getattr(m, "bank_images", True):
                assert torch.equal(pw.fwd, ops.pack_weight(desc, w3, scale)), (name, type(m).__name__, tuple(w3.shape))
                if with_bwd:
                    assert pw._bwd is not None
                assert torch.equal(pw.bwd(desc), ops.pack_weight_bwd(desc, w3, scale)), (name, tuple(w3.shape))
            else:
                assert pw._fwd is None  # built lazily, only if the un-fused path ever runs
        if name != "plain_g":
            assert n_wn > 0
        # spectral-norm layers (first scale discriminator) are not the bank's: their weight is a fresh node per forward
        assert all(not m.has_spectral_norm for m in layers)
        # a second call with unchanged parameters launches nothing and keeps the objects
        before = [m._cache_packed for m in layers]
        with ops.profile() as prof:
            bank.ensure(with_bwd)
        assert not prof.results and all(a is b for a, b in zip(before, (m._cache_packed for m in layers)))


def test_bank_refreshes_after_a_parameter_update_and_flags_stale_images(device):
    model = _models(device)["hifigan_d"]
    bank = WeightBank(model)
    bank.ensure(True)
    conv = bank._layers[3]
    old = conv._cache_packed
    old_fwd = old.fwd.clone()
    with torch.no_grad():
        conv.raw_weight.mul_(1.5)  # (bumps torch's version counter, as an optimizer step bumps ops.param_epoch)
        if conv.has_weight_norm:
            conv.weight_g.mul_(0.5)
    with ops.profile() as prof:
        bank.ensure(True)
    assert sum(v["launches"] for v in prof.results.values()) == 2, prof.results  # one row-scale + one packing launch
    new = conv._cache_packed
    assert new is not old and not torch.equal(new.fwd, old_fwd)
    with pytest.raises(RuntimeError, match="overwritten"):
        old.fwd  # noqa: B018
    desc = conv.make_desc(1, conv._probe_len())
    w3 = conv._w3(conv.raw_weight.detach())
    scale = ops.weight_norm_scale(w3, conv.weight_g.detach().reshape(-1).contiguous()) if conv.has_weight_norm else None
    assert torch.equal(new.fwd, ops.pack_weight(desc, w3, scale))


def test_forward_and_gradients_are_identical_with_and_without_the_bank(device):
    torch.manual_seed(3)
    model = _models(device)["hifigan_d"]
    x = torch.randn(2, 1, 2400, device=device)

    def run(use_bank):
        ops.bump_param_epoch()  # drop every cached image
        for p in model.parameters():
            p.grad = None
        if use_bank:
            WeightBank(model).ensure(True)
        outs = model(x)
        loss = sum(o[-1].square().mean() for o in outs)
        loss.backward()
        return loss.detach().clone(), [p.grad.clone() for p in model.parameters()]

    model.eval()  # (no power iteration: the two runs see the same spectral-norm state)
    l0, g0 = run(False)
    l1, g1 = run(True)
    assert torch.equal(l0, l1)
    sn = {id(p) for m in model.modules() if getattr(m, "has_spectral_norm", False) for p in m.parameters(recurse=False)}
    for p, a, b in zip(model.parameters(), g0, g1):
        if id(p) in sn:  # (the spectral-norm backward sums <dW, W> with fp32 atomics: equal to ~1e-7, DESIGN s4)
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-7)
        else:
            assert torch.equal(a, b)
