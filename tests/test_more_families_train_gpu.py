"""GPU: the Trainer drives the StyleMelGAN and UHiFiGAN families end to end (forward, losses, backward,
fused Adam) -- finite losses, parameters move, the auxiliary spectral loss on a fixed batch DESCENDS (a step
whose gradients, exchange or update were inconsistent with its forward would not), StyleMelGAN stays out of
hipGraph capture (random windows).  Forward values and gradients of both families are pinned separately
(tests/test_style_melgan_gpu.py, tests/test_uhifigan_gpu.py)."""
import tempfile

import numpy as np
import pytest
import torch

from parallelwavegan_amd import losses, models, optimizers
from parallelwavegan_amd.bin.train import Trainer
from tests.golden import synth

pytestmark = pytest.mark.gpu


def _run(device, gtype, g, d, batch, n_steps=24, use_graph=False):
    model = {"generator": g.to(device), "discriminator": d.to(device)}
    criterion = {"gen_adv": losses.GeneratorAdversarialLoss(), "dis_adv": losses.DiscriminatorAdversarialLoss(),
                 "stft": losses.MultiResolutionSTFTLoss(fft_sizes=[256, 512], hop_sizes=[32, 64], win_lengths=[128, 256]).to(device)}
    opt = {"generator": optimizers.Adam(model["generator"].parameters(), lr=5e-4, betas=(0.5, 0.9)),
           "discriminator": optimizers.Adam(model["discriminator"].parameters(), lr=1e-4, betas=(0.5, 0.9))}
    sched = {k: optimizers.lr_scheduler.StepLR(opt[k], step_size=10 ** 6, gamma=0.5) for k in model}
    config = dict(generator_type=gtype, generator_params={"out_channels": 1}, use_stft_loss=True,
                  use_subband_stft_loss=False, use_mel_loss=False, use_feat_match_loss=False, lambda_aux=1.0,
                  lambda_adv=1.0, generator_grad_norm=-1, discriminator_grad_norm=-1,
                  generator_train_start_steps=0, discriminator_train_start_steps=0, train_max_steps=100,
                  save_interval_steps=10 ** 9, eval_interval_steps=10 ** 9, log_interval_steps=10 ** 9,
                  distributed=False, rank=0, outdir=tempfile.mkdtemp(), progress=False, use_hip_graph=use_graph,
                  graph_warmup_steps=1)
    tr = Trainer(steps=1, epochs=0, data_loader={"train": [batch], "dev": [batch]}, sampler={"train": None, "dev": None},
                 model=model, criterion=criterion, optimizer=opt, scheduler=sched, config=config, device=device)
    tr.tqdm = None
    before = [p.detach().clone() for p in g.parameters()]
    aux, prev = [], 0.0
    for _ in range(n_steps):
        tr._train_step(batch)
        tr._flush_pending()
        cur = tr.total_train_loss["train/spectral_convergence_loss"] + tr.total_train_loss["train/log_stft_magnitude_loss"]
        aux.append(cur - prev)
        prev = cur
    assert all(np.isfinite(v) for v in tr.total_train_loss.values()), dict(tr.total_train_loss)
    assert any(not torch.equal(a, b) for a, b in zip(before, g.parameters()))
    first, last = float(np.mean(aux[:3])), float(np.mean(aux[-3:]))
    print(f"[family-train] {gtype}: aux loss {first:.4f} -> {last:.4f}")
    assert last < 0.97 * first, (gtype, aux)
    return tr


def test_style_melgan_training_steps(device):
    np.random.seed(0)
    # the generator draws z of length 1, so a batch holds exactly noise_upsample_factor = 16 frames (x256)
    cfg = dict(synth.STYLE_MELGAN_TINY, noise_upsample_scales=[4, 4], upsample_scales=[4, 4, 4, 4])
    g = models.StyleMelGANGenerator(**cfg)
    d = models.StyleMelGANDiscriminator(repeats=1, window_sizes=[128, 256, 512, 1024])
    gen = torch.Generator().manual_seed(0)
    c = torch.randn(2, 80, 16, generator=gen).to(device)
    y = (0.3 * torch.randn(2, 1, 16 * 256, generator=gen)).to(device)
    tr = _run(device, "StyleMelGANGenerator", g, d, ((c,), y), use_graph=True)
    assert not tr._graphs  # the random-window discriminator opts out of capture


def test_uhifigan_training_steps(device):
    g = models.UHiFiGANGenerator(**synth.UHIFIGAN_TINY)
    d = models.ParallelWaveGANDiscriminator(layers=4, conv_channels=16)
    gen = torch.Generator().manual_seed(1)
    frames = 256
    c = torch.randn(2, 80, frames, generator=gen).to(device)
    f0 = torch.rand(2, 1, frames, generator=gen).to(device)
    e = torch.randn(2, 1, frames * 8, generator=gen).to(device)
    y = (0.3 * torch.randn(2, 1, frames * 8, generator=gen)).to(device)
    tr = _run(device, "UHiFiGANGenerator", g, d, ((c, f0, e), y), use_graph=True)
    assert len(tr._graphs) == 1  # dropout inside a captured step (device-resident mask seed)


@pytest.mark.parametrize("family", ["style_melgan", "uhifigan"])
def test_two_training_steps_match_the_reference_trainer(family, device):
    """Two ``Trainer._train_step`` calls of the UNMODIFIED reference for StyleMelGAN and UHiFiGAN
    (tests/golden/{style_melgan,uhifigan}_train.npz, made by tests/golden/make_golden.py: family_train_steps): every
    logged loss of both steps, every parameter's first-moment norm and <first update, first moment>, the final
    parameter sums.  StyleMelGAN's randomness is reproduced, not avoided: the generator draws z from torch's CPU
    generator and the discriminator its window starts from numpy's, in the reference's call order, so seeding both
    right before the steps gives the reference's own draws."""
    from tests.test_pwg_mb_train_gpu import _run_and_compare
    from tests.util import load_golden

    gold = load_golden(f"{family}_train")
    n_steps, seed = (int(v) for v in gold["meta"])
    if family == "style_melgan":
        g = models.StyleMelGANGenerator(**synth.STYLE_MELGAN_TRAIN)
        d = models.StyleMelGANDiscriminator(**synth.STYLE_MELGAN_TRAIN_D)
        g.load_state_dict(synth.synth_state_dict(g.state_dict(), seed=seed, g_scale=1.1))
        d.load_state_dict(synth.synth_state_dict(d.state_dict(), seed=seed + 1, g_scale=1.2, skip=synth.PQMF_BUFFERS),
                          strict=False)
        c = synth.synth_input("c", (2, 80, 16), seed=seed)
        y = 0.3 * synth.synth_input("y", (2, 1, 16 * 256), seed=seed)
        x, gtype = (c,), "StyleMelGANGenerator"
    else:
        g = models.UHiFiGANGenerator(**synth.UHIFIGAN_TRAIN)
        d = models.ParallelWaveGANDiscriminator(layers=4, conv_channels=16)
        g.load_state_dict(synth.synth_state_dict(g.state_dict(), seed=seed, g_scale=0.6))
        d.load_state_dict(synth.synth_state_dict(d.state_dict(), seed=seed + 1, g_scale=1.4))
        frames = 64
        c = synth.synth_input("c", (2, 80, frames), seed=seed)
        f0 = synth.synth_input("f0", (2, 1, frames), seed=seed).abs()
        e = synth.synth_input("excitation", (2, 1, frames * 8), seed=seed)
        y = 0.3 * synth.synth_input("y", (2, 1, frames * 8), seed=seed)
        x, gtype = (c, f0, e), "UHiFiGANGenerator"
    model = {"generator": g.to(device), "discriminator": d.to(device)}
    criterion = {"gen_adv": losses.GeneratorAdversarialLoss(), "dis_adv": losses.DiscriminatorAdversarialLoss(),
                 "stft": losses.MultiResolutionSTFTLoss(**synth.FAMILY_TRAIN_STFT).to(device)}
    opt = {k: optimizers.Adam(model[k].parameters(), lr=synth.FAMILY_TRAIN_LR[k], betas=(0.5, 0.9)) for k in model}
    sched = {k: optimizers.lr_scheduler.StepLR(opt[k], step_size=10 ** 6, gamma=0.5) for k in model}
    config = dict(synth.FAMILY_TRAIN_CFG, generator_type=gtype, generator_params={"out_channels": 1},
                  train_max_steps=1 + n_steps, save_interval_steps=10 ** 9, eval_interval_steps=10 ** 9,
                  log_interval_steps=10 ** 9, distributed=False, rank=0, outdir=tempfile.mkdtemp(), progress=False)
    batches = [(x, y)] * n_steps
    tr = Trainer(steps=1, epochs=0, data_loader={"train": batches, "dev": batches}, sampler={"train": None, "dev": None},
                 model=model, criterion=criterion, optimizer=opt, scheduler=sched, config=config, device=device)
    torch.manual_seed(seed)
    np.random.seed(seed)
    # (final parameter sums at 1e-2, as for multi-band MelGAN: Adam's first steps are +-lr by the SIGN of the gradient, which
    # is rounding noise on the entries of small tensors whose gradient is ~0; the losses, first moments and <update, moment>
    # above are held to 2e-4 / 3e-3 / 5e-3)
    _run_and_compare(tr, batches, gold, model, opt, 0.1, final_rtol=1e-2)


def test_trainer_steps_an_optimizer_of_torchs_own(device):
    """``*_optimizer_type`` may name anything ``torch.optim`` has (parallel_wavegan/optimizers/__init__.py re-exports
    it): such an optimizer runs eagerly from ``.grad`` -- never captured, never handed the fused optimizers' folded
    averaging factor (torch's Adam family asserts that an attribute called ``grad_scale`` is unset)."""
    g = models.MelGANGenerator(in_channels=80, channels=64, upsample_scales=[4, 4, 2, 2], stack_kernel_size=3, stacks=1)
    d = models.ParallelWaveGANDiscriminator(layers=4, conv_channels=16)
    gen = torch.Generator().manual_seed(3)
    c = torch.randn(2, 80, 16, generator=gen).to(device)
    with torch.no_grad():
        t = g.to(device)(c).shape[-1]
    y = (0.3 * torch.randn(2, 1, t, generator=gen)).to(device)
    model = {"generator": g.to(device), "discriminator": d.to(device)}
    criterion = {"gen_adv": losses.GeneratorAdversarialLoss(), "dis_adv": losses.DiscriminatorAdversarialLoss(),
                 "stft": losses.MultiResolutionSTFTLoss(fft_sizes=[256, 512], hop_sizes=[32, 64], win_lengths=[128, 256]).to(device)}
    opt = {"generator": torch.optim.NAdam(model["generator"].parameters(), lr=5e-4),
           "discriminator": torch.optim.SGD(model["discriminator"].parameters(), lr=1e-3, momentum=0.9)}
    sched = {k: optimizers.lr_scheduler.StepLR(opt[k], step_size=10 ** 6, gamma=0.5) for k in model}
    config = dict(generator_type="MelGANGenerator", generator_params={"out_channels": 1}, use_stft_loss=True,
                  use_subband_stft_loss=False, use_mel_loss=False, use_feat_match_loss=False, lambda_aux=1.0,
                  lambda_adv=1.0, generator_grad_norm=10.0, discriminator_grad_norm=-1,
                  generator_train_start_steps=0, discriminator_train_start_steps=0, train_max_steps=100,
                  save_interval_steps=10 ** 9, eval_interval_steps=10 ** 9, log_interval_steps=10 ** 9,
                  distributed=False, rank=0, outdir=tempfile.mkdtemp(), progress=False, use_hip_graph=True,
                  graph_warmup_steps=1)
    batch = ((c,), y)
    tr = Trainer(steps=1, epochs=0, data_loader={"train": [batch], "dev": [batch]}, sampler={"train": None, "dev": None},
                 model=model, criterion=criterion, optimizer=opt, scheduler=sched, config=config, device=device)
    tr.tqdm = None
    before = {k: [p.detach().clone() for p in m.parameters()] for k, m in model.items()}
    for _ in range(4):
        tr._train_step(batch)
    tr._flush_pending()
    assert not tr._graphs  # use_hip_graph is ignored for optimizers the capture cannot replay
    assert all(np.isfinite(v) for v in tr.total_train_loss.values()), dict(tr.total_train_loss)
    for k, m in model.items():
        assert any(not torch.equal(a, b) for a, b in zip(before[k], m.parameters())), k
    assert opt["generator"].state and opt["discriminator"].state
