"""GPU parity of whole training steps for configs C2 (Parallel WaveGAN.v1: RAdam, StepLR, gradient
clipping, multi-resolution STFT loss) and C4 (Multi-band MelGAN.v2: Adam amsgrad, PQMF, full-band +
sub-band STFT losses) against two steps of the reference's own Trainer."""
import tempfile

import numpy as np
import pytest
import torch

from parallelwavegan_amd import layers, losses, models, optimizers
from parallelwavegan_amd.bin.train import Trainer
from tests.golden import synth
from tests.test_pwg_melgan_gpu import MB_D, MB_G, PWG_D, PWG_G
from tests.util import load_golden

pytestmark = pytest.mark.gpu


def _run_and_compare(tr, batches, gold, model, opt, mom_scale, loose=None, final_rtol=1e-4):
    loose = loose or {}
    tr.tqdm = None
    prev = {}
    p0 = {key: {n: p.detach().clone() for n, p in model[key].named_parameters()} for key in model}
    for i, b in enumerate(batches):
        tr._train_step(b)
        tr._flush_pending()
        cur = dict(tr.total_train_loss)
        for k, v in cur.items():
            want = float(gold[f"step{i}/{k}"])
            got = v - prev.get(k, 0.0)
            tol = loose.get((i, k), 2e-4)
            print(f"[train-parity] step {i} {k}: got {got:.7g} want {want:.7g} rel {abs(got - want) / max(abs(want), 1e-3):.2e}")
            assert abs(got - want) <= tol * max(abs(want), 1e-3), (i, k, got, want)
        prev = cur
        if i == 0:
            for key in ("generator", "discriminator"):
                names = {p: n for n, p in model[key].named_parameters()}
                norms = {names[p]: float(s["exp_avg"].double().norm()) for p, s in opt[key].state.items()}
                gn = [str(n) for n in gold[f"momnorm_names/{key}"]]
                assert sorted(norms) == gn
                got = np.array([norms[n] for n in gn])
                want = gold[f"momnorm/{key}"]
                rel = np.abs(got - want) / (np.abs(want) + 1e-3 * np.abs(want).max())
                assert rel.max() <= 3e-3, (key, gn[int(rel.argmax())], rel.max())
                # the optimizer UPDATE of the first step, pinned by <p1 - p0, exp_avg> per tensor: entries whose
                # gradient is rounding noise (where Adam's first step is +-lr by sign) carry no weight here
                dots = {names[p]: float(((p.detach() - p0[key][names[p]]).double() * s["exp_avg"].double()).sum())
                        for p, s in opt[key].state.items()}
                got = np.array([dots[n] for n in gn])
                want = gold[f"upddot/{key}"]
                rel = np.abs(got - want) / (np.abs(want) + 1e-3 * np.abs(want).max())
                assert rel.max() <= 5e-3, (key, "upddot", gn[int(rel.argmax())], rel.max())
    for key, tag in (("generator", "g"), ("discriminator", "d")):
        sd = model[key].state_dict()
        names = [str(n) for n in gold[f"final_names/{tag}"]]
        assert sorted(sd) == names
        got = np.array([float(sd[n].double().abs().sum()) for n in names])
        want = gold[f"final_abs/{tag}"]
        assert (np.abs(got - want) / (want + 1e-9)).max() <= final_rtol


def test_pwg_v1_two_train_steps(device):
    gold = load_golden("pwg_v1_train")
    frames, n_steps, seed = (int(v) for v in gold["meta"])
    g = models.ParallelWaveGANGenerator(**PWG_G)
    d = models.ParallelWaveGANDiscriminator(**PWG_D)
    g.load_state_dict(synth.synth_state_dict(g.state_dict(), seed=seed, g_scale=synth.PWG_G_SCALE))
    d.load_state_dict(synth.synth_state_dict(d.state_dict(), seed=seed + 1, g_scale=1.4))
    model = {"generator": g.to(device), "discriminator": d.to(device)}
    criterion = {"gen_adv": losses.GeneratorAdversarialLoss(), "dis_adv": losses.DiscriminatorAdversarialLoss(),
                 "stft": losses.MultiResolutionSTFTLoss(fft_sizes=[1024, 2048, 512], hop_sizes=[120, 240, 50],
                                                        win_lengths=[600, 1200, 240], window="hann_window").to(device)}
    opt = {"generator": optimizers.RAdam(model["generator"].parameters(), lr=1e-4, eps=1e-6, weight_decay=0.0),
           "discriminator": optimizers.RAdam(model["discriminator"].parameters(), lr=5e-5, eps=1e-6, weight_decay=0.0)}
    sched = {k: optimizers.lr_scheduler.StepLR(opt[k], step_size=200000, gamma=0.5) for k in opt}
    config = dict(generator_type="ParallelWaveGANGenerator", generator_params=PWG_G, use_stft_loss=True,
                  use_subband_stft_loss=False, use_mel_loss=False, use_feat_match_loss=False, lambda_adv=4.0,
                  generator_grad_norm=10, discriminator_grad_norm=1, discriminator_train_start_steps=0,
                  train_max_steps=1 + n_steps, save_interval_steps=10 ** 9, eval_interval_steps=10 ** 9,
                  log_interval_steps=10 ** 9, distributed=False, rank=0, outdir=tempfile.mkdtemp(), progress=False)
    c = synth.synth_input("c", (2, 80, frames + 4), seed=seed)
    z = synth.synth_input("z", (2, 1, frames * 256), seed=seed)
    y = 0.5 * synth.synth_input("y", (2, 1, frames * 256), seed=seed)
    batches = [((z, c), y)] * n_steps
    tr = Trainer(steps=1, epochs=0, data_loader={"train": batches, "dev": batches}, sampler={"train": None, "dev": None},
                 model=model, criterion=criterion, optimizer=opt, scheduler=sched, config=config, device=device)
    _run_and_compare(tr, batches, gold, model, opt, 0.1)


# Calibration: two CPU runs of the REFERENCE Trainer on this configuration (3 vs 8 threads) differ by
# 2.0 % in the step-1 fake loss and 9e-4 in the step-1 discriminator loss (Adam with lr 1e-3 turns
# rounding noise of near-zero gradients into +-lr parameter steps; the fake loss is a small
# difference-sensitive quantity); every other logged value agrees to < 1e-5.  The bars for these two
# values are 3x the reference's own run-to-run spread (measured HIP deviation: 3.1e-2 and 1.4e-3; every
# other step-1 value <= 8e-5).  What pins the discriminator UPDATE itself, free of that sign noise, is
# the per-tensor <p1 - p0, exp_avg> check in _run_and_compare (bar 5e-3) together with the per-tensor
# first-moment norms (3e-3): entries whose gradient is rounding noise carry no weight in either.
MB_LOOSE = {(1, "train/fake_loss"): 6e-2, (1, "train/discriminator_loss"): 3e-3}


def test_mb_melgan_v2_two_train_steps(device):
    gold = load_golden("mb_melgan_v2_train")
    frames, n_steps, seed = (int(v) for v in gold["meta"])
    g = models.MelGANGenerator(**MB_G)
    d = models.MelGANMultiScaleDiscriminator(**MB_D)
    g.load_state_dict(synth.synth_state_dict(g.state_dict(), seed=seed, g_scale=synth.MELGAN_G_SCALE))
    d.load_state_dict(synth.synth_state_dict(d.state_dict(), seed=seed + 1, g_scale=1.2))
    model = {"generator": g.to(device), "discriminator": d.to(device)}
    criterion = {"gen_adv": losses.GeneratorAdversarialLoss(), "dis_adv": losses.DiscriminatorAdversarialLoss(),
                 "stft": losses.MultiResolutionSTFTLoss(fft_sizes=[1024, 2048, 512], hop_sizes=[120, 240, 50],
                                                        win_lengths=[600, 1200, 240]).to(device),
                 "sub_stft": losses.MultiResolutionSTFTLoss(fft_sizes=[384, 683, 171], hop_sizes=[30, 60, 10],
                                                            win_lengths=[150, 300, 60]).to(device),
                 "pqmf": layers.PQMF(subbands=4).to(device)}
    opt = {k: optimizers.Adam(model[k].parameters(), lr=1e-3, eps=1e-7, weight_decay=0.0, amsgrad=True) for k in model}
    sched = {k: optimizers.lr_scheduler.MultiStepLR(opt[k], gamma=0.5, milestones=[100000, 200000]) for k in opt}
    config = dict(generator_type="MelGANGenerator", generator_params=MB_G, use_stft_loss=True,
                  use_subband_stft_loss=True, use_mel_loss=False, use_feat_match_loss=False, lambda_adv=2.5,
                  generator_grad_norm=-1, discriminator_grad_norm=-1, discriminator_train_start_steps=0,
                  train_max_steps=1 + n_steps, save_interval_steps=10 ** 9, eval_interval_steps=10 ** 9,
                  log_interval_steps=10 ** 9, distributed=False, rank=0, outdir=tempfile.mkdtemp(), progress=False)
    c = synth.synth_input("c", (2, 80, frames), seed=seed)
    y = 0.5 * synth.synth_input("y", (2, 1, frames * 256), seed=seed)
    batches = [((c,), y)] * n_steps
    tr = Trainer(steps=1, epochs=0, data_loader={"train": batches, "dev": batches}, sampler={"train": None, "dev": None},
                 model=model, criterion=criterion, optimizer=opt, scheduler=sched, config=config, device=device)
    _run_and_compare(tr, batches, gold, model, opt, 0.1, loose=MB_LOOSE, final_rtol=1e-2)  # lr 1e-3 sign noise on small biases


def test_pwg_hip_graph_with_grad_clipping_matches_eager(device):
    """C2 clips gradients (generator_grad_norm 10, discriminator_grad_norm 1): the clip launches are part of
    the captured step, and the replayed step follows the eager one."""
    small = dict(PWG_G, layers=6, stacks=2)
    logs = {}
    for use_graph in (False, True):
        g = models.ParallelWaveGANGenerator(**small)
        d = models.ParallelWaveGANDiscriminator(**PWG_D)
        g.load_state_dict(synth.synth_state_dict(g.state_dict(), seed=5, g_scale=synth.PWG_G_SCALE))
        d.load_state_dict(synth.synth_state_dict(d.state_dict(), seed=6, g_scale=1.4))
        model = {"generator": g.to(device), "discriminator": d.to(device)}
        criterion = {"gen_adv": losses.GeneratorAdversarialLoss(), "dis_adv": losses.DiscriminatorAdversarialLoss(),
                     "stft": losses.MultiResolutionSTFTLoss().to(device)}
        opt = {"generator": optimizers.RAdam(model["generator"].parameters(), lr=1e-4, eps=1e-6),
               "discriminator": optimizers.RAdam(model["discriminator"].parameters(), lr=5e-5, eps=1e-6)}
        sched = {k: optimizers.lr_scheduler.StepLR(opt[k], step_size=200000, gamma=0.5) for k in opt}
        config = dict(generator_type="ParallelWaveGANGenerator", generator_params=small, use_stft_loss=True,
                      use_subband_stft_loss=False, use_mel_loss=False, use_feat_match_loss=False, lambda_adv=4.0,
                      generator_grad_norm=0.5, discriminator_grad_norm=0.05, discriminator_train_start_steps=0,
                      train_max_steps=100, save_interval_steps=10 ** 9, eval_interval_steps=10 ** 9,
                      log_interval_steps=10 ** 9, distributed=False, rank=0, outdir=tempfile.mkdtemp(), progress=False,
                      use_hip_graph=use_graph, graph_warmup_steps=2)
        c = synth.synth_input("c", (2, 80, 12 + 4), seed=5)
        z = synth.synth_input("z", (2, 1, 12 * 256), seed=5)
        y = 0.5 * synth.synth_input("y", (2, 1, 12 * 256), seed=5)
        batch = ((z, c), y)
        tr = Trainer(steps=1, epochs=0, data_loader={"train": [batch], "dev": [batch]},
                     sampler={"train": None, "dev": None}, model=model, criterion=criterion, optimizer=opt,
                     scheduler=sched, config=config, device=device)
        tr.tqdm = None
        log = []
        for _ in range(6):
            tr._train_step(batch)
            tr._flush_pending()
            log.append(dict(tr.total_train_loss))
        if use_graph:
            assert len(tr._graphs) == 1
        logs[use_graph] = log
    for i, (a, b) in enumerate(zip(logs[False], logs[True])):
        for k in a:
            assert abs(a[k] - b[k]) <= 2e-4 * max(abs(a[k]), 1e-3), (i, k, a[k], b[k])
