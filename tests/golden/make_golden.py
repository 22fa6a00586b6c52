"""Generate golden vectors by running the UNMODIFIED reference (/root/reference)
on CPU through oracle/ref_shim.py.  Run in the build container only:

    python tests/golden/make_golden.py [names...]

Inputs and weights are regenerated from seeds by tests/golden/synth.py, so only
the reference's outputs are stored (float32 .npz, a few hundred KB in total).
"""
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402
from tests.golden import synth  # noqa: E402

OUT = os.environ.get("GOLDEN_OUT", os.path.dirname(os.path.abspath(__file__)))  # (GOLDEN_OUT: calibration runs)
warnings.filterwarnings("ignore")
G_SCALE = 1.25


def _stats(t):
    t = t.double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


def hifigan_generator(name, cfg, batch, frames, seed):
    import parallel_wavegan.models as RM

    g = RM.HiFiGANGenerator(**cfg).eval()
    sd = synth.synth_state_dict(g.state_dict(), seed=seed, g_scale=G_SCALE)
    g.load_state_dict(sd)
    c = synth.synth_input("c", (batch, cfg["in_channels"], frames), seed=seed)
    feats = []
    hooks = [g.input_conv.register_forward_hook(lambda m, i, o: feats.append(o))]
    for up in g.upsamples:
        hooks.append(up.register_forward_hook(lambda m, i, o: feats.append(o)))
    pre = []
    hooks.append(g.output_conv[1].register_forward_hook(lambda m, i, o: pre.append(o)))
    with torch.no_grad():
        y = g(c)
        # same weights after remove_weight_norm + the (T, C) inference API
        g.remove_weight_norm()
        y_inf = g.inference(c[0].transpose(0, 1).numpy())
    for h in hooks:
        h.remove()
    np.savez_compressed(
        os.path.join(OUT, f"{name}.npz"),
        y=y.numpy(),
        y_inference=y_inf.numpy(),
        y_pre_tanh=pre[0].numpy(),
        input_conv_head=feats[0][:, :8, :].numpy(),
        upsample_stats=np.stack([_stats(f) for f in feats[1:]]),
        upsample0_head=feats[1][:, :4, :64].numpy(),
        meta=np.array([batch, frames, seed], dtype=np.int64),
        g_scale=np.float64(G_SCALE),
    )
    print(name, "y", tuple(y.shape), "std %.4f max %.4f pre-tanh std %.3f" % (y.std().item(), y.abs().max().item(), pre[0].std().item()))


def _load_yaml(name, corpus="ljspeech"):
    import yaml

    with open(os.path.join(ref_shim.REF_ROOT, "egs", corpus, "voc1", "conf", name)) as f:
        return yaml.safe_load(f)


def hifigan_discriminator(name, seed):
    """MSD+MPD outputs in eval mode and after two training-mode calls (spectral-norm state)."""
    import parallel_wavegan.models as RM

    cfg = _load_yaml("hifigan.v1.yaml")
    d = RM.HiFiGANMultiScaleMultiPeriodDiscriminator(**cfg["discriminator_params"])
    d.load_state_dict(synth.synth_state_dict(d.state_dict(), seed=seed, g_scale=1.0))
    x = 0.5 * synth.synth_input("wave", (2, 1, 8192), seed=seed)
    out = {}
    with torch.no_grad():
        d.eval()
        o = d(x)
        out["eval_logits"] = np.concatenate([t[-1].reshape(-1).numpy() for t in o])
        out["eval_feat_stats"] = np.stack([_stats(f) for t in o for f in t])
        d.train()
        d(x)
        o = d(x)
        out["train2_logits"] = np.concatenate([t[-1].reshape(-1).numpy() for t in o])
        out["train2_feat_stats"] = np.stack([_stats(f) for t in o for f in t])
        out["train2_u0"] = d.msd.discriminators[0].layers[1][0].weight_u.numpy()
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), meta=np.array([2, 8192, seed]), **out)
    print(name, "logits", out["eval_logits"].shape, float(np.abs(out["eval_logits"]).max()))


def losses(name, seed):
    """Spectral / adversarial / feature-matching losses and their input gradients."""
    import parallel_wavegan.losses as RL

    cfg = _load_yaml("hifigan.v1.yaml")
    out = {}
    y = 0.5 * synth.synth_input("y", (2, 1, 8192), seed=seed)
    yh = (0.5 * synth.synth_input("yh", (2, 1, 8192), seed=seed)).requires_grad_()
    mel = RL.MelSpectrogramLoss(**cfg["mel_loss_params"])
    l = mel(yh, y)
    l.backward()
    out["mel_loss"], out["mel_grad"] = l.item(), yh.grad.numpy().copy()
    out["mel_spec"] = mel.mel_spectrogram(y).detach().numpy()
    # LibriTTS mel loss (n_fft 2048, hop 300, win 1200)
    mel2 = RL.MelSpectrogramLoss(fs=24000, fft_size=2048, hop_size=300, win_length=1200, window="hann", num_mels=80,
                                 fmin=0, fmax=12000, log_base=None)
    yh.grad = None
    t84 = slice(0, 8100)
    l = mel2(yh[..., t84], y[..., t84])
    l.backward()
    out["mel2_loss"], out["mel2_grad"] = l.item(), yh.grad.numpy().copy()
    # full-band multi-resolution STFT loss (PWG / MB-MelGAN) and the sub-band variant
    for tag, kw, shape in [("stft", dict(), (2, 6000)),
                           ("substft", dict(fft_sizes=[384, 683, 171], hop_sizes=[30, 60, 10],
                                            win_lengths=[150, 300, 60]), (2, 4, 1500))]:
        st = RL.MultiResolutionSTFTLoss(**kw)
        a = (0.5 * synth.synth_input(tag + "x", shape, seed=seed)).requires_grad_()
        b = 0.5 * synth.synth_input(tag + "y", shape, seed=seed)
        sc, mag = st(a, b)
        (sc + mag).backward()
        out[tag + "_sc"], out[tag + "_mag"], out[tag + "_grad"] = sc.item(), mag.item(), a.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), meta=np.array([seed]), **out)
    print(name, {k: (v if np.isscalar(v) else v.shape) for k, v in out.items()})


def adv_losses(name, seed):
    """GeneratorAdversarialLoss / DiscriminatorAdversarialLoss (mse AND hinge) and FeatureMatchLoss of the
    reference: values and gradients w.r.t. the logits / feature maps, every averaging switch."""
    import parallel_wavegan.losses as RL

    out = {}
    real = synth.adv_logits(seed)
    for lt in ("mse", "hinge"):
        for avg in (True, False):
            fake = [[t.clone().requires_grad_() for t in o] for o in synth.adv_logits(seed + 1)]
            g = RL.GeneratorAdversarialLoss(average_by_discriminators=avg, loss_type=lt)(fake)
            g.backward()
            tag = f"{lt}_{int(avg)}"
            out[f"gen_{tag}"] = g.item()
            out[f"gen_{tag}_grad"] = np.concatenate([o[-1].grad.numpy().ravel() for o in fake])
            fake = [[t.clone().requires_grad_() for t in o] for o in synth.adv_logits(seed + 1)]
            realg = [[t.clone().requires_grad_() for t in o] for o in real]
            r, f = RL.DiscriminatorAdversarialLoss(average_by_discriminators=avg, loss_type=lt)(fake, realg)
            (r + 2.0 * f).backward()
            out[f"dis_real_{tag}"], out[f"dis_fake_{tag}"] = r.item(), f.item()
            out[f"dis_{tag}_grad_real"] = np.concatenate([o[-1].grad.numpy().ravel() for o in realg])
            out[f"dis_{tag}_grad_fake"] = np.concatenate([o[-1].grad.numpy().ravel() for o in fake])
    for al in (True, False):
        for ad in (True, False):
            for fin in (True, False):
                fake = [[t.clone().requires_grad_() for t in o] for o in synth.adv_logits(seed + 1)]
                fm = RL.FeatureMatchLoss(average_by_layers=al, average_by_discriminators=ad,
                                         include_final_outputs=fin)(fake, real)
                fm.backward()
                tag = f"{int(al)}{int(ad)}{int(fin)}"
                out[f"fm_{tag}"] = fm.item()
                out[f"fm_{tag}_grad"] = np.concatenate(
                    [(t.grad if t.grad is not None else torch.zeros_like(t)).numpy().ravel() for o in fake for t in o])
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), meta=np.array([seed]), **out)
    print(name, len(out), "entries")


def hifigan_train_steps(name, seed, batch=2, n_steps=2):
    """Two full ``Trainer._train_step`` calls of the reference (HiFi-GAN V1 YAML, G+D active)."""
    import tempfile

    import parallel_wavegan.losses as RL
    import parallel_wavegan.models as RM
    from parallel_wavegan.bin.train import Trainer

    cfg = _load_yaml("hifigan.v1.yaml")
    g = RM.HiFiGANGenerator(**cfg["generator_params"])
    d = RM.HiFiGANMultiScaleMultiPeriodDiscriminator(**cfg["discriminator_params"])
    g.load_state_dict(synth.synth_state_dict(g.state_dict(), seed=seed, g_scale=G_SCALE))
    d.load_state_dict(synth.synth_state_dict(d.state_dict(), seed=seed + 1, g_scale=1.0))
    model = {"generator": g, "discriminator": d}
    criterion = {
        "gen_adv": RL.GeneratorAdversarialLoss(**cfg["generator_adv_loss_params"]),
        "dis_adv": RL.DiscriminatorAdversarialLoss(**cfg["discriminator_adv_loss_params"]),
        "mel": RL.MelSpectrogramLoss(**cfg["mel_loss_params"]),
        "feat_match": RL.FeatureMatchLoss(**cfg["feat_match_loss_params"]),
    }
    optimizer = {
        "generator": torch.optim.Adam(g.parameters(), **cfg["generator_optimizer_params"]),
        "discriminator": torch.optim.Adam(d.parameters(), **cfg["discriminator_optimizer_params"]),
    }
    scheduler = {
        k: torch.optim.lr_scheduler.MultiStepLR(optimizer[k], **cfg[f"{k}_scheduler_params"])
        for k in ("generator", "discriminator")
    }
    c = synth.synth_input("c", (batch, 80, 32), seed=seed)
    y = 0.5 * synth.synth_input("y", (batch, 1, 8192), seed=seed)
    batches = [((c,), y)] * n_steps
    cfg.update(distributed=False, rank=0, outdir=tempfile.mkdtemp(), train_max_steps=2 + n_steps,
               save_interval_steps=10 ** 9, eval_interval_steps=10 ** 9, log_interval_steps=10 ** 9,
               use_stft_loss=False, use_subband_stft_loss=False)
    tr = Trainer(steps=2, epochs=0, data_loader={"train": batches, "dev": batches},
                 sampler={"train": None, "dev": None}, model=model, criterion=criterion, optimizer=optimizer,
                 scheduler=scheduler, config=cfg, device=torch.device("cpu"))
    out = {}
    from tqdm import tqdm

    tr.tqdm = tqdm(disable=True)
    prev = {}
    for i in range(n_steps):
        tr._train_step(batches[i])
        cur = dict(tr.total_train_loss)
        for k, v in cur.items():
            out[f"step{i}/{k}"] = v - prev.get(k, 0.0)
        prev = cur
        if i == 0:
            # first moments after step 1 = (1 - beta1) * gradient: pins every parameter gradient
            for key in ("generator", "discriminator"):
                st = optimizer[key].state
                names = {p: n for n, p in model[key].named_parameters()}
                norms = {names[p]: float(s["exp_avg"].double().norm()) for p, s in st.items()}
                out[f"gradnorm_names/{key}"] = np.array(sorted(norms))
                out[f"gradnorm/{key}"] = np.array([norms[k] for k in sorted(norms)])
            out["grad/g/output_conv.1.weight_v"] = optimizer["generator"].state[g.output_conv[1].weight_v]["exp_avg"].numpy().copy()
            out["grad/g/input_conv.bias"] = optimizer["generator"].state[g.input_conv.bias]["exp_avg"].numpy().copy()
            out["grad/d/msd.discriminators.0.layers.0.0.weight_orig"] = optimizer["discriminator"].state[
                d.msd.discriminators[0].layers[0][0].weight_orig]["exp_avg"].numpy().copy()
            out["grad/d/mpd.discriminators.4.convs.0.0.weight_v"] = optimizer["discriminator"].state[
                d.mpd.discriminators[4].convs[0][0].weight_v]["exp_avg"].numpy().copy()
    for key, m in (("g", g), ("d", d)):
        sd = m.state_dict()
        names = sorted(sd)
        out[f"final_names/{key}"] = np.array(names)
        out[f"final_sum/{key}"] = np.array([float(sd[n].double().sum()) for n in names])
    out["final/g/output_conv.1.weight_v"] = g.output_conv[1].weight_v.detach().numpy().copy()
    out["final/g/input_conv.bias"] = g.input_conv.bias.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), meta=np.array([batch, n_steps, seed]), g_scale=np.float64(G_SCALE), **out)
    print(name, {k: round(float(v), 6) for k, v in out.items() if k.startswith("step")})


def pwg(name, seed):
    """PWG.v1 generator (config C1: 80 x 100 mel, + 2 context frames each side) and discriminator."""
    import parallel_wavegan.models as RM

    cfg = _load_yaml("parallel_wavegan.v1.yaml")
    out = {}
    g = RM.ParallelWaveGANGenerator(**cfg["generator_params"]).eval()
    g.load_state_dict(synth.synth_state_dict(g.state_dict(), seed=seed, g_scale=synth.PWG_G_SCALE))
    frames = 100
    c = synth.synth_input("c", (1, 80, frames + 4), seed=seed)
    z = synth.synth_input("z", (1, 1, frames * 256), seed=seed)
    with torch.no_grad():
        y = g(z, c)
        # inference API: (T', C) mel without context (ReplicationPad inside), explicit noise x
        y_inf = g.inference(c=c[0, :, 2:-2].transpose(0, 1).numpy(), x=z[0].transpose(0, 1).numpy())
    out["g_y"], out["g_y_inference"] = y.numpy(), y_inf.numpy()
    d = RM.ParallelWaveGANDiscriminator(**cfg["discriminator_params"]).eval()
    d.load_state_dict(synth.synth_state_dict(d.state_dict(), seed=seed + 1, g_scale=1.4))
    x = 0.5 * synth.synth_input("wave", (2, 1, 6000), seed=seed)
    with torch.no_grad():
        out["d_y"] = d(x).numpy()
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), meta=np.array([frames, seed]), **out)
    print(name, "G std %.4f max %.4f | D std %.4f" % (y.std().item(), y.abs().max().item(), out["d_y"].std()))


def pwg_melgan_upsampler(name, seed):
    """ParallelWaveGANGenerator(upsample_net="MelGANGenerator") (models/parallel_wavegan.py:90-98), small."""
    import copy

    import parallel_wavegan.models as RM

    g = RM.ParallelWaveGANGenerator(**copy.deepcopy(synth.PWG_MELGAN_UPSAMPLER))
    g.load_state_dict(synth.synth_state_dict(g.state_dict(), seed=seed, g_scale=synth.PWG_G_SCALE))
    g.eval()
    c = synth.synth_input("c", (2, 80, 9), seed=seed)
    z = synth.synth_input("z", (2, 1, 9 * 256), seed=seed)
    with torch.no_grad():
        y = g(z, c)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), meta=np.array([seed]), y=y.numpy())
    print(name, tuple(y.shape), float(y.abs().max()))


def mb_melgan(name, seed):
    """Multi-band MelGAN.v2 generator (+PQMF synthesis), MelGAN multi-scale discriminator, PQMF."""
    import parallel_wavegan.layers as RLy
    import parallel_wavegan.models as RM

    cfg = _load_yaml("multi_band_melgan.v2.yaml")
    out = {}
    g = RM.MelGANGenerator(**cfg["generator_params"]).eval()
    g.load_state_dict(synth.synth_state_dict(g.state_dict(), seed=seed, g_scale=synth.MELGAN_G_SCALE))
    c = synth.synth_input("c", (2, 80, 24), seed=seed)
    pqmf = RLy.PQMF()
    with torch.no_grad():
        y_mb = g(c)
        out["g_y_mb"] = y_mb.numpy()
        out["g_y_full"] = pqmf.synthesis(y_mb).numpy()
        g.pqmf = pqmf
        out["g_y_inference"] = g.inference(c[0].transpose(0, 1).numpy()).numpy()
        w = 0.5 * synth.synth_input("wave", (2, 1, 4096), seed=seed)
        out["pqmf_analysis"] = pqmf.analysis(w).numpy()
        out["pqmf_round_trip"] = pqmf.synthesis(pqmf.analysis(w)).numpy()
    d = RM.MelGANMultiScaleDiscriminator(**cfg["discriminator_params"]).eval()
    d.load_state_dict(synth.synth_state_dict(d.state_dict(), seed=seed + 1, g_scale=1.2))
    with torch.no_grad():
        o = d(w)
    out["d_logits"] = np.concatenate([t[-1].reshape(-1).numpy() for t in o])
    out["d_feat_stats"] = np.stack([_stats(f) for t in o for f in t])
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), meta=np.array([seed]), **out)
    print(name, "G mb std %.4f max %.4f | D logits std %.4f" % (y_mb.std().item(), y_mb.abs().max().item(), out["d_logits"].std()))


def _run_reference_trainer(cfg, model, criterion, optimizer, scheduler, batches, start_step):
    import tempfile

    from parallel_wavegan.bin.train import Trainer
    from tqdm import tqdm

    cfg.update(distributed=False, rank=0, outdir=tempfile.mkdtemp(), train_max_steps=start_step + len(batches),
               save_interval_steps=10 ** 9, eval_interval_steps=10 ** 9, log_interval_steps=10 ** 9)
    tr = Trainer(steps=start_step, epochs=0, data_loader={"train": batches, "dev": batches},
                 sampler={"train": None, "dev": None}, model=model, criterion=criterion, optimizer=optimizer,
                 scheduler=scheduler, config=cfg, device=torch.device("cpu"))
    tr.tqdm = tqdm(disable=True)
    out, prev = {}, {}
    p0 = {key: {n: p.detach().clone() for n, p in model[key].named_parameters()} for key in model}
    for i, b in enumerate(batches):
        tr._train_step(b)
        cur = dict(tr.total_train_loss)
        for k, v in cur.items():
            out[f"step{i}/{k}"] = v - prev.get(k, 0.0)
        prev = cur
        if i == 0:
            for key in ("generator", "discriminator"):
                names = {p: n for n, p in model[key].named_parameters()}
                norms = {names[p]: float(s["exp_avg"].double().norm()) for p, s in optimizer[key].state.items()}
                out[f"momnorm_names/{key}"] = np.array(sorted(norms))
                out[f"momnorm/{key}"] = np.array([norms[k] for k in sorted(norms)])
                # <first update, first moment> per tensor: pins the optimizer update without being limited by
                # the sign noise of near-zero gradient entries (they carry no weight in this inner product)
                dots = {names[p]: float(((p.detach() - p0[key][names[p]]).double() * s["exp_avg"].double()).sum())
                        for p, s in optimizer[key].state.items()}
                out[f"upddot/{key}"] = np.array([dots[k] for k in sorted(dots)])
    for key, tag in (("generator", "g"), ("discriminator", "d")):
        sd = model[key].state_dict()
        names = sorted(sd)
        out[f"final_names/{tag}"] = np.array(names)
        out[f"final_sum/{tag}"] = np.array([float(sd[n].double().sum()) for n in names])
        out[f"final_abs/{tag}"] = np.array([float(sd[n].double().abs().sum()) for n in names])
    return out


def pwg_train_steps(name, seed, n_steps=2):
    """Two ``Trainer._train_step`` calls of the reference with parallel_wavegan.v1.yaml (RAdam, StepLR,
    grad clipping 10 / 1, multi-resolution STFT loss, D active from the first step)."""
    import parallel_wavegan.losses as RL
    import parallel_wavegan.models as RM
    from parallel_wavegan.optimizers import RAdam

    cfg = _load_yaml("parallel_wavegan.v1.yaml")
    cfg["discriminator_train_start_steps"] = 0
    g = RM.ParallelWaveGANGenerator(**cfg["generator_params"])
    d = RM.ParallelWaveGANDiscriminator(**cfg["discriminator_params"])
    g.load_state_dict(synth.synth_state_dict(g.state_dict(), seed=seed, g_scale=synth.PWG_G_SCALE))
    d.load_state_dict(synth.synth_state_dict(d.state_dict(), seed=seed + 1, g_scale=1.4))
    model = {"generator": g, "discriminator": d}
    criterion = {"gen_adv": RL.GeneratorAdversarialLoss(), "dis_adv": RL.DiscriminatorAdversarialLoss(),
                 "stft": RL.MultiResolutionSTFTLoss(**cfg["stft_loss_params"])}
    optimizer = {"generator": RAdam(g.parameters(), **cfg["generator_optimizer_params"]),
                 "discriminator": RAdam(d.parameters(), **cfg["discriminator_optimizer_params"])}
    scheduler = {k: torch.optim.lr_scheduler.StepLR(optimizer[k], **cfg[f"{k}_scheduler_params"])
                 for k in ("generator", "discriminator")}
    frames = 10
    c = synth.synth_input("c", (2, 80, frames + 4), seed=seed)
    z = synth.synth_input("z", (2, 1, frames * 256), seed=seed)
    y = 0.5 * synth.synth_input("y", (2, 1, frames * 256), seed=seed)
    cfg.update(use_stft_loss=True, use_subband_stft_loss=False, use_mel_loss=False, use_feat_match_loss=False)
    out = _run_reference_trainer(cfg, model, criterion, optimizer, scheduler, [((z, c), y)] * n_steps, 1)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), meta=np.array([frames, n_steps, seed]), **out)
    print(name, {k: round(float(v), 6) for k, v in out.items() if k.startswith("step")})


def mb_melgan_train_steps(name, seed, n_steps=2):
    """Two reference Trainer steps with multi_band_melgan.v2.yaml (Adam amsgrad eps 1e-7, full-band +
    sub-band STFT losses through PQMF, MelGAN MSD active from the first step)."""
    import parallel_wavegan.layers as RLy
    import parallel_wavegan.losses as RL
    import parallel_wavegan.models as RM

    cfg = _load_yaml("multi_band_melgan.v2.yaml")
    cfg["discriminator_train_start_steps"] = 0
    g = RM.MelGANGenerator(**cfg["generator_params"])
    d = RM.MelGANMultiScaleDiscriminator(**cfg["discriminator_params"])
    g.load_state_dict(synth.synth_state_dict(g.state_dict(), seed=seed, g_scale=synth.MELGAN_G_SCALE))
    d.load_state_dict(synth.synth_state_dict(d.state_dict(), seed=seed + 1, g_scale=1.2))
    model = {"generator": g, "discriminator": d}
    criterion = {"gen_adv": RL.GeneratorAdversarialLoss(), "dis_adv": RL.DiscriminatorAdversarialLoss(),
                 "stft": RL.MultiResolutionSTFTLoss(**cfg["stft_loss_params"]),
                 "sub_stft": RL.MultiResolutionSTFTLoss(**cfg["subband_stft_loss_params"]),
                 "pqmf": RLy.PQMF(subbands=cfg["generator_params"]["out_channels"])}
    optimizer = {k: torch.optim.Adam(model[k].parameters(), **cfg[f"{k}_optimizer_params"])
                 for k in ("generator", "discriminator")}
    scheduler = {k: torch.optim.lr_scheduler.MultiStepLR(optimizer[k], **cfg[f"{k}_scheduler_params"])
                 for k in ("generator", "discriminator")}
    c = synth.synth_input("c", (2, 80, 16), seed=seed)
    y = 0.5 * synth.synth_input("y", (2, 1, 4096), seed=seed)
    cfg.update(use_stft_loss=True, use_mel_loss=False)
    out = _run_reference_trainer(cfg, model, criterion, optimizer, scheduler, [((c,), y)] * n_steps, 1)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), meta=np.array([16, n_steps, seed]), **out)
    print(name, {k: round(float(v), 6) for k, v in out.items() if k.startswith("step")})


def train_full_shape(name, tag, seed):
    """ONE ``Trainer._train_step`` of the unmodified reference at a BASELINE configuration's OWN batch shape
    (C2 6 x 25600, C3 16 x 8192, C4 64 x 16384; VERDICT r02 item 1c -- and, round 4, C5 = BASELINE configs[4]:
    HiFi-GAN V1 LibriTTS 24 kHz, egs/libritts/voc1/conf/hifigan.v1.yaml, 16 x 8400, the data-parallel workload): the tile / split-K / slab plans the HIP
    engine selects at these sizes are the ones the benchmark times, and the B = 2 fixtures never exercise them.
    Same stored quantities as ``_run_reference_trainer`` (losses, per-tensor first-moment norms, <update, moment>)."""
    import parallel_wavegan.layers as RLy
    import parallel_wavegan.losses as RL
    import parallel_wavegan.models as RM
    from parallel_wavegan.optimizers import RAdam

    yaml_name = {"c2": "parallel_wavegan.v1.yaml", "c3": "hifigan.v1.yaml", "c4": "multi_band_melgan.v2.yaml",
                 "c5": "hifigan.v1.yaml"}[tag]
    cfg = _load_yaml(yaml_name, corpus="libritts" if tag == "c5" else "ljspeech")
    cfg["discriminator_train_start_steps"] = 0
    cfg["generator_train_start_steps"] = 0
    b, t, hop = cfg["batch_size"], cfg["batch_max_steps"], cfg["hop_size"]
    gcls = getattr(RM, cfg.get("generator_type", "ParallelWaveGANGenerator"))
    dcls = getattr(RM, cfg.get("discriminator_type", "ParallelWaveGANDiscriminator"))
    g, d = gcls(**cfg["generator_params"]), dcls(**cfg["discriminator_params"])
    gs, ds = {"c2": (synth.PWG_G_SCALE, 1.4), "c3": (G_SCALE, 1.0), "c4": (synth.MELGAN_G_SCALE, 1.2),
              "c5": (G_SCALE, 1.0)}[tag]
    g.load_state_dict(synth.synth_state_dict(g.state_dict(), seed=seed, g_scale=gs))
    d.load_state_dict(synth.synth_state_dict(d.state_dict(), seed=seed + 1, g_scale=ds))
    model = {"generator": g, "discriminator": d}
    criterion = {"gen_adv": RL.GeneratorAdversarialLoss(**cfg.get("generator_adv_loss_params", {})),
                 "dis_adv": RL.DiscriminatorAdversarialLoss(**cfg.get("discriminator_adv_loss_params", {}))}
    if cfg.get("use_stft_loss", True):
        criterion["stft"] = RL.MultiResolutionSTFTLoss(**cfg["stft_loss_params"])
    if cfg.get("use_subband_stft_loss", False):
        criterion["sub_stft"] = RL.MultiResolutionSTFTLoss(**cfg["subband_stft_loss_params"])
    if cfg["generator_params"]["out_channels"] > 1:
        criterion["pqmf"] = RLy.PQMF(subbands=cfg["generator_params"]["out_channels"])
    if cfg.get("use_mel_loss", False):
        criterion["mel"] = RL.MelSpectrogramLoss(**cfg["mel_loss_params"])
    if cfg.get("use_feat_match_loss", False):
        criterion["feat_match"] = RL.FeatureMatchLoss(**cfg.get("feat_match_loss_params", {}))
    cfg.setdefault("use_stft_loss", True)
    for k in ("use_subband_stft_loss", "use_mel_loss", "use_feat_match_loss"):
        cfg.setdefault(k, False)
    opt_cls = {"RAdam": RAdam, "Adam": torch.optim.Adam}
    optimizer = {k: opt_cls[cfg.get(f"{k}_optimizer_type", "RAdam")](model[k].parameters(), **cfg[f"{k}_optimizer_params"])
                 for k in ("generator", "discriminator")}
    scheduler = {k: getattr(torch.optim.lr_scheduler, cfg.get(f"{k}_scheduler_type", "StepLR"))(
        optimizer[k], **cfg[f"{k}_scheduler_params"]) for k in ("generator", "discriminator")}
    acw = cfg["generator_params"].get("aux_context_window", 0)
    c = synth.synth_input("c", (b, cfg["num_mels"], t // hop + 2 * acw), seed=seed)
    y = 0.5 * synth.synth_input("y", (b, 1, t), seed=seed)
    x = (c,)
    if tag == "c2":
        x = (synth.synth_input("z", (b, 1, t), seed=seed), c)
    import time

    t0 = time.time()
    # The reference forms the spectral-convergence loss with fp32 torch.norm over up to 13 M magnitudes; at these
    # sizes that accumulation alone is 4e-4 .. 6e-4 away from exact arithmetic (the ratio of the two norms
    # 1e-5 .. 2e-4).  Stored next to the reference's own values: the same loss from the reference's (fp32)
    # generator output with the STFT, the clamp, the square root and both norms evaluated in float64.
    sc64 = {}
    if "stft" in criterion:
        def sc_double(crit, xh, yt):
            xh, yt = xh.reshape(-1, xh.size(-1)).double(), yt.reshape(-1, yt.size(-1)).double()
            tot = 0.0
            for f in crit.stft_losses:
                mags = []
                for sig in (xh, yt):
                    sp = torch.stft(sig, f.fft_size, f.shift_size, f.win_length, f.window.double(), return_complex=True)
                    mags.append(torch.sqrt(torch.clamp(sp.real ** 2 + sp.imag ** 2, min=1e-7)))
                tot += float(torch.norm(mags[1] - mags[0], p="fro") / torch.norm(mags[1], p="fro"))
            return tot / len(crit.stft_losses)

        with torch.no_grad():
            y_hat = g(*x)
            if "pqmf" in criterion:
                sc64["sub"] = sc_double(criterion["sub_stft"], y_hat, criterion["pqmf"].analysis(y))
                y_hat = criterion["pqmf"].synthesis(y_hat)
            sc64["full"] = sc_double(criterion["stft"], y_hat.squeeze(1), y.squeeze(1))
    out = _run_reference_trainer(cfg, model, criterion, optimizer, scheduler, [(x, y)], 1)
    for k, v in sc64.items():
        out[f"sc64/{k}"] = np.float64(v)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), meta=np.array([b, t, seed]), **out)
    print(name, f"{time.time() - t0:.0f} s", {k: round(float(v), 6) for k, v in out.items() if k.startswith("step")})


def causal_variants(name, seed):
    """Forward outputs of the use_causal_conv=True generators and of the residual PWG discriminator."""
    import parallel_wavegan.models as RM

    out = {}
    with torch.no_grad():
        g = RM.HiFiGANGenerator(**synth.HIFIGAN_CAUSAL).eval()
        g.load_state_dict(synth.synth_state_dict(g.state_dict(), seed=seed, g_scale=G_SCALE))
        out["hifigan"] = g(synth.synth_input("c", (2, 80, 24), seed=seed)).numpy()
        m = RM.MelGANGenerator(**synth.MELGAN_CAUSAL).eval()
        m.load_state_dict(synth.synth_state_dict(m.state_dict(), seed=seed + 1, g_scale=synth.MELGAN_G_SCALE))
        out["melgan"] = m(synth.synth_input("c", (2, 80, 20), seed=seed + 1)).numpy()
        p = RM.ParallelWaveGANGenerator(**synth.PWG_CAUSAL).eval()
        p.load_state_dict(synth.synth_state_dict(p.state_dict(), seed=seed + 2, g_scale=1.0))
        z = synth.synth_input("z", (2, 1, 18 * 16), seed=seed + 2)
        c = synth.synth_input("c", (2, 80, 18 + 4), seed=seed + 2)
        out["pwg"] = p(z, c).numpy()
        for key, causal in (("res_d", False), ("res_d_causal", True)):
            d = RM.ResidualParallelWaveGANDiscriminator(use_causal_conv=causal, **synth.RESIDUAL_PWG_D).eval()
            d.load_state_dict(synth.synth_state_dict(d.state_dict(), seed=seed + 3, g_scale=1.0))
            out[key] = d(0.5 * synth.synth_input("wave", (2, 1, 700), seed=seed + 3)).numpy()
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), meta=np.array([seed]), g_scale=np.float64(G_SCALE), **out)
    print(name, {k: (v.shape, float(np.abs(v).max())) for k, v in out.items()})


def style_melgan(name, seed):
    """StyleMelGAN: tiny generators (softmax / sigmoid gate), the default generator (88 frames) and the
    random-window discriminator (numpy-seeded window positions, stored)."""
    import parallel_wavegan.models as RM

    out = {}
    with torch.no_grad():
        for key, cfg in (("tiny", synth.STYLE_MELGAN_TINY), ("tiny_sigmoid", synth.STYLE_MELGAN_TINY_SIGMOID)):
            g = RM.StyleMelGANGenerator(**cfg).eval()
            g.load_state_dict(synth.synth_state_dict(g.state_dict(), seed=seed, g_scale=1.1))
            z = synth.synth_input("z", (2, cfg["in_channels"], 5), seed=seed)
            c = synth.synth_input("c", (2, 80, 20), seed=seed)
            out[key] = g(c, z).numpy()
        g = RM.StyleMelGANGenerator().eval()
        # (gain 0.8: at 1.1 the 9-block default net is chaotic at random init -- a 1e-6 relative input
        # change moves the output by 3e-2 -- which would test rounding noise, not the implementation)
        g.load_state_dict(synth.synth_state_dict(g.state_dict(), seed=seed + 1, g_scale=0.8))
        z = synth.synth_input("z", (1, 128, 1), seed=seed + 1)
        c = synth.synth_input("c", (1, 80, 88), seed=seed + 1)
        y = g(c, z)
        out["default_head"] = y[..., :4096].numpy()
        out["default_stats"] = _stats(y)
        d = RM.StyleMelGANDiscriminator(**synth.STYLE_MELGAN_D).eval()
        d.load_state_dict(synth.synth_state_dict(d.state_dict(), seed=seed + 2, g_scale=1.2, skip=synth.PQMF_BUFFERS),
                          strict=False)
        x = 0.5 * synth.synth_input("wave", (2, 1, 8192), seed=seed + 2)
        np.random.seed(seed)
        outs = d(x)
        np.random.seed(seed)
        starts = [np.random.randint(8192 - ws) for _ in range(d.repeats) for ws in d.window_sizes]
        out["d_starts"] = np.array(starts)
        out["d_logits"] = np.stack([o[-1].numpy() for o in outs])          # (8, 2, 1, T'') equal lengths
        out["d_feat_stats"] = np.stack([np.stack([_stats(f) for f in o]) for o in outs])
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), meta=np.array([seed]), **out)
    print(name, {k: getattr(v, "shape", None) for k, v in out.items()}, float(np.abs(out["tiny"]).max()),
          float(np.abs(out["default_head"]).max()), float(np.abs(out["d_logits"]).max()))


def uhifigan(name, seed):
    """UHiFiGAN generator in eval mode (dropout off): c (2, 80, 24), excitation (2, 1, 192)."""
    import parallel_wavegan.models as RM

    g = RM.UHiFiGANGenerator(**synth.UHIFIGAN_TINY).eval()
    gs = 0.6  # keeps the tanh output away from saturation
    g.load_state_dict(synth.synth_state_dict(g.state_dict(), seed=seed, g_scale=gs))
    c = synth.synth_input("c", (2, 80, 24), seed=seed)
    e = synth.synth_input("excitation", (2, 1, 24 * 8), seed=seed)
    with torch.no_grad():
        y = g(c, None, e)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), meta=np.array([seed]), g_scale=np.float64(gs), y=y.numpy())
    print(name, tuple(y.shape), float(y.abs().max()))


def family_train_steps(name, family, seed, n_steps=2):
    """Two ``Trainer._train_step`` calls of the unmodified reference for the families outside BASELINE's configs
    ("their training is smoke only"): StyleMelGAN (TADE generator + random-window
    discriminator: z from torch's CPU generator, window starts from numpy's -- both seeded right before the steps, and
    drawn in the reference's call order) and UHiFiGAN (U-Net generator on (mel, f0, excitation) + the PWG discriminator).
    Adam (lr 5e-4 / 1e-4, betas 0.5 / 0.9), multi-resolution STFT loss 256 / 512, mse adversarial losses."""
    import parallel_wavegan.losses as RL
    import parallel_wavegan.models as RM

    if family == "style_melgan":
        g = RM.StyleMelGANGenerator(**synth.STYLE_MELGAN_TRAIN)
        d = RM.StyleMelGANDiscriminator(**synth.STYLE_MELGAN_TRAIN_D)
        g.load_state_dict(synth.synth_state_dict(g.state_dict(), seed=seed, g_scale=1.1))
        d.load_state_dict(synth.synth_state_dict(d.state_dict(), seed=seed + 1, g_scale=1.2, skip=synth.PQMF_BUFFERS),
                          strict=False)
        c = synth.synth_input("c", (2, 80, 16), seed=seed)
        y = 0.3 * synth.synth_input("y", (2, 1, 16 * 256), seed=seed)
        x, gtype = (c,), "StyleMelGANGenerator"
    else:
        g = RM.UHiFiGANGenerator(**synth.UHIFIGAN_TRAIN)
        d = RM.ParallelWaveGANDiscriminator(layers=4, conv_channels=16)
        g.load_state_dict(synth.synth_state_dict(g.state_dict(), seed=seed, g_scale=0.6))
        d.load_state_dict(synth.synth_state_dict(d.state_dict(), seed=seed + 1, g_scale=1.4))
        frames = 64
        c = synth.synth_input("c", (2, 80, frames), seed=seed)
        f0 = synth.synth_input("f0", (2, 1, frames), seed=seed).abs()
        e = synth.synth_input("excitation", (2, 1, frames * 8), seed=seed)
        y = 0.3 * synth.synth_input("y", (2, 1, frames * 8), seed=seed)
        x, gtype = (c, f0, e), "UHiFiGANGenerator"
    model = {"generator": g, "discriminator": d}
    criterion = {"gen_adv": RL.GeneratorAdversarialLoss(), "dis_adv": RL.DiscriminatorAdversarialLoss(),
                 "stft": RL.MultiResolutionSTFTLoss(**synth.FAMILY_TRAIN_STFT)}
    optimizer = {k: torch.optim.Adam(model[k].parameters(), lr=synth.FAMILY_TRAIN_LR[k], betas=(0.5, 0.9)) for k in model}
    scheduler = {k: torch.optim.lr_scheduler.StepLR(optimizer[k], step_size=10 ** 6, gamma=0.5) for k in model}
    cfg = dict(synth.FAMILY_TRAIN_CFG, generator_type=gtype, generator_params={"out_channels": 1})
    torch.manual_seed(seed)
    np.random.seed(seed)
    out = _run_reference_trainer(cfg, model, criterion, optimizer, scheduler, [(x, y)] * n_steps, 1)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), meta=np.array([n_steps, seed]), **out)
    print(name, {k: round(float(v), 6) for k, v in out.items() if k.startswith("step")})


JOBS = {
    "style_melgan_train": lambda: family_train_steps("style_melgan_train", "style_melgan", 195),
    "uhifigan_train": lambda: family_train_steps("uhifigan_train", "uhifigan", 197),
    "uhifigan": lambda: uhifigan("uhifigan", 97),
    "style_melgan": lambda: style_melgan("style_melgan", 95),
    "causal_variants": lambda: causal_variants("causal_variants", 91),
    "hifigan_v1_g": lambda: hifigan_generator("hifigan_v1_g", synth.HIFIGAN_V1, 2, 32, 11),
    "hifigan_v1_libritts_g": lambda: hifigan_generator("hifigan_v1_libritts_g", synth.HIFIGAN_V1_LIBRITTS, 1, 28, 12),
    "hifigan_tiny_g": lambda: hifigan_generator("hifigan_tiny_g", synth.HIFIGAN_TINY, 3, 21, 13),
    "pwg_v1": lambda: pwg("pwg_v1", 51),
    "pwg_melgan_upsampler": lambda: pwg_melgan_upsampler("pwg_melgan_upsampler", 53),
    "mb_melgan_v2": lambda: mb_melgan("mb_melgan_v2", 61),
    "pwg_v1_train": lambda: pwg_train_steps("pwg_v1_train", 71),
    "mb_melgan_v2_train": lambda: mb_melgan_train_steps("mb_melgan_v2_train", 81),
    "hifigan_v1_d": lambda: hifigan_discriminator("hifigan_v1_d", 21),
    "losses": lambda: losses("losses", 31),
    "adv_losses": lambda: adv_losses("adv_losses", 33),
    "hifigan_v1_train": lambda: hifigan_train_steps("hifigan_v1_train", 41),
    "c2_train_full": lambda: train_full_shape("c2_train_full", "c2", 171),
    "c3_train_full": lambda: train_full_shape("c3_train_full", "c3", 141),
    "c4_train_full": lambda: train_full_shape("c4_train_full", "c4", 181),
    "c5_train_full": lambda: train_full_shape("c5_train_full", "c5", 191),
}

if __name__ == "__main__":
    ref_shim.install()
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", os.cpu_count())))
    names = sys.argv[1:] or list(JOBS)
    for n in names:
        JOBS[n]()
