"""Run the reference's OWN modules on the host CPU (TEST INFRASTRUCTURE: only ``bench.py``'s ``cpu_baseline`` leg and
``tests/`` may import this).  Resolves ``parallel_wavegan`` through ``oracle.ref_shim`` -- /root/reference in the
build container, the byte-for-byte staged copy ``oracle/_ref`` (``oracle/make_ref.py``) on the GPU box."""
import tempfile

import torch

from . import ref_shim


def available():
    return ref_shim.available()


def generator(generator_type, generator_params):
    """The reference generator as ``bin/decode.py:141-149`` prepares it (weight norm removed, eval mode)."""
    ref_shim.install()
    import parallel_wavegan.models as RM

    g = getattr(RM, generator_type)(**generator_params)
    g.remove_weight_norm()
    return g.eval()


def trainer(conf, batch, device="cpu"):
    """A reference ``Trainer`` (bin/train.py:52-96) on CPU (or, for tools/bench_reference_rocm.py, on ``device``), everything built from the recipe dict the way
    ``bin/train.py:1364-1493`` does, both phases active from the first step."""
    ref_shim.install()
    import parallel_wavegan.layers as RLy
    import parallel_wavegan.losses as RL
    import parallel_wavegan.models as RM
    import parallel_wavegan.optimizers as RO
    from parallel_wavegan.bin.train import Trainer
    from tqdm import tqdm

    cfg = dict(conf)
    gcls = getattr(RM, cfg.get("generator_type", "ParallelWaveGANGenerator"))
    dcls = getattr(RM, cfg.get("discriminator_type", "ParallelWaveGANDiscriminator"))
    device = torch.device(device)
    model = {"generator": gcls(**cfg["generator_params"]).to(device),
             "discriminator": dcls(**cfg["discriminator_params"]).to(device)}
    criterion = {"gen_adv": RL.GeneratorAdversarialLoss(**cfg.get("generator_adv_loss_params", {})),
                 "dis_adv": RL.DiscriminatorAdversarialLoss(**cfg.get("discriminator_adv_loss_params", {}))}
    cfg.setdefault("use_stft_loss", True)
    for k in ("use_subband_stft_loss", "use_mel_loss", "use_feat_match_loss"):
        cfg.setdefault(k, False)
    if cfg["use_stft_loss"]:
        criterion["stft"] = RL.MultiResolutionSTFTLoss(**cfg["stft_loss_params"]).to(device)
    if cfg["use_subband_stft_loss"]:
        criterion["sub_stft"] = RL.MultiResolutionSTFTLoss(**cfg["subband_stft_loss_params"]).to(device)
    if cfg["generator_params"]["out_channels"] > 1:
        criterion["pqmf"] = RLy.PQMF(subbands=cfg["generator_params"]["out_channels"]).to(device)
    if cfg["use_mel_loss"]:
        criterion["mel"] = RL.MelSpectrogramLoss(**cfg["mel_loss_params"]).to(device)
    if cfg["use_feat_match_loss"]:
        criterion["feat_match"] = RL.FeatureMatchLoss(**cfg.get("feat_match_loss_params", {}))
    opt_cls = {"RAdam": RO.RAdam, "Adam": torch.optim.Adam}
    optimizer = {k: opt_cls[cfg.get(f"{k}_optimizer_type", "RAdam")](model[k].parameters(), **cfg[f"{k}_optimizer_params"])
                 for k in model}
    scheduler = {k: getattr(torch.optim.lr_scheduler, cfg.get(f"{k}_scheduler_type", "StepLR"))(
        optimizer[k], **cfg[f"{k}_scheduler_params"]) for k in model}
    cfg.update(distributed=False, rank=0, outdir=tempfile.mkdtemp(), train_max_steps=10 ** 9, save_interval_steps=10 ** 9,
               eval_interval_steps=10 ** 9, log_interval_steps=10 ** 9, generator_train_start_steps=0,
               discriminator_train_start_steps=0)
    tr = Trainer(steps=1, epochs=0, data_loader={"train": [batch], "dev": [batch]}, sampler={"train": None, "dev": None},
                 model=model, criterion=criterion, optimizer=optimizer, scheduler=scheduler, config=cfg,
                 device=device)
    tr.tqdm = tqdm(disable=True)
    return tr
