"""Inference constructor, training-object factory and stats IO (drop-in subset of
parallel_wavegan.utils.utils plus the object construction of parallel_wavegan.bin.train.main)."""
import logging
import os

import numpy as np
import torch
import yaml


def read_hdf5(hdf5_name, hdf5_path):
    """Read one dataset from an HDF5 file (needs h5py; only used for ``stats.h5``)."""
    try:
        import h5py
    except ImportError as e:  # pragma: no cover
        raise ImportError("h5py is needed to read .h5 statistics; use the .npy form instead") from e
    if not os.path.exists(hdf5_name):
        raise FileNotFoundError(f"There is no such a hdf5 file ({hdf5_name}).")
    with h5py.File(hdf5_name, "r") as f:
        if hdf5_path not in f:
            raise KeyError(f"There is no such a data in hdf5 file. ({hdf5_path})")
        return f[hdf5_path][()]


def load_stats(stats):
    """(mean, scale) from ``stats.h5`` (datasets mean/scale) or ``stats.npy`` ([mean; scale])."""
    assert stats.endswith(".h5") or stats.endswith(".npy")
    if stats.endswith(".h5"):
        return read_hdf5(stats, "mean").reshape(-1), read_hdf5(stats, "scale").reshape(-1)
    arr = np.load(stats)
    return arr[0].reshape(-1), arr[1].reshape(-1)


def load_model(checkpoint, config=None, stats=None):
    """Build the generator named in ``config.yml`` (found beside the checkpoint when not given),
    load ``checkpoint["model"]["generator"]`` and attach stats / PQMF exactly like the reference's
    ``load_model`` (utils/utils.py:294-360), with the generator running on the HIP kernels."""
    if config is None:
        dirname = os.path.dirname(checkpoint)
        with open(os.path.join(dirname, "config.yml")) as f:
            config = yaml.load(f, Loader=yaml.Loader)
    from .. import models
    from ..layers import PQMF

    generator_type = config.get("generator_type", "ParallelWaveGANGenerator")
    generator_params = {k.replace("upsample_kernal_sizes", "upsample_kernel_sizes"): v
                        for k, v in config["generator_params"].items()}  # typo kept for old configs
    if not hasattr(models, generator_type):
        raise NotImplementedError(f"{generator_type} is outside the accelerated hot path (SURVEY.md s2)")
    model = getattr(models, generator_type)(**generator_params)
    model.load_state_dict(torch.load(checkpoint, map_location="cpu")["model"]["generator"])
    if stats is None:
        dirname = os.path.dirname(checkpoint)
        ext = "h5" if config.get("format", "hdf5") == "hdf5" else "npy"
        cand = os.path.join(dirname, f"stats.{ext}")
        stats = cand if os.path.exists(cand) else None
    if stats is not None:
        model.register_stats(stats)
    if generator_params.get("out_channels", 1) > 1:
        pqmf_params = dict(config.get("pqmf_params", {}))
        pqmf_params.setdefault("subbands", generator_params["out_channels"])
        model.pqmf = PQMF(**pqmf_params)
    logging.info(f"Loaded {generator_type} from {checkpoint}.")
    return model


def build_from_config(config, device="cpu", only=None):
    """Everything ``Trainer`` needs, built from a reference training YAML (already parsed into a dict):
    ``(model, criterion, optimizer, scheduler)`` -- each a dict with the keys of the reference's
    ``main()`` (/root/reference/parallel_wavegan/bin/train.py:1364-1493).  The same defaulting rules
    ("keep compatibility": PWG generator / discriminator, RAdam, StepLR, ``use_*_loss`` flags written
    back into ``config``) so that any recipe config that names in-scope classes works unchanged; the
    classes themselves come from this package, i.e. run on the HIP kernels.

    ``only``: optional subset of ("model", "criterion", "optimizer", "scheduler") to build (the rest
    is returned as None); optimizers need the models, schedulers the optimizers."""
    import torch.optim.lr_scheduler as lr_scheduler

    from .. import losses, models, optimizers
    from ..layers import PQMF

    want = set(only) if only is not None else {"model", "criterion", "optimizer", "scheduler"}
    if "scheduler" in want:
        want.add("optimizer")
    if "optimizer" in want:
        want.add("model")
    device = torch.device(device)

    def cls(namespace, name, what):
        if not hasattr(namespace, name):
            raise NotImplementedError(f"{what} {name!r} is outside the accelerated hot path (SURVEY.md s2)")
        return getattr(namespace, name)

    model = criterion = optimizer = scheduler = None
    if "model" in want:
        g_params = {k.replace("upsample_kernal_sizes", "upsample_kernel_sizes"): v
                    for k, v in config["generator_params"].items()}
        model = {
            "generator": cls(models, config.get("generator_type", "ParallelWaveGANGenerator"), "generator")(
                **g_params).to(device),
            "discriminator": cls(models, config.get("discriminator_type", "ParallelWaveGANDiscriminator"),
                                 "discriminator")(**config["discriminator_params"]).to(device),
        }
    if "criterion" in want:
        criterion = {
            "gen_adv": losses.GeneratorAdversarialLoss(**config.get("generator_adv_loss_params", {})).to(device),
            "dis_adv": losses.DiscriminatorAdversarialLoss(**config.get("discriminator_adv_loss_params", {})).to(device),
        }
        # the flags are normalised in place, as the reference does, because Trainer reads them back
        config["use_stft_loss"] = bool(config.get("use_stft_loss", True))
        if config["use_stft_loss"]:
            criterion["stft"] = losses.MultiResolutionSTFTLoss(**config["stft_loss_params"]).to(device)
        config["use_subband_stft_loss"] = bool(config.get("use_subband_stft_loss", False))
        if config["use_subband_stft_loss"]:
            assert config["generator_params"]["out_channels"] > 1
            criterion["sub_stft"] = losses.MultiResolutionSTFTLoss(**config["subband_stft_loss_params"]).to(device)
        config["use_feat_match_loss"] = bool(config.get("use_feat_match_loss", False))
        if config["use_feat_match_loss"]:
            criterion["feat_match"] = losses.FeatureMatchLoss(**config.get("feat_match_loss_params", {})).to(device)
        config["use_mel_loss"] = bool(config.get("use_mel_loss", False))
        if config["use_mel_loss"]:
            mel_params = config.get("mel_loss_params")
            if mel_params is None:  # fall back to the feature-extraction settings of the recipe
                mel_params = dict(fs=config["sampling_rate"], fft_size=config["fft_size"], hop_size=config["hop_size"],
                                  win_length=config["win_length"], window=config["window"],
                                  num_mels=config["num_mels"], fmin=config["fmin"], fmax=config["fmax"])
            criterion["mel"] = losses.MelSpectrogramLoss(**mel_params).to(device)
        if config.get("use_duration_loss", False):
            raise NotImplementedError("duration loss belongs to the discrete-symbol models (out of scope, SURVEY.md s2)")
        config["use_duration_loss"] = False
        if config["generator_params"]["out_channels"] > 1:
            criterion["pqmf"] = PQMF(subbands=config["generator_params"]["out_channels"],
                                     **config.get("pqmf_params", {})).to(device)
    if "optimizer" in want:
        optimizer = {
            k: cls(optimizers, config.get(f"{k}_optimizer_type", "RAdam"), "optimizer")(
                model[k].parameters(), **config[f"{k}_optimizer_params"])
            for k in ("generator", "discriminator")}
    if "scheduler" in want:
        scheduler = {
            k: cls(lr_scheduler, config.get(f"{k}_scheduler_type", "StepLR"), "scheduler")(
                optimizer=optimizer[k], **config[f"{k}_scheduler_params"])
            for k in ("generator", "discriminator")}
    return model, criterion, optimizer, scheduler


def download_pretrained_model(tag_or_url, download_dir=None):
    """Path of the checkpoint of a pretrained model that is ALREADY in the local cache the reference's downloader fills
    (``~/.cache/parallel_wavegan/<tag>/checkpoint*.pkl``, /root/reference/parallel_wavegan/utils/utils.py:363-421).
    Fetching from Google Drive is the reference's distribution plumbing, outside the accelerated path:
    a tag that is not cached raises instead of going to the network."""
    import fnmatch
    import re

    if download_dir is None:
        download_dir = os.path.expanduser("~/.cache/parallel_wavegan")
    tag = tag_or_url
    if "drive.google.com" in tag_or_url:
        ids = re.compile(r"/[-\w]{25,}").findall(tag_or_url)
        if not ids:
            raise ValueError("Unknown URL format. Please use google drive for the model.")
        tag = ids[0][1:]
    root = os.path.join(download_dir, tag)
    found = []
    for base, _, names in os.walk(root):
        found += [os.path.join(base, n) for n in names if fnmatch.fnmatch(n, "checkpoint*.pkl")]
    if not found:
        raise FileNotFoundError(f"pretrained model {tag_or_url!r} is not in the local cache ({root}); download it with the "
                                "reference package's `download_pretrained_model` (this engine does not access the network)")
    return sorted(found)[0]
