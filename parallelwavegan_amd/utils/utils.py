"""Inference constructor and stats IO (drop-in subset of parallel_wavegan.utils.utils)."""
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
