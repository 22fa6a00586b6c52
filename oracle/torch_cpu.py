"""Functional CPU restatement of the reference's hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py): imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.

The reference is pure Python over ATen ops, so each function here is the
reference's forward written as a flat sequence of ``torch.nn.functional`` calls
on CPU tensors, taking a *reference-format* ``state_dict`` (``weight_g`` /
``weight_v`` / ``bias`` keys, or baked ``weight`` keys after
``remove_weight_norm``).  No module classes, no autograd state: one function
per reference ``forward``.  All citations are relative to /root/reference/.

Pinned against the reference itself by tests/golden/make_golden.py (fixtures in
tests/golden/*.npz) and, when /root/reference is present, directly in
tests/test_oracle_vs_reference.py.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# weight reparametrisations
# ----------------------------------------------------------------------------
def weight_norm_weight(g, v):
    """Old-style ``torch.nn.utils.weight_norm`` (dim=0): w = g * v / ||v||.

    The norm runs over every dim but 0, i.e. per OUT channel for Conv1d/Conv2d
    and per IN channel for ConvTranspose1d (SURVEY.md App. A "Weight norm";
    call sites models/hifigan.py:221-231, models/parallel_wavegan.py:187-195).
    """
    dims = tuple(range(1, v.dim()))
    norm = v.pow(2).sum(dim=dims, keepdim=True).sqrt()
    return v * (g / norm)


def get_weight(sd, prefix):
    """Effective conv weight for ``prefix`` from a reference state_dict."""
    if prefix + ".weight" in sd:
        return sd[prefix + ".weight"]
    if prefix + ".weight_g" in sd:
        return weight_norm_weight(sd[prefix + ".weight_g"], sd[prefix + ".weight_v"])
    raise KeyError(prefix)


def get_bias(sd, prefix):
    return sd.get(prefix + ".bias", None)


# ----------------------------------------------------------------------------
# HiFi-GAN generator  (models/hifigan.py:173-192, layers/residual_block.py:243-258)
# ----------------------------------------------------------------------------
def hifigan_resblock(sd, prefix, x, kernel_size, dilations, slope, use_additional_convs=True):
    """``HiFiGANResidualBlock.forward`` layers/residual_block.py:243-258."""
    for idx, d in enumerate(dilations):
        p1 = f"{prefix}.convs1.{idx}.1"
        xt = F.conv1d(
            F.leaky_relu(x, slope),
            get_weight(sd, p1),
            get_bias(sd, p1),
            dilation=d,
            padding=(kernel_size - 1) // 2 * d,
        )
        if use_additional_convs:
            p2 = f"{prefix}.convs2.{idx}.1"
            xt = F.conv1d(
                F.leaky_relu(xt, slope),
                get_weight(sd, p2),
                get_bias(sd, p2),
                padding=(kernel_size - 1) // 2,
            )
        x = xt + x
    return x


def hifigan_generator(
    sd,
    c,
    kernel_size=7,
    upsample_scales=(8, 8, 2, 2),
    upsample_kernel_sizes=(16, 16, 4, 4),
    resblock_kernel_sizes=(3, 7, 11),
    resblock_dilations=((1, 3, 5), (1, 3, 5), (1, 3, 5)),
    use_additional_convs=True,
    slope=0.1,
    return_stages=False,
    **_unused,
):
    """``HiFiGANGenerator.forward`` models/hifigan.py:173-192.  c: (B, 80, F)."""
    stages = []
    c = F.conv1d(
        c,
        get_weight(sd, "input_conv"),
        get_bias(sd, "input_conv"),
        padding=(kernel_size - 1) // 2,
    )
    stages.append(c)
    nb = len(resblock_kernel_sizes)
    for i, (s, k) in enumerate(zip(upsample_scales, upsample_kernel_sizes)):
        p = f"upsamples.{i}.1"
        # models/hifigan.py:99-107: padding = s//2 + s%2, output_padding = s%2
        c = F.conv_transpose1d(
            F.leaky_relu(c, slope),
            get_weight(sd, p),
            get_bias(sd, p),
            stride=s,
            padding=s // 2 + s % 2,
            output_padding=s % 2,
        )
        cs = 0.0
        for j in range(nb):
            cs = cs + hifigan_resblock(
                sd,
                f"blocks.{i * nb + j}",
                c,
                resblock_kernel_sizes[j],
                resblock_dilations[j],
                slope,
                use_additional_convs,
            )
        c = cs / nb
        stages.append(c)
    # models/hifigan.py:139-151: default LeakyReLU slope 0.01 before the last conv
    p = "output_conv.1"
    c = torch.tanh(
        F.conv1d(
            F.leaky_relu(c, 0.01),
            get_weight(sd, p),
            get_bias(sd, p),
            padding=(kernel_size - 1) // 2,
        )
    )
    if return_stages:
        return c, stages
    return c


def hifigan_inference(sd, c, normalize_before=False, **params):
    """``HiFiGANGenerator.inference`` models/hifigan.py:251-267.  c: (T', 80)."""
    c = torch.as_tensor(c, dtype=torch.float)
    if normalize_before:
        c = (c - sd["mean"]) / sd["scale"]
    y = hifigan_generator(sd, c.transpose(1, 0).unsqueeze(0), **params)
    return y.squeeze(0).transpose(1, 0)
