"""Deterministic synthetic state_dicts and inputs shared by the golden-vector
generator (run in the build container against the reference) and the parity
tests (run anywhere, including the GPU box where /root/reference is absent).

numpy's PCG64 streams are stable across platforms, so regenerating from
(name, shape, seed) gives bit-identical float32 tensors on both sides; only the
reference's OUTPUTS need to be committed as fixtures.
"""
import zlib

import numpy as np
import torch


def _rng(seed, name):
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


def synth_tensor(name, shape, seed=0, g_scale=1.0):
    """Value for one reference state_dict entry, chosen by its key suffix."""
    r = _rng(seed, name)
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "weight_g":
        a = g_scale * (1.0 + 0.1 * r.standard_normal(shape))
        if int(np.prod(shape)) == 1:  # single-row weights (PWG's 1x(2s+1) smoothing Conv2d): keep ~unit DC gain
            a = np.abs(a)
    elif leaf == "weight_v" and len(shape) == 1:  # spectral-norm right singular vector estimate
        a = r.standard_normal(shape)
        a = a / np.linalg.norm(a)
    elif leaf in ("weight_v", "weight_orig"):
        a = r.standard_normal(shape)
        if leaf == "weight_orig":
            a = g_scale * a / np.sqrt(max(int(np.prod(shape[1:])), 1))
    elif leaf == "weight":
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        a = g_scale * r.standard_normal(shape) / np.sqrt(max(fan_in, 1))
    elif leaf == "bias":
        a = 0.05 * r.standard_normal(shape)
    elif leaf == "weight_u":
        a = r.standard_normal(shape)
        a = a / np.linalg.norm(a)
    elif leaf == "mean":
        a = 0.1 * r.standard_normal(shape)
    elif leaf == "scale":
        a = 1.0 + 0.1 * r.random(shape)
    else:
        raise KeyError(f"no synthetic rule for state_dict key {name!r}")
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def synth_state_dict(shapes, seed=0, g_scale=1.0, skip=()):
    """shapes: mapping name -> shape (e.g. from ``module.state_dict()``)."""
    out = {}
    for name, shp in shapes.items():
        if any(name.endswith(s) for s in skip):
            continue
        shp = tuple(shp.shape) if hasattr(shp, "shape") else tuple(shp)
        out[name] = synth_tensor(name, shp, seed, g_scale)
    # spectral-norm triplets: make (u, v) a converged singular-vector estimate of weight_orig, as a
    # trained checkpoint would hold (a random u, v gives an arbitrary sigma = u^T W v in eval mode)
    for name in [n for n in out if n.endswith(".weight_orig")]:
        base = name[: -len("weight_orig")]
        if base + "weight_u" in out and base + "weight_v" in out:
            w = out[name].double().numpy().reshape(out[name].shape[0], -1)
            u = out[base + "weight_u"].double().numpy()
            for _ in range(30):
                v = w.T @ u
                v /= np.linalg.norm(v)
                u = w @ v
                u /= np.linalg.norm(u)
            out[base + "weight_u"] = torch.from_numpy(u.astype(np.float32))
            out[base + "weight_v"] = torch.from_numpy(v.astype(np.float32))
    return out


def synth_input(name, shape, seed=0):
    r = _rng(seed, "input:" + name)
    return torch.from_numpy(r.standard_normal(tuple(shape)).astype(np.float32))


PWG_G_SCALE = 1.0
MELGAN_G_SCALE = 0.95

# ---------------------------------------------------------------------------
# Configurations exercised by the golden fixtures (subset of the reference's
# YAML generator_params; full-size ones equal egs/ljspeech/voc1/conf/*.yaml).
# ---------------------------------------------------------------------------
HIFIGAN_V1 = dict(
    in_channels=80,
    out_channels=1,
    channels=512,
    kernel_size=7,
    upsample_scales=[8, 8, 2, 2],
    upsample_kernel_sizes=[16, 16, 4, 4],
    resblock_kernel_sizes=[3, 7, 11],
    resblock_dilations=[[1, 3, 5], [1, 3, 5], [1, 3, 5]],
    use_additional_convs=True,
    bias=True,
    nonlinear_activation="LeakyReLU",
    nonlinear_activation_params={"negative_slope": 0.1},
    use_weight_norm=True,
)
# LibriTTS 24 kHz variant (egs/libritts/voc1/conf/hifigan.v1.yaml): odd scales
HIFIGAN_V1_LIBRITTS = dict(HIFIGAN_V1, upsample_scales=[5, 5, 4, 3], upsample_kernel_sizes=[10, 10, 8, 6])
# small config in the spirit of the reference's own unit tests (test/test_hifigan.py)
HIFIGAN_TINY = dict(
    HIFIGAN_V1,
    channels=64,
    upsample_scales=[4, 3, 2],
    upsample_kernel_sizes=[8, 6, 4],
    resblock_kernel_sizes=[3, 5],
    resblock_dilations=[[1, 3], [1, 2]],
)


# ---- causal variants / residual PWG discriminator: small configurations
HIFIGAN_CAUSAL = dict(in_channels=80, out_channels=1, channels=64, kernel_size=7, upsample_scales=(4, 2, 2),
                      upsample_kernel_sizes=(8, 4, 4), resblock_kernel_sizes=(3, 7),
                      resblock_dilations=[(1, 3, 5), (1, 3)], use_additional_convs=True, use_causal_conv=True)
MELGAN_CAUSAL = dict(in_channels=80, out_channels=1, kernel_size=7, channels=64, upsample_scales=[4, 2, 2],
                     stack_kernel_size=3, stacks=2, use_causal_conv=True)
PWG_CAUSAL = dict(layers=6, stacks=2, aux_context_window=2, use_causal_conv=True,
                  upsample_params={"upsample_scales": [4, 4]})
RESIDUAL_PWG_D = dict(layers=6, stacks=2, residual_channels=32, gate_channels=64, skip_channels=32)

# ---- StyleMelGAN
STYLE_MELGAN_TINY = dict(in_channels=16, aux_channels=80, channels=32, out_channels=1, kernel_size=9, dilation=2,
                         noise_upsample_scales=[2, 2], upsample_scales=[2, 2, 2, 1], gated_function="softmax")
STYLE_MELGAN_TINY_SIGMOID = dict(STYLE_MELGAN_TINY, gated_function="sigmoid", kernel_size=5)
STYLE_MELGAN_D = dict(repeats=2)
PQMF_BUFFERS = ("analysis_filter", "synthesis_filter", "updown_filter")  # fixed filters, not synthesised

# ---- UHiFiGAN: a small configuration (factor 4 * 2 = 8 samples per frame)
# (the decoder mirrors the encoder: upsample scales are the downsample scales reversed)
UHIFIGAN_TINY = dict(in_channels=80, out_channels=1, channels=16, kernel_size=7, downsample_scales=(4, 2),
                     downsample_kernel_sizes=(8, 4), upsample_scales=(2, 4), upsample_kernel_sizes=(4, 8),
                     resblock_kernel_sizes=(3, 7), resblock_dilations=[(1, 3, 5), (1, 3)], dropout=0.3)


PWG_MELGAN_UPSAMPLER = dict(
    aux_context_window=0, layers=6, stacks=2,
    upsample_net="MelGANGenerator",
    upsample_params=dict(in_channels=80, out_channels=80, kernel_size=7, channels=256, upsample_scales=[4, 4, 4, 4],
                         stack_kernel_size=3, stacks=2),
)


def adv_logits(seed):
    """Synthetic discriminator outputs (3 discriminators x [2 feature maps + logits]); the logits hold
    exact ties of the hinge terms (x == 1, x == -1) to pin torch.min's 1/2 tie gradient."""
    outs = []
    for i, t in enumerate((50, 37, 9)):
        fm = [synth_input(f"adv_f{i}{j}", (2, 4 * (j + 1), t), seed=seed) for j in range(2)]
        logit = 1.5 * synth_input(f"adv_l{i}", (2, 1, t), seed=seed)
        logit[0, 0, 0], logit[1, 0, 1] = 1.0, -1.0
        outs.append(fm + [logit])
    return outs


# ---- training-step fixtures of the remaining families (round 4): shared by make_golden.py and the GPU test
STYLE_MELGAN_TRAIN = dict(STYLE_MELGAN_TINY, noise_upsample_scales=[4, 4], upsample_scales=[4, 4, 4, 4])  # 16 frames per z
STYLE_MELGAN_TRAIN_D = dict(repeats=1, window_sizes=[128, 256, 512, 1024])
UHIFIGAN_TRAIN = dict(UHIFIGAN_TINY, dropout=0.0)  # (the dropout masks are pinned separately, tests/test_pwg_dropout_gpu.py)
FAMILY_TRAIN_STFT = dict(fft_sizes=[256, 512], hop_sizes=[32, 64], win_lengths=[128, 256])
FAMILY_TRAIN_CFG = dict(use_stft_loss=True, use_subband_stft_loss=False, use_mel_loss=False, use_feat_match_loss=False,
                        lambda_aux=1.0, lambda_adv=1.0, generator_grad_norm=-1, discriminator_grad_norm=-1,
                        generator_train_start_steps=0, discriminator_train_start_steps=0)
FAMILY_TRAIN_LR = dict(generator=5e-4, discriminator=1e-4)
