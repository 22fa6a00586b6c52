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


# ----------------------------------------------------------------------------
# spectral norm (torch.nn.utils.spectral_norm, old hook API, dim=0, 1 iteration)
# ----------------------------------------------------------------------------
def spectral_norm_weight(sd, prefix, training, eps=1e-12):
    """``weight = weight_orig / sigma``; in training mode one power iteration first, which
    UPDATES ``sd[prefix.weight_u]`` / ``weight_v`` in place like the hook does
    (call sites models/hifigan.py:613-621; torch/nn/utils/spectral_norm.py is third-party)."""
    w = sd[prefix + ".weight_orig"]
    u, v = sd[prefix + ".weight_u"], sd[prefix + ".weight_v"]
    wm = w.reshape(w.shape[0], -1)
    if training:
        with torch.no_grad():
            v = F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps)
            u = F.normalize(torch.mv(wm, v), dim=0, eps=eps)
            sd[prefix + ".weight_u"], sd[prefix + ".weight_v"] = u, v
    sigma = torch.dot(u, torch.mv(wm, v))
    return w / sigma


def get_weight_any(sd, prefix, training=False):
    if prefix + ".weight_orig" in sd:
        return spectral_norm_weight(sd, prefix, training)
    return get_weight(sd, prefix)


# ----------------------------------------------------------------------------
# HiFi-GAN discriminators (models/hifigan.py:354-381, 586-601, 762-777, 850-864)
# ----------------------------------------------------------------------------
def hifigan_period_discriminator(sd, prefix, x, period, kernel_sizes=(5, 3), downsample_scales=(3, 3, 3, 3, 1),
                                 slope=0.1, training=False):
    """``HiFiGANPeriodDiscriminator.forward`` models/hifigan.py:354-381."""
    b, c, t = x.shape
    if t % period != 0:
        n_pad = period - (t % period)
        x = F.pad(x, (0, n_pad), "reflect")
        t += n_pad
    x = x.view(b, c, t // period, period)
    outs = []
    for i, s in enumerate(downsample_scales):
        p = f"{prefix}.convs.{i}.0"
        x = F.leaky_relu(F.conv2d(x, get_weight_any(sd, p, training), get_bias(sd, p), stride=(s, 1),
                                  padding=((kernel_sizes[0] - 1) // 2, 0)), slope)
        outs.append(x)
    p = f"{prefix}.output_conv"
    x = F.conv2d(x, get_weight_any(sd, p, training), get_bias(sd, p), stride=1,
                 padding=((kernel_sizes[1] - 1) // 2, 0))
    outs.append(torch.flatten(x, 1, -1))
    return outs


def hifigan_scale_discriminator(sd, prefix, x, kernel_sizes=(15, 41, 5, 3), channels=128,
                                max_downsample_channels=1024, max_groups=16, downsample_scales=(2, 2, 4, 4, 1),
                                slope=0.1, training=False):
    """``HiFiGANScaleDiscriminator.forward`` models/hifigan.py:586-601 (layer list :501-568)."""
    outs = []
    p = f"{prefix}.layers.0.0"
    x = F.leaky_relu(F.conv1d(x, get_weight_any(sd, p, training), get_bias(sd, p),
                              padding=(kernel_sizes[0] - 1) // 2), slope)
    outs.append(x)
    groups = 4
    n = 1
    for s in downsample_scales:
        p = f"{prefix}.layers.{n}.0"
        x = F.leaky_relu(F.conv1d(x, get_weight_any(sd, p, training), get_bias(sd, p), stride=s,
                                  padding=(kernel_sizes[1] - 1) // 2, groups=groups), slope)
        outs.append(x)
        groups = min(groups * 4, max_groups)
        n += 1
    p = f"{prefix}.layers.{n}.0"
    x = F.leaky_relu(F.conv1d(x, get_weight_any(sd, p, training), get_bias(sd, p),
                              padding=(kernel_sizes[2] - 1) // 2), slope)
    outs.append(x)
    p = f"{prefix}.layers.{n + 1}"
    x = F.conv1d(x, get_weight_any(sd, p, training), get_bias(sd, p), padding=(kernel_sizes[3] - 1) // 2)
    outs.append(x)
    return outs


def hifigan_msmpd(sd, x, scales=3, scale_downsample_pooling_params=None, scale_discriminator_params=None,
                  periods=(2, 3, 5, 7, 11), period_discriminator_params=None, training=False, **_unused):
    """``HiFiGANMultiScaleMultiPeriodDiscriminator.forward`` models/hifigan.py:850-864."""
    pp = scale_downsample_pooling_params or {"kernel_size": 4, "stride": 2, "padding": 2}
    sp = dict(scale_discriminator_params or {})
    dp = dict(period_discriminator_params or {})
    outs = []
    xs = x
    for i in range(scales):
        outs.append(hifigan_scale_discriminator(
            sd, f"msd.discriminators.{i}", xs, kernel_sizes=sp.get("kernel_sizes", (15, 41, 5, 3)),
            channels=sp.get("channels", 128), max_downsample_channels=sp.get("max_downsample_channels", 1024),
            max_groups=sp.get("max_groups", 16), downsample_scales=sp.get("downsample_scales", (2, 2, 4, 4, 1)),
            slope=sp.get("nonlinear_activation_params", {}).get("negative_slope", 0.1), training=training))
        xs = F.avg_pool1d(xs, pp["kernel_size"], pp["stride"], pp["padding"])
    for i, period in enumerate(periods):
        outs.append(hifigan_period_discriminator(
            sd, f"mpd.discriminators.{i}", x, period, kernel_sizes=dp.get("kernel_sizes", (5, 3)),
            downsample_scales=dp.get("downsample_scales", (3, 3, 3, 3, 1)),
            slope=dp.get("nonlinear_activation_params", {}).get("negative_slope", 0.1), training=training))
    return outs


# ----------------------------------------------------------------------------
# losses (losses/stft_loss.py, losses/mel_loss.py, adversarial_loss.py, feat_match_loss.py)
# ----------------------------------------------------------------------------
def stft_magnitude(x, fft_size, hop_size, win_length, eps=1e-7):
    """``stft`` losses/stft_loss.py:16-40 -> (B, frames, bins)."""
    window = torch.hann_window(win_length, dtype=x.dtype)
    s = torch.stft(x, fft_size, hop_size, win_length, window, return_complex=True)
    power = s.real ** 2 + s.imag ** 2
    return torch.sqrt(torch.clamp(power, min=eps)).transpose(2, 1)


def multi_resolution_stft_loss(x, y, fft_sizes=(1024, 2048, 512), hop_sizes=(120, 240, 50),
                               win_lengths=(600, 1200, 240)):
    """``MultiResolutionSTFTLoss.forward`` losses/stft_loss.py:146-170 (x predicted, y target)."""
    if x.dim() == 3:
        x = x.reshape(-1, x.size(2))
        y = y.reshape(-1, y.size(2))
    sc, mag = 0.0, 0.0
    for n_fft, hop, win in zip(fft_sizes, hop_sizes, win_lengths):
        xm = stft_magnitude(x, n_fft, hop, win)
        ym = stft_magnitude(y, n_fft, hop, win)
        sc = sc + torch.norm(ym - xm, p="fro") / torch.norm(ym, p="fro")
        mag = mag + F.l1_loss(torch.log(ym), torch.log(xm))
    return sc / len(fft_sizes), mag / len(fft_sizes)


def mel_spectrogram(x, fs=22050, fft_size=1024, hop_size=256, win_length=None, num_mels=80, fmin=80, fmax=7600,
                    eps=1e-10, log_base=10.0):
    """``MelSpectrogram.forward`` losses/mel_loss.py:81-110 -> (B, mels, frames)."""
    from . import slaney_mel

    if x.dim() == 3:
        x = x.reshape(-1, x.size(2))
    win_length = fft_size if win_length is None else win_length
    fmin = 0 if fmin is None else fmin
    fmax = fs / 2 if fmax is None else fmax
    melmat = torch.from_numpy(slaney_mel.mel(fs, fft_size, num_mels, fmin, fmax).T).float()
    amp = stft_magnitude(x, fft_size, hop_size, win_length, eps=eps)  # (B, frames, bins)
    mel = torch.clamp(torch.matmul(amp, melmat), min=eps)
    log = {None: torch.log, 2.0: torch.log2, 10.0: torch.log10}[log_base]
    return log(mel).transpose(1, 2)


def mel_spectrogram_loss(y_hat, y, **params):
    """``MelSpectrogramLoss.forward`` losses/mel_loss.py:150-165."""
    return F.l1_loss(mel_spectrogram(y_hat, **params), mel_spectrogram(y, **params))


def generator_adversarial_loss(outputs, average_by_discriminators=True, loss_type="mse"):
    """``GeneratorAdversarialLoss.forward`` losses/adversarial_loss.py:29-58 (mse :54-55, hinge :57-58)."""
    loss = 0.0
    for i, o in enumerate(outputs):
        o = o[-1] if isinstance(o, (list, tuple)) else o
        loss = loss + (F.mse_loss(o, torch.ones_like(o)) if loss_type == "mse" else -o.mean())
    return loss / (i + 1) if average_by_discriminators else loss


def discriminator_adversarial_loss(outputs_hat, outputs, average_by_discriminators=True, loss_type="mse"):
    """``DiscriminatorAdversarialLoss.forward`` losses/adversarial_loss.py:80-123 (mse :113-117,
    hinge :119-123: -mean(min(x - 1, 0)) on real, -mean(min(-x - 1, 0)) on generated)."""
    real, fake = 0.0, 0.0
    for i, (oh, o) in enumerate(zip(outputs_hat, outputs)):
        if isinstance(oh, (list, tuple)):
            oh, o = oh[-1], o[-1]
        if loss_type == "mse":
            real = real + F.mse_loss(o, torch.ones_like(o))
            fake = fake + F.mse_loss(oh, torch.zeros_like(oh))
        else:
            real = real - torch.mean(torch.min(o - 1, torch.zeros_like(o)))
            fake = fake - torch.mean(torch.min(-oh - 1, torch.zeros_like(oh)))
    if average_by_discriminators:
        real, fake = real / (i + 1), fake / (i + 1)
    return real, fake


def feature_match_loss(feats_hat, feats, average_by_layers=True, average_by_discriminators=True,
                       include_final_outputs=False):
    """``FeatureMatchLoss.forward`` losses/feat_match_loss.py:27-54."""
    total = 0.0
    for i, (fh, f) in enumerate(zip(feats_hat, feats)):
        if not include_final_outputs:
            fh, f = fh[:-1], f[:-1]
        part = 0.0
        for j, (a, b) in enumerate(zip(fh, f)):
            part = part + F.l1_loss(a, b.detach())
        if average_by_layers:
            part = part / (j + 1)
        total = total + part
    return total / (i + 1) if average_by_discriminators else total


# ----------------------------------------------------------------------------
# Parallel WaveGAN (models/parallel_wavegan.py:144-173,337-349; layers/upsample.py; residual_block.py:102-140)
# ----------------------------------------------------------------------------
def pwg_upsample(sd, c, upsample_scales=(4, 4, 4, 4), prefix="upsample_net"):
    """``ConvInUpsampleNetwork.forward`` layers/upsample.py:178-194 (+ ``UpsampleNetwork`` :112-128)."""
    c = F.conv1d(c, get_weight(sd, f"{prefix}.conv_in"))  # no padding: the input carries the context frames
    c = c.unsqueeze(1)
    for i, s in enumerate(upsample_scales):
        c = F.interpolate(c, scale_factor=(1, s), mode="nearest")
        c = F.conv2d(c, get_weight(sd, f"{prefix}.upsample.up_layers.{2 * i + 1}"), padding=(0, s))
    return c.squeeze(1)


def pwg_generator(sd, z, c, layers=30, stacks=3, kernel_size=3, upsample_params=None, dropout_masks=None, **_unused):
    """``ParallelWaveGANGenerator.forward`` models/parallel_wavegan.py:144-173.  ``dropout_masks``: optional
    per-layer multipliers (kept / (1 - p) or 0) standing in for ``F.dropout`` on the dilated conv's input
    (layers/residual_block.py:114-116; the residual path keeps the undropped x)."""
    scales = (upsample_params or {}).get("upsample_scales", (4, 4, 4, 4))
    c = pwg_upsample(sd, c, scales)
    assert c.size(-1) == z.size(-1)
    x = F.conv1d(z, get_weight(sd, "first_conv"), get_bias(sd, "first_conv"))
    skips = 0
    per_stack = layers // stacks
    for l in range(layers):
        p = f"conv_layers.{l}"
        d = 2 ** (l % per_stack)
        residual = x
        xin = x if dropout_masks is None else x * dropout_masks[l]
        h = F.conv1d(xin, get_weight(sd, p + ".conv"), get_bias(sd, p + ".conv"), dilation=d,
                     padding=(kernel_size - 1) // 2 * d)
        xa, xb = h.split(h.size(1) // 2, dim=1)
        a = F.conv1d(c, get_weight(sd, p + ".conv1x1_aux"))
        ca, cb = a.split(a.size(1) // 2, dim=1)
        g = torch.tanh(xa + ca) * torch.sigmoid(xb + cb)
        s = F.conv1d(g, get_weight(sd, p + ".conv1x1_skip"), get_bias(sd, p + ".conv1x1_skip"))
        x = (F.conv1d(g, get_weight(sd, p + ".conv1x1_out"), get_bias(sd, p + ".conv1x1_out")) + residual) * math.sqrt(0.5)
        skips = skips + s
    skips = skips * math.sqrt(1.0 / layers)
    x = F.conv1d(F.relu(skips), get_weight(sd, "last_conv_layers.1"), get_bias(sd, "last_conv_layers.1"))
    return F.conv1d(F.relu(x), get_weight(sd, "last_conv_layers.3"), get_bias(sd, "last_conv_layers.3"))


def pwg_discriminator(sd, x, layers=10, kernel_size=3, dilation_factor=1, slope=0.2, **_unused):
    """``ParallelWaveGANDiscriminator.forward`` models/parallel_wavegan.py:337-349."""
    for i in range(layers - 1):
        d = 1 if i == 0 else (i if dilation_factor == 1 else dilation_factor ** i)
        p = f"conv_layers.{2 * i}"
        x = F.leaky_relu(F.conv1d(x, get_weight(sd, p), get_bias(sd, p), dilation=d,
                                  padding=(kernel_size - 1) // 2 * d), slope)
    p = f"conv_layers.{2 * (layers - 1)}"
    return F.conv1d(x, get_weight(sd, p), get_bias(sd, p), padding=(kernel_size - 1) // 2)


# ----------------------------------------------------------------------------
# MelGAN / Multi-band MelGAN (models/melgan.py:168-178,364-379,478-493; layers/residual_stack.py:75-85)
# ----------------------------------------------------------------------------
def melgan_generator(sd, c, kernel_size=7, upsample_scales=(8, 8, 2, 2), stack_kernel_size=3, stacks=3, slope=0.2,
                     use_final_nonlinear_activation=True, **_unused):
    """``MelGANGenerator.forward`` models/melgan.py:168-178 (flat Sequential built at :67-156)."""
    p = (kernel_size - 1) // 2
    x = F.conv1d(F.pad(c, (p, p), mode="reflect"), get_weight(sd, "melgan.1"), get_bias(sd, "melgan.1"))
    idx = 2
    for s in upsample_scales:
        x = F.conv_transpose1d(F.leaky_relu(x, slope), get_weight(sd, f"melgan.{idx + 1}"),
                               get_bias(sd, f"melgan.{idx + 1}"), stride=s, padding=s // 2 + s % 2,
                               output_padding=s % 2)
        idx += 2
        for j in range(stacks):
            d = stack_kernel_size ** j
            pre = f"melgan.{idx}"
            pd = (stack_kernel_size - 1) // 2 * d
            t = F.conv1d(F.pad(F.leaky_relu(x, slope), (pd, pd), mode="reflect"), get_weight(sd, pre + ".stack.2"),
                         get_bias(sd, pre + ".stack.2"), dilation=d)
            t = F.conv1d(F.leaky_relu(t, slope), get_weight(sd, pre + ".stack.4"), get_bias(sd, pre + ".stack.4"))
            x = t + F.conv1d(x, get_weight(sd, pre + ".skip_layer"), get_bias(sd, pre + ".skip_layer"))
            idx += 1
    x = F.conv1d(F.pad(F.leaky_relu(x, slope), (p, p), mode="reflect"), get_weight(sd, f"melgan.{idx + 2}"),
                 get_bias(sd, f"melgan.{idx + 2}"))
    return torch.tanh(x) if use_final_nonlinear_activation else x


def melgan_discriminator(sd, prefix, x, kernel_sizes=(5, 3), downsample_scales=(4, 4, 4, 4), slope=0.2):
    """``MelGANDiscriminator.forward`` models/melgan.py:364-379 (layers :304-359)."""
    outs = []
    k0 = int(np.prod(kernel_sizes))
    p = f"{prefix}.layers.0.1"
    x = F.leaky_relu(F.conv1d(F.pad(x, ((k0 - 1) // 2, (k0 - 1) // 2), mode="reflect"), get_weight(sd, p),
                              get_bias(sd, p)), slope)
    outs.append(x)
    n = 1
    for s in downsample_scales:
        p = f"{prefix}.layers.{n}.0"
        x = F.leaky_relu(F.conv1d(x, get_weight(sd, p), get_bias(sd, p), stride=s, padding=s * 5,
                                  groups=x.size(1) // 4), slope)
        outs.append(x)
        n += 1
    p = f"{prefix}.layers.{n}.0"
    x = F.leaky_relu(F.conv1d(x, get_weight(sd, p), get_bias(sd, p), padding=(kernel_sizes[0] - 1) // 2), slope)
    outs.append(x)
    p = f"{prefix}.layers.{n + 1}"
    x = F.conv1d(x, get_weight(sd, p), get_bias(sd, p), padding=(kernel_sizes[1] - 1) // 2)
    outs.append(x)
    return outs


def melgan_multi_scale_discriminator(sd, x, scales=3, downsample_pooling_params=None, kernel_sizes=(5, 3),
                                     downsample_scales=(4, 4, 4, 4), nonlinear_activation_params=None, **_unused):
    """``MelGANMultiScaleDiscriminator.forward`` models/melgan.py:478-493."""
    pp = downsample_pooling_params or {"kernel_size": 4, "stride": 2, "padding": 1, "count_include_pad": False}
    slope = (nonlinear_activation_params or {}).get("negative_slope", 0.2)
    outs = []
    for i in range(scales):
        outs.append(melgan_discriminator(sd, f"discriminators.{i}", x, kernel_sizes, downsample_scales, slope))
        x = F.avg_pool1d(x, pp["kernel_size"], pp["stride"], pp["padding"],
                         count_include_pad=pp.get("count_include_pad", True))
    return outs


# ----------------------------------------------------------------------------
# PQMF (layers/pqmf.py:14-48,61-149)
# ----------------------------------------------------------------------------
def pqmf_filters(subbands=4, taps=62, cutoff_ratio=0.142, beta=9.0):
    """Analysis (K,1,taps+1) / synthesis (1,K,taps+1) filters, layers/pqmf.py:37-46,80-107."""
    import scipy.signal.windows

    n = np.arange(taps + 1) - 0.5 * taps
    with np.errstate(invalid="ignore", divide="ignore"):
        h_i = np.sin(np.pi * cutoff_ratio * n) / (np.pi * n)
    h_i[taps // 2] = cutoff_ratio
    h = h_i * scipy.signal.windows.kaiser(taps + 1, beta)
    ha = np.zeros((subbands, taps + 1))
    hs = np.zeros((subbands, taps + 1))
    for k in range(subbands):
        arg = (2 * k + 1) * (np.pi / (2 * subbands)) * n
        ha[k] = 2 * h * np.cos(arg + (-1) ** k * np.pi / 4)
        hs[k] = 2 * h * np.cos(arg - (-1) ** k * np.pi / 4)
    return torch.from_numpy(ha).float().unsqueeze(1), torch.from_numpy(hs).float().unsqueeze(0)


def _updown(subbands):
    f = torch.zeros(subbands, subbands, subbands)
    for k in range(subbands):
        f[k, k, 0] = 1.0
    return f


def pqmf_analysis(x, subbands=4, taps=62, cutoff_ratio=0.142, beta=9.0):
    """``PQMF.analysis`` layers/pqmf.py:120-131."""
    ha, _ = pqmf_filters(subbands, taps, cutoff_ratio, beta)
    x = F.conv1d(F.pad(x, (taps // 2, taps // 2)), ha)
    return F.conv1d(x, _updown(subbands), stride=subbands)


def pqmf_synthesis(x, subbands=4, taps=62, cutoff_ratio=0.142, beta=9.0):
    """``PQMF.synthesis`` layers/pqmf.py:133-149."""
    _, hs = pqmf_filters(subbands, taps, cutoff_ratio, beta)
    x = F.conv_transpose1d(x, _updown(subbands) * subbands, stride=subbands)
    return F.conv1d(F.pad(x, (taps // 2, taps // 2)), hs)


# ----------------------------------------------------------------------------
# causal variants (layers/causal_conv.py:12-77 and the use_causal_conv branches of the models)
# ----------------------------------------------------------------------------
def causal_conv1d(x, w, b, dilation=1, pad_mode="constant"):
    """``CausalConv1d.forward`` layers/causal_conv.py:33-43: left pad (k-1)*d, conv, keep the first T."""
    p = (w.shape[-1] - 1) * dilation
    xp = F.pad(x, (p, p), mode=pad_mode) if pad_mode != "constant" else F.pad(x, (p, p))
    return F.conv1d(xp, w, b, dilation=dilation)[:, :, : x.size(2)]


def causal_conv_transpose1d(x, w, b, stride):
    """``CausalConvTranspose1d.forward`` layers/causal_conv.py:67-77: replicate-pad one sample on the
    left, transposed conv, drop ``stride`` samples at both ends."""
    return F.conv_transpose1d(F.pad(x, (1, 0), mode="replicate"), w, b, stride=stride)[:, :, stride:-stride]


def hifigan_generator_causal(sd, c, upsample_scales=(8, 8, 2, 2), resblock_kernel_sizes=(3, 7, 11),
                             resblock_dilations=((1, 3, 5), (1, 3, 5), (1, 3, 5)), use_additional_convs=True,
                             slope=0.1, **_unused):
    """``HiFiGANGenerator.forward`` with ``use_causal_conv=True`` (models/hifigan.py:82-88,115-123,156-162;
    layers/residual_block.py:196-241).  NOTE: ``CausalConv1d`` pads on BOTH sides and trims the tail
    (causal_conv.py:27,43), which equals left-only padding."""
    c = causal_conv1d(c, get_weight(sd, "input_conv.conv"), get_bias(sd, "input_conv.conv"))
    nb = len(resblock_kernel_sizes)
    for i, s in enumerate(upsample_scales):
        p = f"upsamples.{i}.1.deconv"
        c = causal_conv_transpose1d(F.leaky_relu(c, slope), get_weight(sd, p), get_bias(sd, p), s)
        cs = 0.0
        for j in range(nb):
            x = c
            for idx, d in enumerate(resblock_dilations[j]):
                p1 = f"blocks.{i * nb + j}.convs1.{idx}.1.conv"
                xt = causal_conv1d(F.leaky_relu(x, slope), get_weight(sd, p1), get_bias(sd, p1), d)
                if use_additional_convs:
                    p2 = f"blocks.{i * nb + j}.convs2.{idx}.1.conv"
                    xt = causal_conv1d(F.leaky_relu(xt, slope), get_weight(sd, p2), get_bias(sd, p2), 1)
                x = xt + x
            cs = cs + x
        c = cs / nb
    p = "output_conv.1.conv"
    return torch.tanh(causal_conv1d(F.leaky_relu(c, 0.01), get_weight(sd, p), get_bias(sd, p)))


def melgan_generator_causal(sd, c, upsample_scales=(8, 8, 2, 2), stack_kernel_size=3, stacks=3, slope=0.2,
                            use_final_nonlinear_activation=True, **_unused):
    """``MelGANGenerator.forward`` with ``use_causal_conv=True`` (models/melgan.py:75-84,104-112,142-151;
    layers/residual_stack.py:56-69): reflection padding, left side only after the trim."""
    x = causal_conv1d(c, get_weight(sd, "melgan.0.conv"), get_bias(sd, "melgan.0.conv"), pad_mode="reflect")
    idx = 1
    for s in upsample_scales:
        p = f"melgan.{idx + 1}.deconv"
        x = causal_conv_transpose1d(F.leaky_relu(x, slope), get_weight(sd, p), get_bias(sd, p), s)
        idx += 2
        for j in range(stacks):
            d = stack_kernel_size ** j
            pre = f"melgan.{idx}"
            t = causal_conv1d(F.leaky_relu(x, slope), get_weight(sd, pre + ".stack.1.conv"),
                              get_bias(sd, pre + ".stack.1.conv"), d, pad_mode="reflect")
            t = F.conv1d(F.leaky_relu(t, slope), get_weight(sd, pre + ".stack.3"), get_bias(sd, pre + ".stack.3"))
            x = t + F.conv1d(x, get_weight(sd, pre + ".skip_layer"), get_bias(sd, pre + ".skip_layer"))
            idx += 1
    p = f"melgan.{idx + 1}.conv"
    x = causal_conv1d(F.leaky_relu(x, slope), get_weight(sd, p), get_bias(sd, p), pad_mode="reflect")
    return torch.tanh(x) if use_final_nonlinear_activation else x


def pwg_generator_causal(sd, z, c, layers=30, stacks=3, kernel_size=3, aux_context_window=2, upsample_params=None,
                         **_unused):
    """``ParallelWaveGANGenerator.forward`` with ``use_causal_conv=True`` (layers/upsample.py:96-99,
    121-125,160-164,192-193; layers/residual_block.py:74-76,118-119)."""
    scales = (upsample_params or {}).get("upsample_scales", (4, 4, 4, 4))
    c = F.conv1d(c, get_weight(sd, "upsample_net.conv_in"))  # kernel aux_context_window + 1
    if aux_context_window > 0:
        c = c[:, :, :-aux_context_window]
    c = c.unsqueeze(1)
    for i, s in enumerate(scales):
        c = F.interpolate(c, scale_factor=(1, s), mode="nearest")
        c = F.conv2d(c, get_weight(sd, f"upsample_net.upsample.up_layers.{2 * i + 1}"), padding=(0, 2 * s))[..., : c.size(-1)]
    c = c.squeeze(1)
    assert c.size(-1) == z.size(-1)
    x = F.conv1d(z, get_weight(sd, "first_conv"), get_bias(sd, "first_conv"))
    skips = 0
    per_stack = layers // stacks
    for l in range(layers):
        p = f"conv_layers.{l}"
        d = 2 ** (l % per_stack)
        residual = x
        h = F.conv1d(x, get_weight(sd, p + ".conv"), get_bias(sd, p + ".conv"), dilation=d,
                     padding=(kernel_size - 1) * d)[:, :, : residual.size(-1)]
        xa, xb = h.split(h.size(1) // 2, dim=1)
        a = F.conv1d(c, get_weight(sd, p + ".conv1x1_aux"))
        ca, cb = a.split(a.size(1) // 2, dim=1)
        g = torch.tanh(xa + ca) * torch.sigmoid(xb + cb)
        s = F.conv1d(g, get_weight(sd, p + ".conv1x1_skip"), get_bias(sd, p + ".conv1x1_skip"))
        x = (F.conv1d(g, get_weight(sd, p + ".conv1x1_out"), get_bias(sd, p + ".conv1x1_out")) + residual) * math.sqrt(0.5)
        skips = skips + s
    skips = skips * math.sqrt(1.0 / layers)
    x = F.conv1d(F.relu(skips), get_weight(sd, "last_conv_layers.1"), get_bias(sd, "last_conv_layers.1"))
    return F.conv1d(F.relu(x), get_weight(sd, "last_conv_layers.3"), get_bias(sd, "last_conv_layers.3"))


def residual_pwg_discriminator(sd, x, layers=30, stacks=3, kernel_size=3, slope=0.2, use_causal_conv=False, **_unused):
    """``ResidualParallelWaveGANDiscriminator.forward`` models/parallel_wavegan.py:471-494."""
    x = F.leaky_relu(F.conv1d(x, get_weight(sd, "first_conv.0"), get_bias(sd, "first_conv.0")), slope)
    skips = 0
    per_stack = layers // stacks
    for l in range(layers):
        p = f"conv_layers.{l}"
        d = 2 ** (l % per_stack)
        residual = x
        pad = (kernel_size - 1) * d if use_causal_conv else (kernel_size - 1) // 2 * d
        h = F.conv1d(x, get_weight(sd, p + ".conv"), get_bias(sd, p + ".conv"), dilation=d, padding=pad)
        if use_causal_conv:
            h = h[:, :, : residual.size(-1)]
        xa, xb = h.split(h.size(1) // 2, dim=1)
        g = torch.tanh(xa) * torch.sigmoid(xb)
        s = F.conv1d(g, get_weight(sd, p + ".conv1x1_skip"), get_bias(sd, p + ".conv1x1_skip"))
        x = (F.conv1d(g, get_weight(sd, p + ".conv1x1_out"), get_bias(sd, p + ".conv1x1_out")) + residual) * math.sqrt(0.5)
        skips = skips + s
    skips = skips * math.sqrt(1.0 / layers)
    x = F.conv1d(F.leaky_relu(skips, slope), get_weight(sd, "last_conv_layers.1"), get_bias(sd, "last_conv_layers.1"))
    return F.conv1d(F.leaky_relu(x, slope), get_weight(sd, "last_conv_layers.3"), get_bias(sd, "last_conv_layers.3"))


# ----------------------------------------------------------------------------
# StyleMelGAN (layers/tade_res_block.py:11-161, models/style_melgan.py:18-362)
# ----------------------------------------------------------------------------
def tade_layer(sd, prefix, x, c, upsample_factor):
    """``TADELayer.forward`` tade_res_block.py:53-73."""
    x = F.instance_norm(x)  # InstanceNorm1d(affine=False), eps 1e-5
    c = F.interpolate(c, scale_factor=upsample_factor, mode="nearest") if upsample_factor != 1 else c
    k = get_weight(sd, prefix + ".aux_conv.0").shape[-1]
    c = F.conv1d(c, get_weight(sd, prefix + ".aux_conv.0"), get_bias(sd, prefix + ".aux_conv.0"), padding=(k - 1) // 2)
    cg = F.conv1d(c, get_weight(sd, prefix + ".gated_conv.0"), get_bias(sd, prefix + ".gated_conv.0"),
                  padding=(k - 1) // 2)
    cg1, cg2 = cg.split(cg.size(1) // 2, dim=1)
    xu = F.interpolate(x, scale_factor=upsample_factor, mode="nearest") if upsample_factor != 1 else x
    return cg1 * xu + cg2, c


def tade_res_block(sd, prefix, x, c, upsample_factor, dilation=2, gated_function="softmax"):
    """``TADEResBlock.forward`` tade_res_block.py:136-161."""
    def gate(v):
        va, vb = v.split(v.size(1) // 2, dim=1)
        g = torch.softmax(va, dim=1) if gated_function == "softmax" else torch.sigmoid(va)
        return g * torch.tanh(vb)

    residual = x
    x, c = tade_layer(sd, prefix + ".tade1", x, c, 1)
    k = get_weight(sd, prefix + ".gated_conv1").shape[-1]
    x = gate(F.conv1d(x, get_weight(sd, prefix + ".gated_conv1"), get_bias(sd, prefix + ".gated_conv1"),
                      padding=(k - 1) // 2))
    x, c = tade_layer(sd, prefix + ".tade2", x, c, upsample_factor)
    x = gate(F.conv1d(x, get_weight(sd, prefix + ".gated_conv2"), get_bias(sd, prefix + ".gated_conv2"),
                      dilation=dilation, padding=(k - 1) // 2 * dilation))
    ru = F.interpolate(residual, scale_factor=upsample_factor, mode="nearest") if upsample_factor != 1 else residual
    return ru + x, c


def style_melgan_generator(sd, c, z, noise_upsample_scales=(11, 2, 2, 2), upsample_scales=(2, 2, 2, 2, 2, 2, 2, 2, 1),
                           dilation=2, gated_function="softmax", noise_slope=0.2, **_unused):
    """``StyleMelGANGenerator.forward`` models/style_melgan.py:123-143."""
    x = z
    for i, s in enumerate(noise_upsample_scales):
        p = f"noise_upsample.{2 * i}"
        x = F.leaky_relu(F.conv_transpose1d(x, get_weight(sd, p), get_bias(sd, p), stride=s, padding=s // 2 + s % 2,
                                            output_padding=s % 2), noise_slope)
    for i, s in enumerate(upsample_scales):
        x, c = tade_res_block(sd, f"blocks.{i}", x, c, s, dilation, gated_function)
    k = get_weight(sd, "output_conv.0").shape[-1]
    return torch.tanh(F.conv1d(x, get_weight(sd, "output_conv.0"), get_bias(sd, "output_conv.0"), padding=(k - 1) // 2))


def style_melgan_discriminator(sd, x, starts, repeats=2, window_sizes=(512, 1024, 2048, 4096),
                               pqmf_params=((1, None, None, None), (2, 62, 0.26700, 9.0), (4, 62, 0.14200, 9.0),
                                            (8, 62, 0.07949, 9.0)), discriminator_params=None, **_unused):
    """``StyleMelGANDiscriminator.forward`` models/style_melgan.py:307-337 with the window start indices
    (``np.random.randint`` draws of the reference, in call order) given explicitly."""
    dp = dict(discriminator_params or {})
    outs = []
    it = iter(starts)
    for _ in range(repeats):
        for idx, (ws, pq) in enumerate(zip(window_sizes, pqmf_params)):
            s = next(it)
            x_ = x[:, :, s: s + ws]
            if idx != 0:
                x_ = pqmf_analysis(x_, *pq)
            outs.append(melgan_discriminator(sd, f"discriminators.{idx}", x_, kernel_sizes=dp.get("kernel_sizes", (5, 3)),
                                             downsample_scales=dp.get("downsample_scales", (4, 4, 4, 1)),
                                             slope=dp.get("nonlinear_activation_params", {}).get("negative_slope", 0.2)))
    return outs


# ----------------------------------------------------------------------------
# UHiFiGAN (models/uhifigan.py:261-297; eval mode: dropout is the identity)
# ----------------------------------------------------------------------------
def uhifigan_generator(sd, c, excitation, kernel_size=7, downsample_scales=(8, 8, 2, 2),
                       downsample_kernel_sizes=(16, 16, 4, 4), upsample_scales=(8, 8, 2, 2),
                       upsample_kernel_sizes=(16, 16, 4, 4), resblock_kernel_sizes=(3, 7, 11),
                       resblock_dilations=((1, 3, 5), (1, 3, 5), (1, 3, 5)), use_additional_convs=True, slope=0.1,
                       **_unused):
    """``UHiFiGANGenerator.forward`` (f0 is unused by the reference's forward)."""
    nb = len(resblock_kernel_sizes)

    def mrf(prefix, i, x):
        cs = 0.0
        for j in range(nb):
            cs = cs + hifigan_resblock(sd, f"{prefix}.{i * nb + j}", x, resblock_kernel_sizes[j], resblock_dilations[j],
                                       slope, use_additional_convs)
        return cs / nb

    hidden = F.leaky_relu(F.conv1d(excitation, get_weight(sd, "input_conv.0"), get_bias(sd, "input_conv.0"),
                                   padding=(kernel_size - 1) // 2), slope)
    residuals = []
    for i, s in enumerate(downsample_scales):
        hidden = mrf("downsamples_mrf", i, hidden)
        p = f"downsamples.{i}.0"
        hidden = F.leaky_relu(F.conv1d(hidden, get_weight(sd, p), get_bias(sd, p), stride=s, padding=s // 2 + s % 2), slope)
        residuals.append(hidden)
    residuals.reverse()
    h = F.conv1d(c, get_weight(sd, "hidden_conv"), get_bias(sd, "hidden_conv"), padding=(kernel_size - 1) // 2)
    for i, s in enumerate(upsample_scales):
        h = torch.cat((h, residuals[i]), dim=1)
        p = f"upsamples.{i}.1"
        h = F.conv_transpose1d(F.leaky_relu(h, slope), get_weight(sd, p), get_bias(sd, p), stride=s,
                               padding=s // 2 + s % 2, output_padding=s % 2)
        h = mrf("upsamples_mrf", i, h)
    p = "output_conv.1"
    return torch.tanh(F.conv1d(F.leaky_relu(h, 0.01), get_weight(sd, p), get_bias(sd, p), padding=(kernel_size - 1) // 2))
