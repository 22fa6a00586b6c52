"""StyleMelGAN generator / discriminator (drop-in for parallel_wavegan.models.style_melgan).

Same constructor kwargs, module tree and state-dict keys as the reference
(/root/reference/parallel_wavegan/models/style_melgan.py:18-362).
"""
import copy
import logging

import numpy as np
import torch

from .. import functional as Fn
from ..layers.activation import FusedActivation
from ..layers.conv import Conv1d, ConvTranspose1d
from ..layers.pqmf import PQMF
from ..layers.tade_res_block import TADEResBlock
from .melgan import MelGANDiscriminator as BaseDiscriminator

__all__ = ["StyleMelGANGenerator", "StyleMelGANDiscriminator"]


def _each_conv(module):
    for m in module.modules():
        if isinstance(m, (Conv1d, ConvTranspose1d)):
            yield m


class _NormMixin:
    def apply_weight_norm(self):
        for m in _each_conv(self):
            m.apply_weight_norm()
            logging.debug(f"Weight norm is applied to {m}.")

    def remove_weight_norm(self):
        for m in _each_conv(self):
            if m.has_weight_norm:
                m.remove_weight_norm()
                logging.debug(f"Weight norm is removed from {m}.")

    def reset_parameters(self):
        """N(0, 0.02) on conv weights (style_melgan.py:169-179); as in the reference this only reaches
        plain ``weight`` parameters, i.e. it is a no-op once weight norm has been applied."""
        for m in _each_conv(self):
            if not m.has_weight_norm:
                with torch.no_grad():
                    m.weight.normal_(0.0, 0.02)


class StyleMelGANGenerator(torch.nn.Module, _NormMixin):
    """Noise (B, in_channels, T/44) is upsampled by transposed convolutions to the mel rate, then 9 TADE
    residual blocks modulate it with the (progressively upsampled) mel and upsample x256."""

    def __init__(self, in_channels=128, aux_channels=80, channels=64, out_channels=1, kernel_size=9, dilation=2,
                 bias=True, noise_upsample_scales=[11, 2, 2, 2], noise_upsample_activation="LeakyReLU",
                 noise_upsample_activation_params={"negative_slope": 0.2},
                 upsample_scales=[2, 2, 2, 2, 2, 2, 2, 2, 1], upsample_mode="nearest", gated_function="softmax",
                 use_weight_norm=True):
        super().__init__()
        self.in_channels = in_channels
        noise_upsample = []
        in_chs = in_channels
        for s in noise_upsample_scales:
            noise_upsample.append(ConvTranspose1d(in_chs, channels, s * 2, stride=s, padding=s // 2 + s % 2,
                                                  output_padding=s % 2, bias=bias))
            noise_upsample.append(FusedActivation(noise_upsample_activation, **noise_upsample_activation_params))
            in_chs = channels
        self.noise_upsample = torch.nn.Sequential(*noise_upsample)
        self.noise_upsample_factor = int(np.prod(noise_upsample_scales))
        self.blocks = torch.nn.ModuleList()
        aux_chs = aux_channels
        for s in upsample_scales:
            self.blocks.append(TADEResBlock(in_channels=channels, aux_channels=aux_chs, kernel_size=kernel_size,
                                            dilation=dilation, bias=bias, upsample_factor=s,
                                            upsample_mode=upsample_mode, gated_function=gated_function))
            aux_chs = channels
        self.upsample_factor = int(np.prod(upsample_scales))
        self.output_conv = torch.nn.Sequential(
            Conv1d(channels, out_channels, kernel_size, 1, bias=bias, padding=(kernel_size - 1) // 2),
            torch.nn.Identity(),  # the (fused) Tanh of the reference
        )
        if use_weight_norm:
            self.apply_weight_norm()
        self.reset_parameters()

    def _noise_upsample(self, z):
        mods = list(self.noise_upsample)
        for i in range(0, len(mods), 2):
            act = mods[i + 1]
            z = mods[i](z, post_act=act.kind, post_slope=act.slope)
        return z

    def forward(self, c, z=None):
        """c (B, aux_channels, T'), z (B, in_channels, T' / noise_upsample_factor) -> (B, out, T' * 256)."""
        if z is None:
            z = torch.randn(c.size(0), self.in_channels, 1).to(device=c.device, dtype=c.dtype)
        x = self._noise_upsample(z)
        for block in self.blocks:
            x, c = block(x, c)
        return self.output_conv[0](x, post_act="tanh")

    def register_stats(self, stats):
        from ..utils import load_stats

        mean, scale = load_stats(stats)
        self.register_buffer("mean", torch.from_numpy(mean).float())
        self.register_buffer("scale", torch.from_numpy(scale).float())
        logging.info("Successfully registered stats as buffer.")

    def inference(self, c, normalize_before=False, z=None):
        """c (T', aux_channels) -> (T' * upsample_factor, out_channels).  ``z`` (optional) fixes the noise
        (1, in_channels, ceil(T' / noise_upsample_factor)); the reference always draws it."""
        dev = next(self.parameters()).device
        if not isinstance(c, torch.Tensor):
            c = torch.tensor(c, dtype=torch.float).to(dev)
        if normalize_before:
            c = (c - self.mean) / self.scale
        c = c.transpose(1, 0).unsqueeze(0).contiguous()
        if z is None:
            z = torch.randn(1, self.in_channels, (c.size(2) - 1) // self.noise_upsample_factor + 1,
                            dtype=torch.float).to(dev)
        x = self._noise_upsample(z)
        total_length = c.size(2) * self.upsample_factor
        # pad the features to the noise length (replicate) and cut the audio afterwards (style_melgan.py:226-238)
        c = Fn.pad1d(c, 0, x.size(2) - c.size(2), "replicate")
        for block in self.blocks:
            x, c = block(x, c)
        x = self.output_conv[0](x, post_act="tanh")[..., :total_length]
        return x.squeeze(0).transpose(1, 0)


class StyleMelGANDiscriminator(torch.nn.Module, _NormMixin):
    """Random-window discriminators on PQMF sub-bands (style_melgan.py:243-362).  Window positions are
    drawn with ``np.random.randint`` exactly like the reference (same draws for the same numpy seed)."""

    # the window positions are chosen on the host at every call: a captured hipGraph would freeze them
    hip_graph_safe = False

    def __init__(self, repeats=2, window_sizes=[512, 1024, 2048, 4096],
                 pqmf_params=[[1, None, None, None], [2, 62, 0.26700, 9.0], [4, 62, 0.14200, 9.0],
                              [8, 62, 0.07949, 9.0]],
                 discriminator_params={"out_channels": 1, "kernel_sizes": [5, 3], "channels": 16,
                                       "max_downsample_channels": 512, "bias": True,
                                       "downsample_scales": [4, 4, 4, 1], "nonlinear_activation": "LeakyReLU",
                                       "nonlinear_activation_params": {"negative_slope": 0.2},
                                       "pad": "ReflectionPad1d", "pad_params": {}},
                 use_weight_norm=True):
        super().__init__()
        assert len(window_sizes) == len(pqmf_params)
        sizes = [ws // p[0] for ws, p in zip(window_sizes, pqmf_params)]
        assert len(window_sizes) == sum([sizes[0] == size for size in sizes])
        self.repeats = repeats
        self.window_sizes = window_sizes
        self.pqmfs = torch.nn.ModuleList()
        self.discriminators = torch.nn.ModuleList()
        for pqmf_param in pqmf_params:
            d_params = copy.deepcopy(discriminator_params)
            d_params["in_channels"] = pqmf_param[0]
            self.pqmfs.append(torch.nn.Identity() if pqmf_param[0] == 1 else PQMF(*pqmf_param))
            self.discriminators.append(BaseDiscriminator(**d_params))
        if use_weight_norm:
            self.apply_weight_norm()
        self.reset_parameters()

    def forward(self, x):
        outs = []
        for _ in range(self.repeats):
            outs += self._forward(x)
        return outs

    def _forward(self, x):
        outs = []
        for idx, (ws, pqmf, disc) in enumerate(zip(self.window_sizes, self.pqmfs, self.discriminators)):
            start_idx = np.random.randint(x.size(-1) - ws)
            x_ = x[:, :, start_idx: start_idx + ws].contiguous()
            x_ = pqmf(x_) if idx == 0 else pqmf.analysis(x_)
            outs.append(disc(x_))
        return outs
