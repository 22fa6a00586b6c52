"""HiFi-GAN modules on the gfx950 kernel library.

Drop-in for ``parallel_wavegan.models.hifigan`` (constructor kwargs, method
names and state-dict keys follow /root/reference/parallel_wavegan/models/hifigan.py);
the arithmetic is hand-written HIP behind ``parallelwavegan_amd.ops``.
"""
import logging

import numpy as np
import torch

from ..layers.activation import FusedActivation
from ..layers.conv import Conv1d, ConvTranspose1d
from ..layers.residual_block import HiFiGANResidualBlock as ResidualBlock


def _each_conv(module):
    for m in module.modules():
        if isinstance(m, (Conv1d, ConvTranspose1d)):
            yield m


class HiFiGANGenerator(torch.nn.Module):
    """HiFi-GAN generator (reference: models/hifigan.py:23-267).

    Forward = 1 input conv, then per upsampling stage one polyphase
    ConvTranspose1d launch (LeakyReLU fused on its input) and
    ``2 * len(dilations)`` fused conv launches per MRF block; the MRF sum and
    ``/ num_blocks`` are folded into the last launch of each block, the final
    LeakyReLU(0.01) + conv + tanh is one launch.
    """

    def __init__(self, in_channels=80, out_channels=1, channels=512, kernel_size=7, upsample_scales=(8, 8, 2, 2),
                 upsample_kernel_sizes=(16, 16, 4, 4), resblock_kernel_sizes=(3, 7, 11),
                 resblock_dilations=[(1, 3, 5), (1, 3, 5), (1, 3, 5)], use_additional_convs=True, bias=True,
                 nonlinear_activation="LeakyReLU", nonlinear_activation_params={"negative_slope": 0.1},
                 use_causal_conv=False, use_weight_norm=True):
        super().__init__()
        assert kernel_size % 2 == 1, "Kernel size must be odd number."
        assert len(upsample_scales) == len(upsample_kernel_sizes)
        assert len(resblock_dilations) == len(resblock_kernel_sizes)
        if use_causal_conv:
            raise NotImplementedError("use_causal_conv=True is outside the accelerated path (SURVEY.md s8f-3)")
        self.num_upsamples = len(upsample_kernel_sizes)
        self.num_blocks = len(resblock_kernel_sizes)
        self.use_causal_conv = use_causal_conv
        self.upsample_factor = int(np.prod(upsample_scales))
        self.input_conv = Conv1d(in_channels, channels, kernel_size, bias=bias, padding=(kernel_size - 1) // 2)
        self.upsamples = torch.nn.ModuleList()
        self.blocks = torch.nn.ModuleList()
        ch = channels
        for i, (s, k) in enumerate(zip(upsample_scales, upsample_kernel_sizes)):
            assert k == 2 * s
            self.upsamples.append(torch.nn.Sequential(
                FusedActivation(nonlinear_activation, **nonlinear_activation_params),
                ConvTranspose1d(ch, ch // 2, k, s, padding=s // 2 + s % 2, output_padding=s % 2, bias=bias),
            ))
            ch //= 2
            for ks, dil in zip(resblock_kernel_sizes, resblock_dilations):
                self.blocks.append(ResidualBlock(
                    kernel_size=ks, channels=ch, dilations=dil, bias=bias,
                    use_additional_convs=use_additional_convs, nonlinear_activation=nonlinear_activation,
                    nonlinear_activation_params=nonlinear_activation_params, use_causal_conv=use_causal_conv))
        # the reference uses torch.nn.LeakyReLU() here, i.e. the DEFAULT slope 0.01
        # (models/hifigan.py:139-151), not nonlinear_activation_params
        self.output_conv = torch.nn.Sequential(
            FusedActivation("LeakyReLU"),
            Conv1d(ch, out_channels, kernel_size, bias=bias, padding=(kernel_size - 1) // 2),
            torch.nn.Identity(),  # index 2 is the (fused) Tanh in the reference
        )
        if use_weight_norm:
            self.apply_weight_norm()
        self.reset_parameters()

    def forward(self, c):
        """c: (B, in_channels, T) -> (B, out_channels, T * prod(upsample_scales))."""
        c = self.input_conv(c)
        nb = self.num_blocks
        for i in range(self.num_upsamples):
            act, up = self.upsamples[i][0], self.upsamples[i][1]
            c = up(c, pre_act=act.kind, pre_slope=act.slope)
            cs = None
            for j in range(nb):
                last = j == nb - 1
                cs = self.blocks[i * nb + j](c, accum=cs, out_div=float(nb) if last else 1.0)
            c = cs
        act, conv = self.output_conv[0], self.output_conv[1]
        return conv(c, pre_act=act.kind, pre_slope=act.slope, post_act="tanh")

    def reset_parameters(self):
        """N(0, 0.01) on conv weights as in the official implementation.

        With weight norm already applied the reference's version of this writes the
        *derived* ``.weight`` attribute and is therefore a no-op on the trainable
        ``weight_g``/``weight_v`` (SURVEY.md App. A "Init quirk"); reproduced here by
        only touching plain ``weight`` parameters.
        """
        for m in _each_conv(self):
            if not m.has_weight_norm:
                with torch.no_grad():
                    m.weight.normal_(0.0, 0.01)
                logging.debug(f"Reset parameters in {m}.")

    def remove_weight_norm(self):
        for m in _each_conv(self):
            if m.has_weight_norm:
                m.remove_weight_norm()
                logging.debug(f"Weight norm is removed from {m}.")

    def apply_weight_norm(self):
        for m in _each_conv(self):
            m.apply_weight_norm()
            logging.debug(f"Weight norm is applied to {m}.")

    def register_stats(self, stats):
        """Register de-normalisation stats (``.npy`` [mean; scale] or ``.h5``)."""
        assert stats.endswith(".h5") or stats.endswith(".npy")
        if stats.endswith(".h5"):
            from ..utils import read_hdf5

            mean = read_hdf5(stats, "mean").reshape(-1)
            scale = read_hdf5(stats, "scale").reshape(-1)
        else:
            mean = np.load(stats)[0].reshape(-1)
            scale = np.load(stats)[1].reshape(-1)
        self.register_buffer("mean", torch.from_numpy(mean).float())
        self.register_buffer("scale", torch.from_numpy(scale).float())
        logging.info("Successfully registered stats as buffer.")

    def inference(self, c, normalize_before=False):
        """c: (T, in_channels) tensor/ndarray -> (T * prod(upsample_scales), out_channels)."""
        if not isinstance(c, torch.Tensor):
            c = torch.tensor(c, dtype=torch.float).to(next(self.parameters()).device)
        if normalize_before:
            c = (c - self.mean) / self.scale
        c = self.forward(c.transpose(1, 0).unsqueeze(0).contiguous())
        return c.squeeze(0).transpose(1, 0)
