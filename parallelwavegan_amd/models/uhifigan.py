"""Unet-based HiFi-GAN generator (drop-in for parallel_wavegan.models.uhifigan).

Same constructor kwargs, module tree and state-dict keys as the reference
(/root/reference/parallel_wavegan/models/uhifigan.py:19-387).  The excitation signal is encoded by a
downsampling MRF path whose stages are concatenated to the mel-conditioned decoder path.  Convolutions
run on the MFMA kernel with the activations fused; the channel concatenation and the training-mode
dropout are HIP kernels.  ``use_causal_conv=True`` raises in the reference itself (its CausalConv1d has
no padding / stride arguments, uhifigan.py:93-99), so only the non-causal model exists.
"""
import logging

import numpy as np
import torch

from .. import functional as Fn
from ..layers.activation import FusedActivation
from ..layers.conv import Conv1d, ConvTranspose1d
from ..layers.dropout import Dropout as _Dropout
from ..layers.residual_block import HiFiGANResidualBlock as ResidualBlock

__all__ = ["UHiFiGANGenerator"]


def _each_conv(module):
    for m in module.modules():
        if isinstance(m, (Conv1d, ConvTranspose1d)):
            yield m


class UHiFiGANGenerator(torch.nn.Module):
    def __init__(self, in_channels=80, out_channels=1, channels=512, kernel_size=7, downsample_scales=(8, 8, 2, 2),
                 downsample_kernel_sizes=(16, 16, 4, 4), upsample_scales=(8, 8, 2, 2),
                 upsample_kernel_sizes=(16, 16, 4, 4), resblock_kernel_sizes=(3, 7, 11),
                 resblock_dilations=[(1, 3, 5), (1, 3, 5), (1, 3, 5)], dropout=0.3, use_additional_convs=True,
                 bias=True, nonlinear_activation="LeakyReLU", nonlinear_activation_params={"negative_slope": 0.1},
                 use_causal_conv=False, use_weight_norm=True):
        super().__init__()
        assert kernel_size % 2 == 1, "Kernel size must be odd number."
        assert len(upsample_scales) == len(upsample_kernel_sizes)
        assert len(resblock_dilations) == len(resblock_kernel_sizes)
        if use_causal_conv:
            raise NotImplementedError("UHiFiGANGenerator(use_causal_conv=True) cannot be constructed in the "
                                      "reference either (CausalConv1d takes no padding/stride arguments)")
        self.num_upsamples = len(upsample_kernel_sizes)
        self.num_blocks = len(resblock_kernel_sizes)
        self.use_causal_conv = use_causal_conv
        self.upsample_factor = int(np.prod(upsample_scales))

        def act():
            return FusedActivation(nonlinear_activation, **nonlinear_activation_params)

        def mrf(ch):
            return [ResidualBlock(kernel_size=k, channels=ch, dilations=d, bias=bias,
                                  use_additional_convs=use_additional_convs, nonlinear_activation=nonlinear_activation,
                                  nonlinear_activation_params=nonlinear_activation_params, use_causal_conv=False)
                    for k, d in zip(resblock_kernel_sizes, resblock_dilations)]

        self.downsamples = torch.nn.ModuleList()
        self.downsamples_mrf = torch.nn.ModuleList()
        self.upsamples = torch.nn.ModuleList()
        self.upsamples_mrf = torch.nn.ModuleList()
        self.input_conv = torch.nn.Sequential(
            Conv1d(out_channels, channels, kernel_size, bias=bias, padding=(kernel_size - 1) // 2), act(),
            _Dropout(dropout))
        for s, k in zip(downsample_scales, downsample_kernel_sizes):
            self.downsamples_mrf.extend(mrf(channels))
            self.downsamples.append(torch.nn.Sequential(
                Conv1d(channels, channels * 2, k, stride=s, bias=bias, padding=s // 2 + s % 2), act(), _Dropout(dropout)))
            channels = channels * 2
        self.hidden_conv = Conv1d(in_channels, channels, kernel_size, bias=bias, padding=(kernel_size - 1) // 2)
        for s, k in zip(upsample_scales, upsample_kernel_sizes):
            self.upsamples.append(torch.nn.Sequential(
                act(), ConvTranspose1d(channels * 2, channels // 2, k, s, padding=s // 2 + s % 2, output_padding=s % 2,
                                       bias=bias)))
            self.upsamples_mrf.extend(mrf(channels // 2))
            channels = channels // 2
        self.output_conv = torch.nn.Sequential(
            FusedActivation("LeakyReLU"),  # default slope 0.01, as the reference (uhifigan.py:228-232)
            Conv1d(channels, out_channels, kernel_size, bias=bias, padding=(kernel_size - 1) // 2),
            torch.nn.Identity(),  # the (fused) Tanh
        )
        if use_weight_norm:
            self.apply_weight_norm()
        self.reset_parameters()

    def _mrf(self, blocks, i, x):
        cs = None
        nb = self.num_blocks
        for j in range(nb):
            cs = blocks[i * nb + j](x, accum=cs, out_div=float(nb) if j == nb - 1 else 1.0)
        return cs

    def forward(self, c=None, f0=None, excitation=None):
        """c (B, in_channels, T'), excitation (B, out_channels, T' * prod(scales)) -> (B, out_channels, T)."""
        a0 = self.input_conv[1]
        hidden = self.input_conv[2](self.input_conv[0](excitation, post_act=a0.kind, post_slope=a0.slope))
        residual_results = []
        for i, down in enumerate(self.downsamples):
            hidden = self._mrf(self.downsamples_mrf, i, hidden)
            hidden = down[2](down[0](hidden, post_act=down[1].kind, post_slope=down[1].slope))
            residual_results.append(hidden)
        residual_results.reverse()
        hidden_mel = self.hidden_conv(c)
        for i, up in enumerate(self.upsamples):
            hidden_mel = Fn.ConcatChannelsFn.apply(hidden_mel, residual_results[i])
            hidden_mel = up[1](hidden_mel, pre_act=up[0].kind, pre_slope=up[0].slope)
            hidden_mel = self._mrf(self.upsamples_mrf, i, hidden_mel)
        act, conv = self.output_conv[0], self.output_conv[1]
        return conv(hidden_mel, pre_act=act.kind, pre_slope=act.slope, post_act="tanh")

    def reset_parameters(self):
        """N(0, 0.01) on conv weights; with weight norm applied this touches no trainable parameter,
        exactly as the reference's version (it writes the derived ``.weight``)."""
        for m in _each_conv(self):
            if not m.has_weight_norm:
                with torch.no_grad():
                    m.weight.normal_(0.0, 0.01)

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
        from ..utils import load_stats

        mean, scale = load_stats(stats)
        self.register_buffer("mean", torch.from_numpy(mean).float())
        self.register_buffer("scale", torch.from_numpy(scale).float())
        logging.info("Successfully registered stats as buffer.")

    def inference(self, excitation=None, f0=None, c=None, normalize_before=False):
        """excitation (T, 1), c (T', in_channels) -> (T, out_channels)  (uhifigan.py:351-387)."""
        dev = next(self.parameters()).device

        def prep(v):
            if v is None:
                return None
            if not isinstance(v, torch.Tensor):
                v = torch.tensor(v, dtype=torch.float)
            return v.to(dev)

        c, f0, excitation = prep(c), prep(f0), prep(excitation)
        # (like the reference, `normalize_before` is accepted but the features are used as given)
        c = self.forward(c.transpose(1, 0).unsqueeze(0).contiguous(),
                         None if f0 is None else f0.reshape(1, 1, -1),
                         excitation.reshape(1, 1, -1).contiguous())
        return c.squeeze(0).transpose(1, 0)
