"""Parallel WaveGAN modules on the gfx950 kernel library (drop-in for
``parallel_wavegan.models.parallel_wavegan``; reference: models/parallel_wavegan.py)."""
import logging
import math

import numpy as np
import torch

from .. import functional as Fn
from ..layers import upsample
from ..layers.activation import FusedActivation
from ..layers.conv import Conv1d as _AnyConv1d
from ..layers.residual_block import Conv1d, Conv1d1x1
from ..layers.residual_block import WaveNetResidualBlock as ResidualBlock
from ..layers.upsample import Conv2d as UpsampleConv2d


def _norm_modules(module):
    # every Conv1d / Conv2d of the tree, as the reference's ``isinstance(m, torch.nn.Conv1d) or ... Conv2d``
    # (models/parallel_wavegan.py:187-195): includes the convolutions of a MelGAN upsampler, not its
    # transposed convolutions
    for m in module.modules():
        if isinstance(m, (_AnyConv1d, UpsampleConv2d)):
            yield m


class _WeightNormMixin:
    def remove_weight_norm(self):
        for m in _norm_modules(self):
            if m.has_weight_norm:
                m.remove_weight_norm()
                logging.debug(f"Weight norm is removed from {m}.")

    def apply_weight_norm(self):
        for m in _norm_modules(self):
            m.apply_weight_norm()
            logging.debug(f"Weight norm is applied to {m}.")


class ParallelWaveGANGenerator(torch.nn.Module, _WeightNormMixin):
    """Non-autoregressive WaveNet generator (reference: models/parallel_wavegan.py:21-261)."""

    def __init__(self, in_channels=1, out_channels=1, kernel_size=3, layers=30, stacks=3, residual_channels=64,
                 gate_channels=128, skip_channels=64, aux_channels=80, aux_context_window=2, dropout=0.0, bias=True,
                 use_weight_norm=True, use_causal_conv=False, upsample_conditional_features=True,
                 upsample_net="ConvInUpsampleNetwork", upsample_params={"upsample_scales": [4, 4, 4, 4]}):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.aux_channels, self.aux_context_window = aux_channels, aux_context_window
        self.layers, self.stacks, self.kernel_size = layers, stacks, kernel_size
        assert layers % stacks == 0
        layers_per_stack = layers // stacks
        self.first_conv = Conv1d1x1(in_channels, residual_channels, bias=True)
        if upsample_conditional_features:
            upsample_params = dict(upsample_params)
            upsample_params.update({"use_causal_conv": use_causal_conv})
            if upsample_net == "MelGANGenerator":
                # a MelGAN generator as the mel upsampler (models/parallel_wavegan.py:90-98 of the reference):
                # its out_channels must equal aux_channels; weight norm is applied once, by this model
                assert aux_context_window == 0
                from .melgan import MelGANGenerator

                upsample_params.update({"use_weight_norm": False, "use_final_nonlinear_activation": False})
                self.upsample_net = MelGANGenerator(**upsample_params)
            else:
                if upsample_net == "ConvInUpsampleNetwork":
                    upsample_params.update({"aux_channels": aux_channels, "aux_context_window": aux_context_window})
                self.upsample_net = getattr(upsample, upsample_net)(**upsample_params)
            self.upsample_factor = int(np.prod(upsample_params["upsample_scales"]))
        else:
            self.upsample_net = None
            self.upsample_factor = 1
        self.conv_layers = torch.nn.ModuleList()
        for layer in range(layers):
            self.conv_layers.append(ResidualBlock(
                kernel_size=kernel_size, residual_channels=residual_channels, gate_channels=gate_channels,
                skip_channels=skip_channels, aux_channels=aux_channels, dilation=2 ** (layer % layers_per_stack),
                dropout=dropout, bias=bias, use_causal_conv=use_causal_conv))
        self.last_conv_layers = torch.nn.ModuleList([
            FusedActivation("ReLU"),
            Conv1d1x1(skip_channels, skip_channels, bias=True),
            FusedActivation("ReLU"),
            Conv1d1x1(skip_channels, out_channels, bias=True),
        ])
        if use_weight_norm:
            self.apply_weight_norm()

    def forward(self, z, c):
        """z: noise (B, 1, T), c: mel (B, C, T' + 2*aux_context_window) -> (B, out_channels, T)."""
        if c is not None and self.upsample_net is not None:
            c = self.upsample_net(c)
            assert c.size(-1) == z.size(-1)
        x = self.first_conv(z)
        skips = None
        n = len(self.conv_layers)
        for i, f in enumerate(self.conv_layers):
            # `skips += h` and the final `skips *= sqrt(1/n)` ride on the skip conv's epilogue
            # (c is handed from layer to layer so that its gradient is summed along that chain, see WaveNetLayerFn)
            x, skips, c = f(x, c, skips=skips, skip_scale=math.sqrt(1.0 / n) if i == n - 1 else 1.0, chain_aux=True,
                            inplace_skips=True)
        x = self.last_conv_layers[1](skips, pre_act="relu")
        return self.last_conv_layers[3](x, pre_act="relu")

    @staticmethod
    def _get_receptive_field_size(layers, stacks, kernel_size, dilation=lambda x: 2 ** x):
        assert layers % stacks == 0
        per_cycle = layers // stacks
        return (kernel_size - 1) * sum(dilation(i % per_cycle) for i in range(layers)) + 1

    @property
    def receptive_field_size(self):
        return self._get_receptive_field_size(self.layers, self.stacks, self.kernel_size)

    def register_stats(self, stats):
        from ..utils import load_stats

        mean, scale = load_stats(stats)
        self.register_buffer("mean", torch.from_numpy(mean).float())
        self.register_buffer("scale", torch.from_numpy(scale).float())
        logging.info("Successfully registered stats as buffer.")

    def inference(self, c=None, x=None, normalize_before=False):
        """c: (T', C) mel, x: (T, 1) noise (drawn here if omitted) -> (T, out_channels)."""
        dev = next(self.parameters()).device
        if x is not None:
            if not isinstance(x, torch.Tensor):
                x = torch.tensor(x, dtype=torch.float).to(dev)
            x = x.transpose(1, 0).unsqueeze(0).contiguous()
        else:
            assert c is not None
            x = torch.randn(1, 1, len(c) * self.upsample_factor).to(dev)
        if c is not None:
            if not isinstance(c, torch.Tensor):
                c = torch.tensor(c, dtype=torch.float).to(dev)
            if normalize_before:
                c = (c - self.mean) / self.scale
            c = c.transpose(1, 0).unsqueeze(0).contiguous()
            c = Fn.pad1d(c, self.aux_context_window, self.aux_context_window, "replicate")
        return self.forward(x, c).squeeze(0).transpose(1, 0)


class ParallelWaveGANDiscriminator(torch.nn.Module, _WeightNormMixin):
    """Dilated conv stack discriminator (reference: models/parallel_wavegan.py:264-371)."""

    def __init__(self, in_channels=1, out_channels=1, kernel_size=3, layers=10, conv_channels=64, dilation_factor=1,
                 nonlinear_activation="LeakyReLU", nonlinear_activation_params={"negative_slope": 0.2}, bias=True,
                 use_weight_norm=True):
        super().__init__()
        assert (kernel_size - 1) % 2 == 0, "Not support even number kernel size."
        assert dilation_factor > 0, "Dilation factor must be > 0."
        self.conv_layers = torch.nn.ModuleList()
        conv_in_channels = in_channels
        for i in range(layers - 1):
            if i == 0:
                dilation = 1
            else:
                dilation = i if dilation_factor == 1 else dilation_factor ** i
                conv_in_channels = conv_channels
            self.conv_layers.append(Conv1d(conv_in_channels, conv_channels, kernel_size=kernel_size,
                                           padding=(kernel_size - 1) // 2 * dilation, dilation=dilation, bias=bias))
            self.conv_layers.append(FusedActivation(nonlinear_activation, **nonlinear_activation_params))
        self.conv_layers.append(Conv1d(conv_in_channels, out_channels, kernel_size=kernel_size,
                                       padding=(kernel_size - 1) // 2, bias=bias))
        if use_weight_norm:
            self.apply_weight_norm()

    def forward(self, x):
        """(B, 1, T) -> (B, 1, T)."""
        # each LeakyReLU rides on the NEXT convolution as its pre-activation (applied to the staged input tile), so
        # that the backward needs no separate activation-gradient pass: the next convolution's data-gradient kernel
        # multiplies by act'(x) in its epilogue and its weight-gradient kernel re-applies act on load.
        pending = None
        for mod in self.conv_layers:
            if isinstance(mod, FusedActivation):
                pending = mod
                continue
            if pending is None:
                x = mod(x)
            else:
                x = mod(x, pre_act=pending.kind, pre_slope=pending.slope)
            pending = None
        if pending is not None:
            raise RuntimeError("ParallelWaveGANDiscriminator: the layer list must end with a convolution")
        return x


class ResidualParallelWaveGANDiscriminator(torch.nn.Module, _WeightNormMixin):
    """WaveNet-style discriminator (reference: models/parallel_wavegan.py:374-513): 1x1 conv +
    LeakyReLU, ``layers`` gated residual blocks without conditioning, skip sum * sqrt(1/layers),
    LeakyReLU, 1x1, LeakyReLU, 1x1.  Activations, the running skip sum and its scale ride on the
    convolution kernels' epilogues exactly as in the generator."""

    def __init__(self, in_channels=1, out_channels=1, kernel_size=3, layers=30, stacks=3, residual_channels=64,
                 gate_channels=128, skip_channels=64, dropout=0.0, bias=True, use_weight_norm=True,
                 use_causal_conv=False, nonlinear_activation="LeakyReLU",
                 nonlinear_activation_params={"negative_slope": 0.2}):
        super().__init__()
        assert (kernel_size - 1) % 2 == 0, "Not support even number kernel size."
        self.in_channels, self.out_channels = in_channels, out_channels
        self.layers, self.stacks, self.kernel_size = layers, stacks, kernel_size
        assert layers % stacks == 0
        layers_per_stack = layers // stacks

        def act():
            return FusedActivation(nonlinear_activation, **nonlinear_activation_params)

        self.first_conv = torch.nn.Sequential(Conv1d1x1(in_channels, residual_channels, bias=True), act())
        self.conv_layers = torch.nn.ModuleList()
        for layer in range(layers):
            self.conv_layers.append(ResidualBlock(
                kernel_size=kernel_size, residual_channels=residual_channels, gate_channels=gate_channels,
                skip_channels=skip_channels, aux_channels=-1, dilation=2 ** (layer % layers_per_stack),
                dropout=dropout, bias=bias, use_causal_conv=use_causal_conv))
        self.last_conv_layers = torch.nn.ModuleList([
            act(), Conv1d1x1(skip_channels, skip_channels, bias=True), act(),
            Conv1d1x1(skip_channels, out_channels, bias=True)])
        if use_weight_norm:
            self.apply_weight_norm()

    def forward(self, x):
        """(B, 1, T) -> (B, 1, T)."""
        a0 = self.first_conv[1]
        x = self.first_conv[0](x, post_act=a0.kind, post_slope=a0.slope)
        skips = None
        n = len(self.conv_layers)
        for i, f in enumerate(self.conv_layers):
            x, skips = f(x, None, skips=skips, skip_scale=math.sqrt(1.0 / n) if i == n - 1 else 1.0)
        a1, c1, a2, c2 = self.last_conv_layers
        x = c1(skips, pre_act=a1.kind, pre_slope=a1.slope)
        return c2(x, pre_act=a2.kind, pre_slope=a2.slope)
