"""Mel upsampling network of Parallel WaveGAN (drop-in names for parallel_wavegan.layers.upsample).

The reference materialises ``F.interpolate(nearest)`` and then runs a single-channel
``Conv2d(1, 1, (1, 2s+1))`` per scale (layers/upsample.py:16-128); here each scale is ONE
HBM-bound HIP kernel (stretch + smoothing conv fused, functional.StretchConvFn).  Parameter names
and shapes are kept (``up_layers.{1,3,..}.weight`` of shape (1, 1, 1, 2s+1), weight-normed by the
generator), so reference checkpoints load unchanged.
"""
import numpy as np
import torch

from .. import functional as Fn
from .residual_block import Conv1d


class Stretch2d(torch.nn.Module):
    """Marker for the nearest-neighbour stretch (fused into the following Conv2d kernel)."""

    def __init__(self, x_scale, y_scale, mode="nearest"):
        super().__init__()
        if mode != "nearest" or y_scale != 1:
            raise NotImplementedError("only nearest stretching along time has a gfx950 kernel")
        self.x_scale, self.y_scale, self.mode = x_scale, y_scale, mode

    def forward(self, x):  # pragma: no cover
        raise RuntimeError("Stretch2d is fused into the smoothing convolution kernel")


class Conv2d(torch.nn.Module):
    """Single-channel (1, k) smoothing Conv2d, weight (1, 1, 1, k) initialised to 1/k, no bias."""

    def __init__(self, in_channels, out_channels, kernel_size, padding=0, bias=False):
        super().__init__()
        if in_channels != 1 or out_channels != 1 or kernel_size[0] != 1 or bias:
            raise NotImplementedError("upsample Conv2d: only (1 -> 1, (1, k), bias=False) is on the hot path")
        self.kernel_size = tuple(kernel_size)
        self.weight = torch.nn.Parameter(torch.full((1, 1, 1, kernel_size[1]), 1.0 / np.prod(kernel_size)))

    @property
    def has_weight_norm(self):
        return "weight_g" in self._parameters

    def apply_weight_norm(self):
        if not self.has_weight_norm:
            w = self._parameters.pop("weight")
            self.weight_g = torch.nn.Parameter(w.detach().reshape(1, -1).norm(dim=1).reshape(1, 1, 1, 1).clone())
            self.weight_v = torch.nn.Parameter(w.detach().clone())
        return self

    def remove_weight_norm(self):
        if not self.has_weight_norm:
            raise ValueError("weight norm is not applied")
        g, v = self._parameters.pop("weight_g"), self._parameters.pop("weight_v")
        self.weight = torch.nn.Parameter(v.detach() * (g.detach() / v.detach().norm()))
        return self

    def weight_tensor(self):
        if self.has_weight_norm:
            return Fn.WeightNormFn.apply(self.weight_v, self.weight_g)
        return self.weight


class UpsampleNetwork(torch.nn.Module):
    def __init__(self, upsample_scales, nonlinear_activation=None, nonlinear_activation_params={},
                 interpolate_mode="nearest", freq_axis_kernel_size=1, use_causal_conv=False):
        super().__init__()
        if nonlinear_activation is not None or freq_axis_kernel_size != 1:
            raise NotImplementedError("UpsampleNetwork: only the configuration of the shipped YAMLs is accelerated")
        self.use_causal_conv = use_causal_conv
        self.up_layers = torch.nn.ModuleList()
        for scale in upsample_scales:
            self.up_layers.append(Stretch2d(scale, 1, interpolate_mode))
            # causal: padding (0, 2*scale) and the output trimmed to the stretched length
            # (layers/upsample.py:96-99,121-125) == left-only padding of 2*scale inside the kernel
            self.up_layers.append(Conv2d(1, 1, kernel_size=(1, scale * 2 + 1),
                                         padding=(0, scale * 2 if use_causal_conv else scale), bias=False))

    def forward(self, c):
        """c: (B, C, T) -> (B, C, T * prod(upsample_scales))."""
        for i in range(0, len(self.up_layers), 2):
            scale = self.up_layers[i].x_scale
            c = Fn.StretchConvFn.apply(c, self.up_layers[i + 1].weight_tensor(), scale,
                                       2 * scale if self.use_causal_conv else scale)
        return c


class ConvInUpsampleNetwork(torch.nn.Module):
    """Context conv (k = 2*aux_context_window+1, no padding) + UpsampleNetwork (layers/upsample.py:131-194)."""

    def __init__(self, upsample_scales, nonlinear_activation=None, nonlinear_activation_params={},
                 interpolate_mode="nearest", freq_axis_kernel_size=1, aux_channels=80, aux_context_window=0,
                 use_causal_conv=False):
        super().__init__()
        self.aux_context_window = aux_context_window
        self.use_causal_conv = use_causal_conv and aux_context_window > 0
        # causal: kernel aux_context_window + 1 and the last aux_context_window outputs dropped
        # (layers/upsample.py:160-164,192-193): a negative right "padding" makes the kernel stop there
        kernel_size = aux_context_window + 1 if use_causal_conv else 2 * aux_context_window + 1
        self.conv_in = Conv1d(aux_channels, aux_channels, kernel_size=kernel_size, bias=False,
                              padding=(0, -aux_context_window) if self.use_causal_conv else 0)
        self.upsample = UpsampleNetwork(upsample_scales=upsample_scales, nonlinear_activation=nonlinear_activation,
                                        nonlinear_activation_params=nonlinear_activation_params,
                                        interpolate_mode=interpolate_mode, freq_axis_kernel_size=freq_axis_kernel_size,
                                        use_causal_conv=use_causal_conv)

    def forward(self, c):
        return self.upsample(self.conv_in(c))
