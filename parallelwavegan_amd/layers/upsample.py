"""Mel upsampling network of Parallel WaveGAN (drop-in names for parallel_wavegan.layers.upsample).

The reference materialises ``F.interpolate(nearest)`` and then runs a single-channel
``Conv2d(1, 1, (F, 2s+1))`` (+ an optional activation) per scale (layers/upsample.py:16-128); here each
scale is ONE HBM-bound HIP kernel (stretch + smoothing conv + activation fused, functional.StretchConvFn).  Parameter names
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
    """``Conv2d`` of the reference's upsampler (layers/upsample.py:47-59): weight initialised to 1 / prod(kernel_size),
    bias to 0.  On the hot path it is the single-channel (F, k) smoothing convolution that follows a ``Stretch2d`` and
    runs fused with it (``UpsampleNetwork.forward``); any other shape can be constructed and initialised (the
    reference's own unit test does) but has no stand-alone gfx950 kernel."""

    def __init__(self, in_channels, out_channels, kernel_size, padding=0, bias=False, **kwargs):
        super().__init__()
        if kwargs:
            raise NotImplementedError(f"upsample Conv2d: unsupported arguments {sorted(kwargs)}")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        self.padding = padding
        self.weight = torch.nn.Parameter(torch.empty((out_channels, in_channels) + self.kernel_size))
        self.bias = torch.nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        self.weight.data.fill_(1.0 / np.prod(self.kernel_size))
        if self.bias is not None:
            torch.nn.init.constant_(self.bias, 0.0)

    @property
    def fusable(self):
        """(1 -> 1, no bias): the form ``UpsampleNetwork`` fuses with the preceding stretch."""
        return self.in_channels == 1 and self.out_channels == 1 and self.bias is None

    def forward(self, x):  # pragma: no cover
        raise NotImplementedError("upsample Conv2d runs fused with the preceding Stretch2d inside UpsampleNetwork; a "
                                  "stand-alone Conv2d has no gfx950 kernel")

    @property
    def has_weight_norm(self):
        return "weight_g" in self._parameters

    def apply_weight_norm(self):
        if not self.has_weight_norm:
            w = self._parameters.pop("weight")
            n0 = w.shape[0]
            self.weight_g = torch.nn.Parameter(w.detach().reshape(n0, -1).norm(dim=1).reshape(n0, 1, 1, 1).clone())
            self.weight_v = torch.nn.Parameter(w.detach().clone())
        return self

    def remove_weight_norm(self):
        if not self.has_weight_norm:
            raise ValueError("weight norm is not applied")
        g, v = self._parameters.pop("weight_g"), self._parameters.pop("weight_v")
        n0 = v.shape[0]
        self.weight = torch.nn.Parameter(v.detach() * (g.detach() / v.detach().reshape(n0, -1).norm(dim=1).reshape(n0, 1, 1, 1)))
        return self

    def weight_tensor(self):
        if self.has_weight_norm:
            return Fn.WeightNormFn.apply(self.weight_v, self.weight_g)
        return self.weight


_STAGE_ACTS = {"LeakyReLU": "leaky_relu", "ReLU": "relu", "Tanh": "tanh"}


class UpsampleNetwork(torch.nn.Module):
    """Stretch2d + (freq_axis_kernel_size, 2*scale+1) Conv2d [+ nonlinearity] per scale (layers/upsample.py:62-128);
    ``up_layers`` keeps the reference's indices (the activation modules carry no parameters)."""

    def __init__(self, upsample_scales, nonlinear_activation=None, nonlinear_activation_params={},
                 interpolate_mode="nearest", freq_axis_kernel_size=1, use_causal_conv=False):
        super().__init__()
        assert (freq_axis_kernel_size - 1) % 2 == 0, "Not support even number freq axis kernel size."
        self.use_causal_conv = use_causal_conv
        self.act, self.slope = None, 0.0
        if nonlinear_activation is not None:
            if nonlinear_activation not in _STAGE_ACTS:
                raise NotImplementedError(f"UpsampleNetwork: activation {nonlinear_activation!r} has no gfx950 epilogue "
                                          f"(supported: {sorted(_STAGE_ACTS)})")
            self.act = _STAGE_ACTS[nonlinear_activation]
            if self.act == "leaky_relu":
                self.slope = float(nonlinear_activation_params.get("negative_slope", 0.01))
                if not self.slope > 0.0:
                    raise NotImplementedError("UpsampleNetwork: LeakyReLU needs a positive slope (its gradient mask is "
                                              "taken from the stage output)")
        self.up_layers = torch.nn.ModuleList()
        self._stages = []
        freq_axis_padding = (freq_axis_kernel_size - 1) // 2
        for scale in upsample_scales:
            self._stages.append(len(self.up_layers))
            self.up_layers.append(Stretch2d(scale, 1, interpolate_mode))
            # causal: padding (., 2*scale) and the output trimmed to the stretched length
            # (layers/upsample.py:96-99,121-125) == left-only padding of 2*scale inside the kernel
            self.up_layers.append(Conv2d(1, 1, kernel_size=(freq_axis_kernel_size, scale * 2 + 1),
                                         padding=(freq_axis_padding, scale * 2 if use_causal_conv else scale), bias=False))
            if nonlinear_activation is not None:
                self.up_layers.append(getattr(torch.nn, nonlinear_activation)(**nonlinear_activation_params))

    def forward(self, c):
        """c: (B, C, T) -> (B, C, T * prod(upsample_scales))."""
        for i in self._stages:
            scale = self.up_layers[i].x_scale
            c = Fn.StretchConvFn.apply(c, self.up_layers[i + 1].weight_tensor(), scale,
                                       2 * scale if self.use_causal_conv else scale, self.act, self.slope)
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
