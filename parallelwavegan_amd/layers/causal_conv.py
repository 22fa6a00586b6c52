"""Causal convolution modules (drop-in for parallel_wavegan.layers.causal_conv).

Same constructor arguments, sub-module names (``pad`` / ``conv`` / ``deconv``) and hence state-dict
keys as the reference (/root/reference/parallel_wavegan/layers/causal_conv.py:12-77).  No tensor is
padded and trimmed: a causal convolution is the MFMA convolution kernel with left-only padding
``(k-1)*d`` and ``t_out = t_in``; the causal transposed convolution is the polyphase kernel with
``padding = stride`` (which is exactly the reference's ``[stride:-stride]`` trim) on an input that
got one replicated sample on the left.
"""
import torch

from .. import functional as Fn
from .conv import Conv1d, ConvTranspose1d
from .padding import get_pad


class CausalConv1d(torch.nn.Module):
    """``conv(pad_left(x, (k-1)*d))[:, :, :T]`` (causal_conv.py:12-43) as one launch."""

    def __init__(self, in_channels, out_channels, kernel_size, dilation=1, bias=True, pad="ConstantPad1d",
                 pad_params={"value": 0.0}):
        super().__init__()
        p = (kernel_size - 1) * dilation
        self.pad = get_pad(pad, p, **pad_params)  # marker (no parameters); the kernel pads implicitly
        self.conv = Conv1d(in_channels, out_channels, kernel_size, dilation=dilation, bias=bias, padding=(p, 0),
                           pad_mode=self.pad.mode)

    def forward(self, x, **fused):
        """Accepts the fused-epilogue keywords of :class:`Conv1d` (pre_act, add1, post_act, ...)."""
        return self.conv(x, **fused)


class CausalConvTranspose1d(torch.nn.Module):
    """``deconv(replicate_pad_left(x, 1))[:, :, stride:-stride]`` (causal_conv.py:46-77)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, bias=True, pad="ReplicationPad1d",
                 pad_params={}):
        super().__init__()
        self.pad = get_pad(pad, 1, **pad_params)
        if self.pad.mode == "zero" and getattr(self.pad, "value", 0.0) != 0.0:
            raise NotImplementedError("CausalConvTranspose1d: constant padding with a non-zero value")
        # ConvTranspose1d(padding=stride) == full transposed convolution trimmed by `stride` on both sides
        self.deconv = ConvTranspose1d(in_channels, out_channels, kernel_size, stride, padding=stride, bias=bias)
        self.stride = stride

    def forward(self, x, **fused):
        # the left pad and an element-wise pre-activation commute, so the activation stays fused
        return self.deconv(Fn.pad1d(x, 1, 0, self.pad.mode), **fused)
