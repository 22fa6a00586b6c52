"""MelGAN residual stack (drop-in for parallel_wavegan.layers.residual_stack)."""
import torch

from .activation import FusedActivation
from .causal_conv import CausalConv1d
from .conv import Conv1d
from .padding import get_pad


class ResidualStack(torch.nn.Module):
    """``[act, pad(d), conv k3 dil d, act, conv 1x1](c) + skip 1x1(c)`` (layers/residual_stack.py:13-85)
    as three launches: skip conv; dilated conv with activation + reflect padding fused on its input;
    1x1 conv with activation fused on its input and the skip branch fused as addend."""

    def __init__(self, kernel_size=3, channels=32, dilation=1, bias=True, nonlinear_activation="LeakyReLU",
                 nonlinear_activation_params={"negative_slope": 0.2}, pad="ReflectionPad1d", pad_params={},
                 use_causal_conv=False):
        super().__init__()
        self.use_causal_conv = use_causal_conv
        if use_causal_conv:
            # layers/residual_stack.py:56-69 of the reference: no separate pad module, hence 4 entries
            self.stack = torch.nn.Sequential(
                FusedActivation(nonlinear_activation, **nonlinear_activation_params),
                CausalConv1d(channels, channels, kernel_size, dilation=dilation, bias=bias, pad=pad,
                             pad_params=pad_params),
                FusedActivation(nonlinear_activation, **nonlinear_activation_params),
                Conv1d(channels, channels, 1, bias=bias),
            )
        else:
            assert (kernel_size - 1) % 2 == 0, "Not support even number kernel size."
            p = (kernel_size - 1) // 2 * dilation
            padm = get_pad(pad, p, **pad_params)
            self.stack = torch.nn.Sequential(
                FusedActivation(nonlinear_activation, **nonlinear_activation_params),
                padm,
                Conv1d(channels, channels, kernel_size, dilation=dilation, bias=bias, padding=p, pad_mode=padm.mode),
                FusedActivation(nonlinear_activation, **nonlinear_activation_params),
                Conv1d(channels, channels, 1, bias=bias),
            )
        self.skip_layer = Conv1d(channels, channels, 1, bias=bias)

    def forward(self, c):
        if self.use_causal_conv:
            a0, conv0, a1, conv1 = self.stack[0], self.stack[1], self.stack[2], self.stack[3]
        else:
            a0, conv0, a1, conv1 = self.stack[0], self.stack[2], self.stack[3], self.stack[4]
        skip = self.skip_layer(c)
        t = conv0(c, pre_act=a0.kind, pre_slope=a0.slope)
        return conv1(t, pre_act=a1.kind, pre_slope=a1.slope, add1=skip)
