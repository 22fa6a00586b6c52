"""Residual blocks (drop-in names for parallel_wavegan.layers.residual_block)."""
import torch

from .activation import FusedActivation
from .conv import Conv1d


class HiFiGANResidualBlock(torch.nn.Module):
    """MRF residual block: ``x = x + conv_{k,1}(act(conv_{k,d}(act(x))))`` per dilation.

    Same constructor / state-dict keys as the reference's
    ``HiFiGANResidualBlock`` (/root/reference/parallel_wavegan/layers/residual_block.py:143-258);
    the forward issues two fused HIP launches per dilation (activation, bias and
    the residual add live inside the convolution kernels).
    """

    def __init__(self, kernel_size=3, channels=512, dilations=(1, 3, 5), bias=True, use_additional_convs=True,
                 nonlinear_activation="LeakyReLU", nonlinear_activation_params={"negative_slope": 0.1},
                 use_causal_conv=False):
        super().__init__()
        assert kernel_size % 2 == 1, "Kernel size must be odd number."
        if use_causal_conv:
            raise NotImplementedError("use_causal_conv=True is outside the accelerated path (SURVEY.md s8f-3)")
        self.use_additional_convs = use_additional_convs
        self.use_causal_conv = use_causal_conv
        self.kernel_size = kernel_size
        self.dilations = tuple(dilations)
        self.convs1 = torch.nn.ModuleList()
        if use_additional_convs:
            self.convs2 = torch.nn.ModuleList()
        for d in self.dilations:
            self.convs1.append(torch.nn.Sequential(
                FusedActivation(nonlinear_activation, **nonlinear_activation_params),
                Conv1d(channels, channels, kernel_size, 1, dilation=d, bias=bias, padding=(kernel_size - 1) // 2 * d),
            ))
            if use_additional_convs:
                self.convs2.append(torch.nn.Sequential(
                    FusedActivation(nonlinear_activation, **nonlinear_activation_params),
                    Conv1d(channels, channels, kernel_size, 1, dilation=1, bias=bias, padding=(kernel_size - 1) // 2),
                ))

    def forward(self, x, accum=None, out_div=1.0):
        """Returns ``(block(x) + accum) / out_div``; accum/out_div let the caller fold the
        MRF sum ``cs += block(c); c = cs / num_blocks`` (models/hifigan.py:186-190 in the
        reference) into this block's last kernel."""
        n = len(self.convs1)
        for idx in range(n):
            last = idx == n - 1
            act1, conv1 = self.convs1[idx][0], self.convs1[idx][1]
            if self.use_additional_convs:
                xt = conv1(x, pre_act=act1.kind, pre_slope=act1.slope)
                act2, conv2 = self.convs2[idx][0], self.convs2[idx][1]
                x = conv2(xt, pre_act=act2.kind, pre_slope=act2.slope, add1=x,
                          add2=accum if last else None, out_div=out_div if last else 1.0)
            else:
                x = conv1(x, pre_act=act1.kind, pre_slope=act1.slope, add1=x,
                          add2=accum if last else None, out_div=out_div if last else 1.0)
        return x
