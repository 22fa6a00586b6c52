"""StyleMelGAN's TADE residual block (drop-in for parallel_wavegan.layers.tade_res_block).

Same constructor arguments and sub-module names (``norm`` / ``aux_conv.0`` / ``gated_conv.0`` /
``tade1`` / ``gated_conv1`` / ``tade2`` / ``gated_conv2``) as the reference
(/root/reference/parallel_wavegan/layers/tade_res_block.py:11-161), hence the same state-dict keys.
Instance norm, nearest upsampling, the TADE modulation and the softmax/sigmoid x tanh gate are HIP
kernels of libpwgkernels.so; the convolutions run on the MFMA convolution kernel.
"""
import torch

from .. import functional as Fn
from .conv import Conv1d


class _Marker(torch.nn.Module):
    """Parameter-free stage that is fused into a HIP kernel (keeps the reference's module tree)."""

    def __init__(self, what):
        super().__init__()
        self.what = what

    def extra_repr(self):
        return self.what

    def forward(self, x):  # pragma: no cover
        raise RuntimeError(f"{self.what} is fused into a HIP kernel and never called as a module")


class TADELayer(torch.nn.Module):
    """``y = gamma(c) * upsample(instance_norm(x)) + beta(c)`` with (gamma, beta) = gated_conv(aux_conv(upsample(c)))."""

    def __init__(self, in_channels=64, aux_channels=80, kernel_size=9, bias=True, upsample_factor=2,
                 upsample_mode="nearest"):
        super().__init__()
        if upsample_mode != "nearest":
            raise NotImplementedError("only nearest upsampling has a gfx950 kernel")
        self.norm = _Marker(f"InstanceNorm1d({in_channels})")
        self.norm_eps = 1e-5  # torch.nn.InstanceNorm1d default
        self.aux_conv = torch.nn.Sequential(
            Conv1d(aux_channels, in_channels, kernel_size, 1, bias=bias, padding=(kernel_size - 1) // 2))
        self.gated_conv = torch.nn.Sequential(
            Conv1d(in_channels, in_channels * 2, kernel_size, 1, bias=bias, padding=(kernel_size - 1) // 2))
        self.upsample = _Marker(f"Upsample(scale_factor={upsample_factor}, mode=nearest)")
        self.upsample_factor = int(upsample_factor)

    def forward(self, x, c):
        """x (B, C, T), c (B, aux, T) -> y (B, C, T * f), c' (B, C, T * f)."""
        xn = Fn.InstanceNormFn.apply(x, self.norm_eps)
        f = self.upsample_factor
        if f != 1:
            c = Fn.UpsampleNearestFn.apply(c, f)
        c = self.aux_conv[0](c)
        cg = self.gated_conv[0](c)
        return Fn.TadeModulateFn.apply(xn, cg, f), c


class TADEResBlock(torch.nn.Module):
    def __init__(self, in_channels=64, aux_channels=80, kernel_size=9, dilation=2, bias=True, upsample_factor=2,
                 upsample_mode="nearest", gated_function="softmax"):
        super().__init__()
        self.tade1 = TADELayer(in_channels=in_channels, aux_channels=aux_channels, kernel_size=kernel_size, bias=bias,
                               upsample_factor=1, upsample_mode=upsample_mode)
        self.gated_conv1 = Conv1d(in_channels, in_channels * 2, kernel_size, 1, bias=bias,
                                  padding=(kernel_size - 1) // 2)
        self.tade2 = TADELayer(in_channels=in_channels, aux_channels=in_channels, kernel_size=kernel_size, bias=bias,
                               upsample_factor=upsample_factor, upsample_mode=upsample_mode)
        self.gated_conv2 = Conv1d(in_channels, in_channels * 2, kernel_size, 1, bias=bias, dilation=dilation,
                                  padding=(kernel_size - 1) // 2 * dilation)
        self.upsample = _Marker(f"Upsample(scale_factor={upsample_factor}, mode=nearest)")
        self.upsample_factor = int(upsample_factor)
        if gated_function not in ("softmax", "sigmoid"):
            raise ValueError(f"{gated_function} is not supported.")
        self.use_softmax = gated_function == "softmax"

    def forward(self, x, c):
        """x (B, C, T), c (B, aux, T') -> (B, C, T * f), (B, C, T * f)."""
        residual = x
        x, c = self.tade1(x, c)
        x = Fn.SoftmaxGateFn.apply(self.gated_conv1(x), self.use_softmax)
        x, c = self.tade2(x, c)
        x = Fn.SoftmaxGateFn.apply(self.gated_conv2(x), self.use_softmax)
        # upsample(residual) + x in one launch
        return Fn.UpsampleNearestFn.apply(residual, self.upsample_factor, x), c
