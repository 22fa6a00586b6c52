"""Convolution modules whose arithmetic runs in libpwgkernels.so.

They replace ``torch.nn.Conv1d`` / ``torch.nn.ConvTranspose1d`` (+ the old-style
``torch.nn.utils.weight_norm`` hook) at the reference's call sites and keep the
same parameter names and shapes, so reference checkpoints load unchanged:
``weight``/``bias`` or, with weight norm, ``weight_g``/``weight_v``/``bias``
(e.g. /root/reference/parallel_wavegan/models/hifigan.py:221-231).

The packed kernel weight image ([group][tap][ci][m], see csrc/conv1d.hip) is
derived from the parameters on the device (weight-norm scale + pack in two tiny
kernels) and cached until a parameter changes.
"""
import math

import torch

from .. import ops


class _ConvNd(torch.nn.Module):
    transposed = False

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, output_padding=0, pad_mode="zero"):
        super().__init__()
        if in_channels % groups or out_channels % groups:
            raise ValueError("in_channels and out_channels must be divisible by groups")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = int(kernel_size), int(stride), int(padding)
        self.dilation, self.groups, self.output_padding = int(dilation), int(groups), int(output_padding)
        self.pad_mode = pad_mode
        if self.transposed:
            shape = (in_channels, out_channels // groups, self.kernel_size)
        else:
            shape = (out_channels, in_channels // groups, self.kernel_size)
        self.weight = torch.nn.Parameter(torch.empty(shape))
        if bias:
            self.bias = torch.nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self._cache_key = None
        self._cache_packed = None
        self.reset_parameters()

    # -- initialisation identical in distribution to torch.nn.Conv1d's default
    def reset_parameters(self):
        w = self._raw_weight_for_init()
        fan_in = w.shape[1] * w.shape[2]
        bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
        with torch.no_grad():
            w.uniform_(-bound, bound)  # kaiming_uniform_(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
            if self.bias is not None:
                self.bias.uniform_(-bound, bound)

    def _raw_weight_for_init(self):
        return self.weight if self.has_weight_norm is False else self.weight_v

    # -- old-style weight norm (dim=0): w = g * v / ||v||
    @property
    def has_weight_norm(self):
        return "weight_g" in self._parameters

    def apply_weight_norm(self):
        if self.has_weight_norm:
            return self
        w = self._parameters.pop("weight")
        with torch.no_grad():
            g = w.detach().reshape(w.shape[0], -1).norm(dim=1).reshape(-1, 1, 1)
        self.weight_g = torch.nn.Parameter(g.clone())
        self.weight_v = torch.nn.Parameter(w.detach().clone())
        self._cache_key = None
        return self

    def remove_weight_norm(self):
        if not self.has_weight_norm:
            raise ValueError(f"weight norm is not applied to {self.__class__.__name__}")
        g, v = self._parameters.pop("weight_g"), self._parameters.pop("weight_v")
        with torch.no_grad():
            if v.is_cuda:
                w = ops.scale_rows(v.detach().contiguous(), ops.weight_norm_scale(v.detach().contiguous(),
                                                                               g.detach().reshape(-1).contiguous()))
            else:  # host-side bookkeeping only (checkpoint conversion); not a compute path
                n = v.detach().reshape(v.shape[0], -1).norm(dim=1).reshape(-1, 1, 1)
                w = v.detach() * (g.detach() / n)
        self.weight = torch.nn.Parameter(w)
        self._cache_key = None
        return self

    def effective_weight(self):
        """torch-layout weight tensor on the device (materialised by HIP kernels)."""
        if not self.has_weight_norm:
            return self.weight
        v = self.weight_v.detach().contiguous()
        return ops.scale_rows(v, ops.weight_norm_scale(v, self.weight_g.detach().reshape(-1).contiguous()))

    # -- packed image
    def _geometry_desc(self, batch=1, t_in=None):
        t_in = t_in if t_in is not None else self.kernel_size * self.dilation + self.stride
        return self.make_desc(batch, t_in)

    def packed_weight(self):
        params = [self.weight_g, self.weight_v] if self.has_weight_norm else [self.weight]
        key = tuple((p.data_ptr(), p._version, str(p.device)) for p in params)
        if key != self._cache_key:
            desc = self._geometry_desc()
            with torch.no_grad():
                if self.has_weight_norm:
                    v = self.weight_v.detach().contiguous()
                    scale = ops.weight_norm_scale(v, self.weight_g.detach().reshape(-1).contiguous())
                    self._cache_packed = ops.pack_weight(desc, v, scale)
                else:
                    self._cache_packed = ops.pack_weight(desc, self.weight.detach().contiguous())
            self._cache_key = key
        return self._cache_packed

    def out_length(self, t_in):
        raise NotImplementedError

    def make_desc(self, batch, t_in, **fused):
        raise NotImplementedError

    def forward(self, x, pre_act=None, pre_slope=0.0, post_act=None, post_slope=0.0, add1=None, add2=None,
                out_mul=1.0, out_div=1.0, out=None):
        """Fused ``post((conv(pre(x)) + bias + add1 + add2) * out_mul / out_div)``."""
        b, _, t_in = x.shape
        desc = self.make_desc(b, t_in, pre_act=pre_act, pre_slope=pre_slope, post_act=post_act,
                              post_slope=post_slope, out_mul=out_mul, out_div=out_div)
        return ops.conv1d_forward(desc, x.contiguous(), self.packed_weight(),
                                  None if self.bias is None else self.bias.detach(), add1, add2, out)

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, "
                f"padding={self.padding}, dilation={self.dilation}, groups={self.groups}, "
                f"weight_norm={self.has_weight_norm}")


class Conv1d(_ConvNd):
    """Drop-in for ``torch.nn.Conv1d`` (zero / reflect / replicate implicit padding)."""

    transposed = False

    def out_length(self, t_in):
        return ops.conv_out_length(t_in, self.kernel_size, self.stride, self.dilation, self.padding, self.padding)

    def make_desc(self, batch, t_in, **fused):
        return ops.make_conv_desc(batch, self.in_channels, self.out_channels, t_in, self.out_length(t_in),
                                  self.kernel_size, self.stride, self.dilation, self.padding, self.groups,
                                  transposed=False, pad_mode=self.pad_mode, **fused)


class ConvTranspose1d(_ConvNd):
    """Drop-in for ``torch.nn.ConvTranspose1d`` (polyphase; weight (C_in, C_out/groups, k))."""

    transposed = True

    def out_length(self, t_in):
        return ops.conv_transpose_out_length(t_in, self.kernel_size, self.stride, self.padding, self.output_padding)

    def make_desc(self, batch, t_in, **fused):
        return ops.make_conv_desc(batch, self.in_channels, self.out_channels, t_in, self.out_length(t_in),
                                  self.kernel_size, self.stride, 1, self.padding, self.groups, transposed=True,
                                  **fused)
