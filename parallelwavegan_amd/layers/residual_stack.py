"""MelGAN residual stack (drop-in for parallel_wavegan.layers.residual_stack)."""
import os

import torch

from .. import functional as Fn
from .. import ops
from .activation import FusedActivation
from .causal_conv import CausalConv1d
from .conv import Conv1d
from .padding import get_pad


class ResidualStack(torch.nn.Module):
    """``[act, pad(d), conv k3 dil d, act, conv 1x1](c) + skip 1x1(c)`` (layers/residual_stack.py:13-85).
    MelGAN's geometries (48 / 96 / 192 channels, kernel 3, reflect padding, LeakyReLU) run as ONE launch
    (csrc/resstack.hip; ``fuse_unit``); everything else as three: skip conv; dilated conv with activation + reflect
    padding fused on its input; 1x1 conv with activation fused on its input and the skip branch fused as addend."""

    fuse_unit = os.environ.get("PWG_RESSTACK", "1") != "0"  # (0: the three-launch path, for A/B measurements)
    # backward of the unit: its data gradient as one launch (functional.ResStackFn) instead of the three layers' own
    fuse_unit_backward = os.environ.get("PWG_RESSTACK_BWD", "1") != "0"

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
        if (self.fuse_unit and not use_causal_conv and kernel_size == 3 and channels in (48, 96, 192) and dilation <= 27
                and padm.mode == "reflect" and self.stack[0].kind == "leaky_relu" and 0.0 < self.stack[0].slope < 1.0):
            # the geometries the one-launch unit covers (csrc/resstack.hip): the unit packs its own image of the three
            # weights, so the weight bank prepares only their row scales; a call the unit turns down at run time
            # (T % 4 != 0, a non-contiguous input) packs the layers' images lazily
            for cv in self.unit_convs():
                cv.bank_images = False

    def unit_convs(self):
        """(dilated convolution, 1x1 convolution, skip 1x1) of the non-causal form."""
        return self.stack[2], self.stack[4], self.skip_layer

    def unit_slope(self):
        return self.stack[0].slope

    def unit_params(self):
        """Parameters in the order functional.ResStackFn returns their gradients."""
        ps = []
        for cv in self.unit_convs():
            ps.append(cv.raw_weight)
            if cv.has_weight_norm:
                ps.append(cv.weight_g)
            if cv.bias is not None:
                ps.append(cv.bias)
        return ps

    def _image(self, attr, pack):
        convs = self.unit_convs()
        key = tuple(cv._params_key() for cv in convs)
        if getattr(self, attr + "_key", None) != key:
            hs = [cv.prepared() for cv in convs]
            with torch.no_grad():
                setattr(self, attr, pack(hs[0].w, hs[0].scale, hs[1].w, hs[1].scale, hs[2].w, hs[2].scale))
            setattr(self, attr + "_key", key)
        return getattr(self, attr)

    def unit_image(self):
        """MFMA A-operand image of the three weights for the current parameter values (cached)."""
        return self._image("_unit_img", ops.resstack_pack_weight)

    def unit_image_bwd(self):
        """The same for the unit's data gradient (transposed weights)."""
        return self._image("_unit_img_bwd", ops.resstack_pack_weight_bwd)

    def _unit_ok(self, c, a0, conv0, a1, conv1):
        if not self.fuse_unit or self.use_causal_conv or not c.is_cuda or c.dim() != 3 or c.dtype != torch.float32:
            return False
        skip = self.skip_layer
        if any(cv.has_spectral_norm or cv.groups != 1 or cv.stride != 1 for cv in (conv0, conv1, skip)):
            return False
        if (conv0.kernel_size != 3 or conv0.pad_mode != "reflect" or conv0.padding != conv0.dilation
                or conv0.padding_right != conv0.dilation or conv1.kernel_size != 1 or skip.kernel_size != 1):
            return False
        if a0.kind != "leaky_relu" or a1.kind != "leaky_relu" or a0.slope != a1.slope or not 0.0 < a0.slope < 1.0:
            return False
        ch = conv0.in_channels
        if any(cv.in_channels != ch or cv.out_channels != ch for cv in (conv0, conv1, skip)) or c.shape[1] != ch:
            return False
        if not c.is_contiguous():  # (rows are staged with 16-B LDS-DMA pieces; 4-B source alignment is enough)
            return False
        return ops.resstack_supported(ch, c.shape[2], conv0.dilation)

    def forward(self, c):
        if self.use_causal_conv:
            a0, conv0, a1, conv1 = self.stack[0], self.stack[1], self.stack[2], self.stack[3]
        else:
            a0, conv0, a1, conv1 = self.stack[0], self.stack[2], self.stack[3], self.stack[4]
        if self._unit_ok(c, a0, conv0, a1, conv1):
            convs = (conv0, conv1, self.skip_layer)
            bias = [None if cv.bias is None else cv.bias.detach() for cv in convs]
            needs_grad = torch.is_grad_enabled() and (c.requires_grad or any(p.requires_grad for p in self.parameters()))
            if needs_grad and self.fuse_unit_backward:
                return Fn.ResStackFn.apply(c, self, *self.unit_params())
            with torch.no_grad():
                y, h = ops.resstack_forward(c, self.unit_image(), conv0.dilation, a0.slope, *bias, save_h=needs_grad)
            if not needs_grad:
                return y
            # the three layers' autograd nodes, without their launches: the values come from the one-launch unit, the
            # backward pass is the layers' own (the skip branch's value is never read: only its gradient path matters)
            skip = self.skip_layer(c, precomputed=torch.empty_like(y))
            t = conv0(c, pre_act=a0.kind, pre_slope=a0.slope, precomputed=h)
            return conv1(t, pre_act=a1.kind, pre_slope=a1.slope, add1=skip, precomputed=y)
        skip = self.skip_layer(c)
        t = conv0(c, pre_act=a0.kind, pre_slope=a0.slope)
        return conv1(t, pre_act=a1.kind, pre_slope=a1.slope, add1=skip)
