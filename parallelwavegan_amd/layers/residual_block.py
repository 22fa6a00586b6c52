"""Residual blocks (drop-in names for parallel_wavegan.layers.residual_block)."""
import math

import torch

from .. import functional as Fn
from .. import ops
from .activation import FusedActivation
from .conv import Conv1d as _Conv1d


from .causal_conv import CausalConv1d  # noqa: E402  (depends on .conv only)
from .dropout import Dropout as _Dropout  # noqa: E402


class Conv1d(_Conv1d):
    """Conv1d with the reference's customised initialisation (kaiming normal for ReLU, zero bias;
    layers/residual_block.py:19-30)."""

    def reset_parameters(self):
        w = self.raw_weight
        fan_in = int(w[0].numel())
        with torch.no_grad():
            w.normal_(0.0, math.sqrt(2.0 / fan_in))  # kaiming_normal_(nonlinearity="relu"), fan_in mode
            if self.bias is not None:
                self.bias.zero_()


class Conv1d1x1(Conv1d):
    def __init__(self, in_channels, out_channels, bias):
        super().__init__(in_channels, out_channels, kernel_size=1, padding=0, dilation=1, bias=bias)


class WaveNetResidualBlock(torch.nn.Module):
    """Gated residual block of the PWG generator (layers/residual_block.py:43-140): dilated conv ->
    + aux 1x1 -> tanh * sigmoid -> skip 1x1 and out 1x1 (+ residual) * sqrt(0.5).  Five launches (+1 for the dropout mask when ``dropout > 0`` in training):
    aux conv, dilated conv (+aux fused as addend), gate, skip conv (+running skip sum fused), out conv
    (+residual and the sqrt(0.5) scale fused)."""

    def __init__(self, kernel_size=3, residual_channels=64, gate_channels=128, skip_channels=64, aux_channels=80,
                 dropout=0.0, dilation=1, bias=True, use_causal_conv=False):
        super().__init__()
        self.dropout = dropout
        self._drop = _Dropout(dropout)  # parameter-free: no state-dict entries
        self.use_causal_conv = use_causal_conv
        if use_causal_conv:
            # the reference pads (k-1)*d on both sides and drops the future part of the output
            # (layers/residual_block.py:74-76,118-119): identical to left-only padding
            padding = ((kernel_size - 1) * dilation, 0)
        else:
            assert (kernel_size - 1) % 2 == 0, "Not support even number kernel size."
            padding = (kernel_size - 1) // 2 * dilation
        self.conv = Conv1d(residual_channels, gate_channels, kernel_size, padding=padding, dilation=dilation,
                           bias=bias)
        self.conv1x1_aux = Conv1d1x1(aux_channels, gate_channels, bias=False) if aux_channels > 0 else None
        gate_out_channels = gate_channels // 2
        self.conv1x1_out = Conv1d1x1(gate_out_channels, residual_channels, bias=bias)
        self.conv1x1_skip = Conv1d1x1(gate_out_channels, skip_channels, bias=bias)
        for cv in self.fused_convs():
            if cv is not None:
                cv.bank_images = False  # weight_bank.WeightBank: row scales only; the layer's own fused image is used

    fuse_layer = True  # one launch per layer where csrc/wavenet.hip covers the geometry (PWG.v1: 64 / 128 / 64 / 80)

    # ---- the one-launch layer (csrc/wavenet.hip)
    def fused_convs(self):
        return (self.conv, self.conv1x1_aux, self.conv1x1_skip, self.conv1x1_out)

    def fused_params(self):
        """Parameters in the order WaveNetLayerFn returns their gradients."""
        ps = []
        for cv in self.fused_convs():
            ps.append(cv.raw_weight)
            if cv.has_weight_norm:
                ps.append(cv.weight_g)
            if cv.bias is not None:
                ps.append(cv.bias)
        return ps

    def fused_desc(self, batch, t, skip_scale=1.0):
        return ops.make_wavenet_desc(batch, t, self.conv.dilation, self.conv.in_channels, self.conv.out_channels,
                                     self.conv1x1_skip.out_channels,
                                     0 if self.conv1x1_aux is None else self.conv1x1_aux.in_channels,
                                     self.conv.kernel_size, self.use_causal_conv, math.sqrt(0.5), skip_scale)

    def fused_image(self):
        """MFMA A-operand image of the layer's four weights for the current parameter values."""
        convs = self.fused_convs()
        key = tuple(cv._params_key() for cv in convs)
        if getattr(self, "_fused_key", None) != key:
            hs = [cv.prepared() for cv in convs]
            with torch.no_grad():
                self._fused_img = ops.wavenet_pack_weights(self.fused_desc(1, 64), hs[0].w, hs[0].scale, hs[1].w,
                                                           hs[1].scale, hs[2].w, hs[2].scale, hs[3].w, hs[3].scale)
            self._fused_key = key
        return self._fused_img

    def fused_image_bwd(self, skip_scale=1.0):
        """Backward-pass image (gate / dilated / aux data gradients) for the current parameter values."""
        convs = self.fused_convs()
        key = (float(skip_scale),) + tuple(cv._params_key() for cv in convs)
        if getattr(self, "_fused_bwd_key", None) != key:
            hs = [cv.prepared() for cv in convs]
            with torch.no_grad():
                self._fused_bwd_img = ops.wavenet_pack_weights_bwd(self.fused_desc(1, 64, skip_scale), hs[0].w, hs[0].scale,
                                                                   hs[1].w, hs[1].scale, hs[2].w, hs[2].scale, hs[3].w,
                                                                   hs[3].scale)
            self._fused_bwd_key = key
        return self._fused_bwd_img

    def _fusable(self, x, c):
        if not self.fuse_layer or c is None or self.conv1x1_aux is None or x.dim() != 3 or not x.is_cuda:
            return False
        if self.use_causal_conv or (self.dropout > 0.0 and self.training):
            return False
        if any(cv.has_spectral_norm or cv.pad_mode != "zero" for cv in self.fused_convs()):
            return False
        if self.conv1x1_out.out_channels != self.conv.in_channels:
            return False
        return ops.wavenet_layer_supported(self.fused_desc(x.shape[0], x.shape[2]))

    def forward(self, x, c, skips=None, skip_scale=1.0, chain_aux=False, inplace_skips=False):
        """Returns (x_out, skips + s) -- the running skip sum is an addend of the skip conv's epilogue
        (``skip_scale`` is the final ``sqrt(1/layers)`` of the generator, applied by the last block).
        ``inplace_skips``: the no-grad fused path may write the new running sum into ``skips`` itself.
        ``chain_aux``: also return the aux features for the NEXT layer (the same values; on the one-launch autograd
        path an alias whose gradient is chained through the layers' data-gradient epilogues)."""
        if self._fusable(x, c):
            needs_grad = torch.is_grad_enabled() and (x.requires_grad or c.requires_grad
                                                      or (skips is not None and skips.requires_grad)
                                                      or any(p.requires_grad for p in self.fused_params()))
            if needs_grad:
                x_out, s_out, c_next = Fn.WaveNetLayerFn.apply(x, c, skips, self, skip_scale, *self.fused_params())
                return (x_out, s_out, c_next) if chain_aux else (x_out, s_out)
            with torch.no_grad():
                convs = self.fused_convs()
                b_d, b_s, b_o = (None if cv.bias is None else cv.bias.detach() for cv in (convs[0], convs[2], convs[3]))
                x_out, s_out, _, _ = ops.wavenet_layer_forward(self.fused_desc(x.shape[0], x.shape[2], skip_scale),
                                                               x.contiguous(), c.contiguous(), skips, self.fused_image(),
                                                               b_d, b_s, b_o,
                                                               # the running skip sum is updated in place only when the caller
                                                               # owns that buffer and says so (the generator's own loop)
                                                               skips_out=skips if inplace_skips else None)
                return (x_out, s_out, c) if chain_aux else (x_out, s_out)
        aux = self.conv1x1_aux(c) if (c is not None and self.conv1x1_aux is not None) else None
        # F.dropout on the dilated conv's input only; the residual path keeps x (residual_block.py:114-116)
        z = self.conv(self._drop(x), add1=aux)
        g = Fn.GateFn.apply(z)
        s = self.conv1x1_skip(g, add1=skips, out_mul=skip_scale)
        x = self.conv1x1_out(g, add1=x, out_mul=math.sqrt(0.5))
        return (x, s, c) if chain_aux else (x, s)


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
        self.use_additional_convs = use_additional_convs
        self.use_causal_conv = use_causal_conv
        self.kernel_size = kernel_size
        self.dilations = tuple(dilations)
        self.convs1 = torch.nn.ModuleList()
        if use_additional_convs:
            self.convs2 = torch.nn.ModuleList()
        def conv(dilation):
            # causal: left-only padding inside CausalConv1d (layers/residual_block.py:196-241 of the reference)
            if use_causal_conv:
                return CausalConv1d(channels, channels, kernel_size, dilation=dilation, bias=bias)
            return Conv1d(channels, channels, kernel_size, 1, dilation=dilation, bias=bias,
                          padding=(kernel_size - 1) // 2 * dilation)

        for d in self.dilations:
            self.convs1.append(torch.nn.Sequential(
                FusedActivation(nonlinear_activation, **nonlinear_activation_params), conv(d)))
            if use_additional_convs:
                self.convs2.append(torch.nn.Sequential(
                    FusedActivation(nonlinear_activation, **nonlinear_activation_params), conv(1)))

    fuse_units = True  # inference: one launch per unit where csrc/resunit.hip covers the geometry (C = 32 / 64)

    def _unit_one_launch(self, idx, x, accum, out_div):
        """``(x + convs2[idx](convs1[idx](x)) [+ accum]) [/ out_div]`` as ONE kernel launch, or None when the
        unit has to run as separate convolutions (training: the intermediate is needed by the backward pass;
        causal / wide layers: no resident-tile kernel)."""
        act1, conv1 = self.convs1[idx][0], self.convs1[idx][1]
        if (not self.fuse_units or self.use_causal_conv or act1.kind != "leaky_relu" or x.dim() != 3
                or not x.is_cuda or conv1._needs_grad(x, None if callable(accum) else accum)):
            return None
        slope2, conv2 = act1.slope, None
        if self.use_additional_convs:
            act2, conv2 = self.convs2[idx][0], self.convs2[idx][1]
            if act2.kind != "leaky_relu" or conv2._needs_grad(x) or conv2.has_spectral_norm:
                return None
            slope2 = act2.slope
        if conv1.has_spectral_norm or conv1.pad_mode != "zero":
            return None
        desc = ops.make_resunit_desc(x.shape[0], x.shape[1], x.shape[2], self.kernel_size, conv1.dilation,
                                     conv2 is not None, act1.slope, slope2, out_div)
        if not ops.resunit_profitable(desc):
            return None
        if callable(accum):
            accum = accum()  # (resolved as late as possible: it may wait for another stream)
        with torch.no_grad():
            x = x.contiguous()
            return ops.resunit_forward(
                desc, x, conv1.prepared().res(), None if conv1.bias is None else conv1.bias.detach(),
                None if conv2 is None else conv2.prepared().res(),
                None if (conv2 is None or conv2.bias is None) else conv2.bias.detach(),
                None if accum is None else accum.contiguous())

    def forward(self, x, accum=None, out_div=1.0, accum_join=None):
        """``accum_join``: optional callable evaluated right before the block's last kernel, returning ``accum``
        (streams.run_branches_chained: the previous MRF branch is only waited for there).
        Returns ``(block(x) + accum) / out_div``; accum/out_div let the caller fold the
        MRF sum ``cs += block(c); c = cs / num_blocks`` (models/hifigan.py:186-190 in the
        reference) into this block's last kernel."""
        n = len(self.convs1)

        def acc():
            """the addend of the last kernel, resolved right before that kernel is launched"""
            nonlocal accum, accum_join
            if accum_join is not None:
                accum, accum_join = accum_join(), None
            return accum

        for idx in range(n):
            last = idx == n - 1
            act1, conv1 = self.convs1[idx][0], self.convs1[idx][1]
            # (a pending join is passed on as a callable and resolved right before the launch)
            y = self._unit_one_launch(idx, x, (acc if accum_join is not None else accum) if last else None,
                                      out_div if last else 1.0)
            if y is not None:
                x = y
                continue
            if self.use_additional_convs:
                act2, conv2 = self.convs2[idx][0], self.convs2[idx][1]
                if not self.use_causal_conv and not (conv1._needs_grad(x) or conv2._needs_grad(x, accum)):
                    # inference: the intermediate has one consumer, so its activation is applied ONCE by the
                    # producer's epilogue instead of on every operand read of the consumer (k reads per
                    # element, 2 VALU ops each inside the MFMA loop); same fp32 values either way
                    xt = conv1(x, pre_act=act1.kind, pre_slope=act1.slope, post_act=act2.kind, post_slope=act2.slope)
                    x = conv2(xt, add1=x, add2=acc() if last else None, out_div=out_div if last else 1.0)
                    continue
                xt = conv1(x, pre_act=act1.kind, pre_slope=act1.slope)
                x = conv2(xt, pre_act=act2.kind, pre_slope=act2.slope, add1=x,
                          add2=acc() if last else None, out_div=out_div if last else 1.0)
            else:
                x = conv1(x, pre_act=act1.kind, pre_slope=act1.slope, add1=x,
                          add2=acc() if last else None, out_div=out_div if last else 1.0)
        return x
