"""MelGAN / Multi-band MelGAN modules on the gfx950 kernel library (drop-in for
``parallel_wavegan.models.melgan``; reference: models/melgan.py)."""
import logging

import numpy as np
import torch

from ..layers.activation import FusedActivation, PreActivated, deferrable
from ..layers.causal_conv import CausalConv1d, CausalConvTranspose1d
from ..layers.conv import Conv1d, ConvTranspose1d
from ..layers.padding import FusedPad, get_pad
from ..layers.pooling import get_pooling
from ..layers.residual_stack import ResidualStack


def _convs(module):
    return [m for m in module.modules() if isinstance(m, (Conv1d, ConvTranspose1d))]


from ..streams import run_branches  # noqa: E402

class _MelGANNormMixin:
    def remove_weight_norm(self):
        for m in _convs(self):
            if m.has_weight_norm:
                m.remove_weight_norm()

    def apply_weight_norm(self):
        for m in _convs(self):
            m.apply_weight_norm()

    def reset_parameters(self):
        """N(0, 0.02) as the official implementation; with weight norm already applied the
        reference's version writes the derived ``.weight`` and is a no-op."""
        for m in _convs(self):
            if not m.has_weight_norm:
                with torch.no_grad():
                    m.weight.normal_(0.0, 0.02)


class MelGANGenerator(torch.nn.Module, _MelGANNormMixin):
    """MelGAN generator (reference: models/melgan.py:17-257).  ``self.melgan`` keeps the flat
    Sequential of the reference (same indices => same state-dict keys); activations, paddings and
    the final tanh are fused into the neighbouring convolution kernels."""

    def __init__(self, in_channels=80, out_channels=1, kernel_size=7, channels=512, bias=True,
                 upsample_scales=[8, 8, 2, 2], stack_kernel_size=3, stacks=3, nonlinear_activation="LeakyReLU",
                 nonlinear_activation_params={"negative_slope": 0.2}, pad="ReflectionPad1d", pad_params={},
                 use_final_nonlinear_activation=True, use_weight_norm=True, use_causal_conv=False):
        super().__init__()
        assert channels >= np.prod(upsample_scales)
        assert channels % (2 ** len(upsample_scales)) == 0
        if not use_causal_conv:
            assert (kernel_size - 1) % 2 == 0, "Not support even number kernel size."
        p = (kernel_size - 1) // 2
        if use_causal_conv:  # models/melgan.py:75-84 of the reference: one module, pad inside
            layers = [CausalConv1d(in_channels, channels, kernel_size, bias=bias, pad=pad, pad_params=pad_params)]
        else:
            pad0 = get_pad(pad, p, **pad_params)
            layers = [pad0, Conv1d(in_channels, channels, kernel_size, bias=bias, padding=p, pad_mode=pad0.mode)]
        for i, s in enumerate(upsample_scales):
            layers.append(FusedActivation(nonlinear_activation, **nonlinear_activation_params))
            if use_causal_conv:
                layers.append(CausalConvTranspose1d(channels // (2 ** i), channels // (2 ** (i + 1)), s * 2, stride=s,
                                                    bias=bias))
            else:
                layers.append(ConvTranspose1d(channels // (2 ** i), channels // (2 ** (i + 1)), s * 2, stride=s,
                                              padding=s // 2 + s % 2, output_padding=s % 2, bias=bias))
            for j in range(stacks):
                layers.append(ResidualStack(kernel_size=stack_kernel_size, channels=channels // (2 ** (i + 1)),
                                            dilation=stack_kernel_size ** j, bias=bias,
                                            nonlinear_activation=nonlinear_activation,
                                            nonlinear_activation_params=nonlinear_activation_params, pad=pad,
                                            pad_params=pad_params, use_causal_conv=use_causal_conv))
        layers.append(FusedActivation(nonlinear_activation, **nonlinear_activation_params))
        if use_causal_conv:
            layers.append(CausalConv1d(channels // (2 ** (i + 1)), out_channels, kernel_size, bias=bias, pad=pad,
                                       pad_params=pad_params))
        else:
            pad1 = get_pad(pad, p, **pad_params)
            layers += [pad1, Conv1d(channels // (2 ** (i + 1)), out_channels, kernel_size, bias=bias, padding=p,
                                    pad_mode=pad1.mode)]
        self.use_final_nonlinear_activation = use_final_nonlinear_activation
        if use_final_nonlinear_activation:
            layers.append(torch.nn.Identity())  # the (fused) Tanh of the reference
        self.melgan = torch.nn.Sequential(*layers)
        self.upsample_factor = int(np.prod(upsample_scales))
        if use_weight_norm:
            self.apply_weight_norm()
        self.reset_parameters()
        self.pqmf = None

    def forward(self, c):
        """(B, in_channels, T) -> (B, out_channels, T * prod(upsample_scales))."""
        mods = list(self.melgan)
        x = c
        act = None
        last_conv = max(i for i, m in enumerate(mods) if isinstance(m, (Conv1d, CausalConv1d)))
        for i, m in enumerate(mods):
            if isinstance(m, FusedActivation):
                act = m
            elif isinstance(m, (FusedPad, torch.nn.Identity)):
                continue
            elif isinstance(m, ResidualStack):
                x = m(x)
            else:  # Conv1d / ConvTranspose1d: consume the pending activation
                kw = dict(pre_act=act.kind, pre_slope=act.slope) if act is not None else {}
                if i == last_conv and self.use_final_nonlinear_activation:
                    kw["post_act"] = "tanh"
                x = m(x, **kw)
                act = None
        return x

    def register_stats(self, stats):
        from ..utils import load_stats

        mean, scale = load_stats(stats)
        self.register_buffer("mean", torch.from_numpy(mean).float())
        self.register_buffer("scale", torch.from_numpy(scale).float())
        logging.info("Successfully registered stats as buffer.")

    def inference(self, c, normalize_before=False):
        """c: (T, in_channels) -> (T * prod(upsample_scales) [* subbands], out_channels or 1)."""
        if not isinstance(c, torch.Tensor):
            c = torch.tensor(c, dtype=torch.float).to(next(self.parameters()).device)
        if normalize_before:
            c = (c - self.mean) / self.scale
        c = self.forward(c.transpose(1, 0).unsqueeze(0).contiguous())
        if self.pqmf is not None:
            c = self.pqmf.synthesis(c)
        return c.squeeze(0).transpose(1, 0)


class MelGANDiscriminator(torch.nn.Module):
    """MelGAN discriminator (reference: models/melgan.py:260-396): reflect-pad conv k15, grouped strided
    convs (k = 10 s + 1, groups = in/4), conv k5, conv k3; returns all layer outputs."""

    def __init__(self, in_channels=1, out_channels=1, kernel_sizes=[5, 3], channels=16, max_downsample_channels=1024,
                 bias=True, downsample_scales=[4, 4, 4, 4], nonlinear_activation="LeakyReLU",
                 nonlinear_activation_params={"negative_slope": 0.2}, pad="ReflectionPad1d", pad_params={}):
        super().__init__()
        self.layers = torch.nn.ModuleList()
        assert len(kernel_sizes) == 2
        assert kernel_sizes[0] % 2 == 1
        assert kernel_sizes[1] % 2 == 1

        def act():
            return FusedActivation(nonlinear_activation, **nonlinear_activation_params)

        k0 = int(np.prod(kernel_sizes))
        pad0 = get_pad(pad, (k0 - 1) // 2, **pad_params)
        self.layers.append(torch.nn.Sequential(
            pad0, Conv1d(in_channels, channels, k0, bias=bias, padding=(k0 - 1) // 2, pad_mode=pad0.mode), act()))
        in_chs = channels
        for s in downsample_scales:
            out_chs = min(in_chs * s, max_downsample_channels)
            self.layers.append(torch.nn.Sequential(
                Conv1d(in_chs, out_chs, kernel_size=s * 10 + 1, stride=s, padding=s * 5, groups=in_chs // 4, bias=bias),
                act()))
            in_chs = out_chs
        out_chs = min(in_chs * 2, max_downsample_channels)
        self.layers.append(torch.nn.Sequential(
            Conv1d(in_chs, out_chs, kernel_sizes[0], padding=(kernel_sizes[0] - 1) // 2, bias=bias), act()))
        self.layers.append(Conv1d(out_chs, out_channels, kernel_sizes[1], padding=(kernel_sizes[1] - 1) // 2, bias=bias))
        self.reset_parameters()

    def forward(self, x):
        outs = []
        seqs = [f for f in self.layers if isinstance(f, torch.nn.Sequential)]
        acts = [[m for m in f if isinstance(m, FusedActivation)][0] for f in seqs]
        if self.deferred_activation and deferrable(acts) and len(seqs) == len(self.layers) - 1:
            # deferred form (layers.activation.PreActivated): every LeakyReLU rides on the NEXT convolution's operand load
            slope, pre = acts[0].slope, None
            for f in self.layers:
                conv = [m for m in f if isinstance(m, Conv1d)][0] if isinstance(f, torch.nn.Sequential) else f
                x = conv(x, pre_act=pre, pre_slope=slope)
                pre = "leaky_relu"
                outs.append(x)
            return PreActivated(outs, slope)
        for f in self.layers:
            if isinstance(f, torch.nn.Sequential):
                conv = [m for m in f if isinstance(m, Conv1d)][0]
                act = [m for m in f if isinstance(m, FusedActivation)][0]
                x = conv(x, post_act=act.kind, post_slope=act.slope)
            else:
                x = f(x)
            outs.append(x)
        return outs

    deferred_activation = False  # set by the trainer (layers.activation.set_deferred_activation)

    def reset_parameters(self):
        for m in _convs(self):
            if not m.has_weight_norm:
                with torch.no_grad():
                    m.weight.normal_(0.0, 0.02)


class MelGANMultiScaleDiscriminator(torch.nn.Module, _MelGANNormMixin):
    """MelGAN multi-scale discriminator (reference: models/melgan.py:399-534)."""

    def __init__(self, in_channels=1, out_channels=1, scales=3, downsample_pooling="AvgPool1d",
                 downsample_pooling_params={"kernel_size": 4, "stride": 2, "padding": 1, "count_include_pad": False},
                 kernel_sizes=[5, 3], channels=16, max_downsample_channels=1024, bias=True,
                 downsample_scales=[4, 4, 4, 4], nonlinear_activation="LeakyReLU",
                 nonlinear_activation_params={"negative_slope": 0.2}, pad="ReflectionPad1d", pad_params={},
                 use_weight_norm=True):
        super().__init__()
        self.discriminators = torch.nn.ModuleList([
            MelGANDiscriminator(in_channels=in_channels, out_channels=out_channels, kernel_sizes=kernel_sizes,
                                channels=channels, max_downsample_channels=max_downsample_channels, bias=bias,
                                downsample_scales=downsample_scales, nonlinear_activation=nonlinear_activation,
                                nonlinear_activation_params=nonlinear_activation_params, pad=pad,
                                pad_params=pad_params)
            for _ in range(scales)])
        self.pooling = get_pooling(downsample_pooling, **downsample_pooling_params)
        if use_weight_norm:
            self.apply_weight_norm()
        self.reset_parameters()

    branch_streams = False  # set True by the trainer's hipGraph mode: the scales become parallel branches of the graph
    # three branches do not fill the chip the way HiFi-GAN's eight do: the planners keep assuming a launch has it to itself
    # (measured on the captured C4 step, same box, two rounds each: serial 27.93 ms, branches with hint 0.5 27.85, with
    # hint 1.0 27.56)
    branch_concurrency_hint = 1.0

    def forward(self, x):
        if self.branch_streams:
            # the pooled inputs first (cheap, sequential; the trailing pooling of the reference's loop, whose result is
            # never used, is dropped), then the scale discriminators as independent branches (streams.run_branches
            # forks only while the stream is being captured)
            xs = []
            for i in range(len(self.discriminators)):
                xs.append(x)
                if i + 1 < len(self.discriminators):
                    x = self.pooling(x)
            return run_branches([(lambda f=f, xi=xi: f(xi)) for f, xi in zip(self.discriminators, xs)], xs[0].device, True,
                                inputs=xs)
        outs = []
        for i, f in enumerate(self.discriminators):
            outs.append(f(x))
            if i + 1 < len(self.discriminators):  # (the reference pools once more; that result is never used)
                x = self.pooling(x)
        return outs
