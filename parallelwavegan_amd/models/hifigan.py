"""HiFi-GAN modules on the gfx950 kernel library.

Drop-in for ``parallel_wavegan.models.hifigan`` (constructor kwargs, method
names and state-dict keys follow /root/reference/parallel_wavegan/models/hifigan.py);
the arithmetic is hand-written HIP behind ``parallelwavegan_amd.ops``.
"""
import copy
import logging

import numpy as np
import torch

from .. import functional as Fn
from ..layers.activation import FusedActivation, PreActivated, deferrable
from ..layers.causal_conv import CausalConv1d, CausalConvTranspose1d
from ..layers.conv import Conv1d, Conv2d, ConvTranspose1d
from ..layers.pooling import get_pooling
from ..streams import fork_now, run_branches, run_branches_chained
from ..layers.residual_block import HiFiGANResidualBlock as ResidualBlock


def _each_conv(module):
    for m in module.modules():
        if isinstance(m, (Conv1d, ConvTranspose1d)):
            yield m


class HiFiGANGenerator(torch.nn.Module):
    """HiFi-GAN generator (reference: models/hifigan.py:23-267).

    Forward = 1 input conv, then per upsampling stage one polyphase
    ConvTranspose1d launch (LeakyReLU fused on its input) and
    ``2 * len(dilations)`` fused conv launches per MRF block; the MRF sum and
    ``/ num_blocks`` are folded into the last launch of each block, the final
    LeakyReLU(0.01) + conv + tanh is one launch.
    """

    branch_streams = False  # True: run the MRF blocks of a stage on separate streams (graph branches)
    # chained branch ends (no combine launch) from this stage size on: B16 x 800 frames 72.6 -> 71.9 ms per forward;
    # single utterances are latency-bound and lose 4 % to the extra dependency (B1 x 800 frames 5.54 -> 5.75 ms)
    chain_min_elems = 1 << 24

    def __init__(self, in_channels=80, out_channels=1, channels=512, kernel_size=7, upsample_scales=(8, 8, 2, 2),
                 upsample_kernel_sizes=(16, 16, 4, 4), resblock_kernel_sizes=(3, 7, 11),
                 resblock_dilations=[(1, 3, 5), (1, 3, 5), (1, 3, 5)], use_additional_convs=True, bias=True,
                 nonlinear_activation="LeakyReLU", nonlinear_activation_params={"negative_slope": 0.1},
                 use_causal_conv=False, use_weight_norm=True):
        super().__init__()
        assert kernel_size % 2 == 1, "Kernel size must be odd number."
        assert len(upsample_scales) == len(upsample_kernel_sizes)
        assert len(resblock_dilations) == len(resblock_kernel_sizes)
        self.num_upsamples = len(upsample_kernel_sizes)
        self.num_blocks = len(resblock_kernel_sizes)
        self.use_causal_conv = use_causal_conv
        self.upsample_factor = int(np.prod(upsample_scales))
        if use_causal_conv:  # models/hifigan.py:82-88,115-123,156-162 of the reference
            self.input_conv = CausalConv1d(in_channels, channels, kernel_size, bias=bias)
        else:
            self.input_conv = Conv1d(in_channels, channels, kernel_size, bias=bias, padding=(kernel_size - 1) // 2)
        self.upsamples = torch.nn.ModuleList()
        self.blocks = torch.nn.ModuleList()
        ch = channels
        for i, (s, k) in enumerate(zip(upsample_scales, upsample_kernel_sizes)):
            assert k == 2 * s
            self.upsamples.append(torch.nn.Sequential(
                FusedActivation(nonlinear_activation, **nonlinear_activation_params),
                CausalConvTranspose1d(ch, ch // 2, k, s, bias=bias) if use_causal_conv else
                ConvTranspose1d(ch, ch // 2, k, s, padding=s // 2 + s % 2, output_padding=s % 2, bias=bias),
            ))
            ch //= 2
            for ks, dil in zip(resblock_kernel_sizes, resblock_dilations):
                self.blocks.append(ResidualBlock(
                    kernel_size=ks, channels=ch, dilations=dil, bias=bias,
                    use_additional_convs=use_additional_convs, nonlinear_activation=nonlinear_activation,
                    nonlinear_activation_params=nonlinear_activation_params, use_causal_conv=use_causal_conv))
        # the reference uses torch.nn.LeakyReLU() here, i.e. the DEFAULT slope 0.01
        # (models/hifigan.py:139-151), not nonlinear_activation_params
        self.output_conv = torch.nn.Sequential(
            FusedActivation("LeakyReLU"),
            CausalConv1d(ch, out_channels, kernel_size, bias=bias) if use_causal_conv else
            Conv1d(ch, out_channels, kernel_size, bias=bias, padding=(kernel_size - 1) // 2),
            torch.nn.Identity(),  # index 2 is the (fused) Tanh in the reference
        )
        if use_weight_norm:
            self.apply_weight_norm()
        self.reset_parameters()

    def forward(self, c):
        """c: (B, in_channels, T) -> (B, out_channels, T * prod(upsample_scales))."""
        c = self.input_conv(c)
        nb = self.num_blocks
        for i in range(self.num_upsamples):
            act, up = self.upsamples[i][0], self.upsamples[i][1]
            c = up(c, pre_act=act.kind, pre_slope=act.slope)
            fork = self.branch_streams and c.is_cuda and fork_now()  # (side streams only while capturing: streams.py)
            if fork and nb >= 2 and c.numel() >= self.chain_min_elems and not c.requires_grad and not (
                    torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())):
                # inference: the MRF blocks as parallel branches whose LAST kernels are chained -- branch j's last
                # kernel waits for branch j-1 and adds its result in the epilogue (the reference's running sum
                # `cs += block(c); c = cs / num_blocks`, models/hifigan.py:186-190, same order, no combine launch)
                c = run_branches_chained(
                    [(lambda join, j=j, c=c: self.blocks[i * nb + j](c, accum_join=join,
                                                                     out_div=float(nb) if j == nb - 1 else 1.0))
                     for j in range(nb)], c.device, inputs=c)
                continue
            if fork and 2 <= nb <= 3:
                # training: the MRF blocks as parallel branches; combined by one small kernel in the same order
                # ((b0 + b1) + b2) / nb as the reference's running sum
                outs = run_branches([(lambda j=j, c=c: self.blocks[i * nb + j](c)) for j in range(nb)], c.device, True,
                                    inputs=c)
                c = Fn.Add3DivFn.apply(outs[0], outs[1], outs[2] if nb == 3 else None, float(nb))
                continue
            cs = None
            for j in range(nb):
                last = j == nb - 1
                cs = self.blocks[i * nb + j](c, accum=cs, out_div=float(nb) if last else 1.0)
            c = cs
        act, conv = self.output_conv[0], self.output_conv[1]
        return conv(c, pre_act=act.kind, pre_slope=act.slope, post_act="tanh")

    def reset_parameters(self):
        """N(0, 0.01) on conv weights as in the official implementation.

        With weight norm already applied the reference's version of this writes the
        *derived* ``.weight`` attribute and is therefore a no-op on the trainable
        ``weight_g``/``weight_v`` (SURVEY.md App. A "Init quirk"); reproduced here by
        only touching plain ``weight`` parameters.
        """
        for m in _each_conv(self):
            if not m.has_weight_norm:
                with torch.no_grad():
                    m.weight.normal_(0.0, 0.01)
                logging.debug(f"Reset parameters in {m}.")

    def remove_weight_norm(self):
        for m in _each_conv(self):
            if m.has_weight_norm:
                m.remove_weight_norm()
                logging.debug(f"Weight norm is removed from {m}.")

    def apply_weight_norm(self):
        for m in _each_conv(self):
            m.apply_weight_norm()
            logging.debug(f"Weight norm is applied to {m}.")

    def register_stats(self, stats):
        """Register de-normalisation stats (``.npy`` [mean; scale] or ``.h5``)."""
        assert stats.endswith(".h5") or stats.endswith(".npy")
        if stats.endswith(".h5"):
            from ..utils import read_hdf5

            mean = read_hdf5(stats, "mean").reshape(-1)
            scale = read_hdf5(stats, "scale").reshape(-1)
        else:
            mean = np.load(stats)[0].reshape(-1)
            scale = np.load(stats)[1].reshape(-1)
        self.register_buffer("mean", torch.from_numpy(mean).float())
        self.register_buffer("scale", torch.from_numpy(scale).float())
        logging.info("Successfully registered stats as buffer.")

    def inference(self, c, normalize_before=False):
        """c: (T, in_channels) tensor/ndarray -> (T * prod(upsample_scales), out_channels)."""
        if not isinstance(c, torch.Tensor):
            c = torch.tensor(c, dtype=torch.float).to(next(self.parameters()).device)
        if normalize_before:
            c = (c - self.mean) / self.scale
        c = self.forward(c.transpose(1, 0).unsqueeze(0).contiguous())
        return c.squeeze(0).transpose(1, 0)


class HiFiGANPeriodDiscriminator(torch.nn.Module):
    """HiFi-GAN period discriminator (reference: models/hifigan.py:270-402).

    The waveform is reflect-padded to a multiple of ``period`` and viewed as (B, 1, T/p, p);
    every layer is a ``(k, 1)`` Conv2d, executed by the 1-D MFMA kernel with the period as the
    row width (no transposition, no copy), LeakyReLU fused into the producing kernel's epilogue.
    Returns the list of the 5 activated feature maps plus the flattened logits.
    """

    def __init__(self, in_channels=1, out_channels=1, period=3, kernel_sizes=[5, 3], channels=32,
                 downsample_scales=[3, 3, 3, 3, 1], max_downsample_channels=1024, bias=True,
                 nonlinear_activation="LeakyReLU", nonlinear_activation_params={"negative_slope": 0.1},
                 use_weight_norm=True, use_spectral_norm=False):
        super().__init__()
        assert len(kernel_sizes) == 2
        assert kernel_sizes[0] % 2 == 1, "Kernel size must be odd number."
        assert kernel_sizes[1] % 2 == 1, "Kernel size must be odd number."
        self.period = period
        self.convs = torch.nn.ModuleList()
        in_chs, out_chs = in_channels, channels
        for scale in downsample_scales:
            self.convs.append(torch.nn.Sequential(
                Conv2d(in_chs, out_chs, (kernel_sizes[0], 1), (scale, 1), padding=((kernel_sizes[0] - 1) // 2, 0),
                       bias=bias),
                FusedActivation(nonlinear_activation, **nonlinear_activation_params),
            ))
            in_chs = out_chs
            out_chs = min(out_chs * 4, max_downsample_channels)
        self.output_conv = Conv2d(out_chs, out_channels, (kernel_sizes[1] - 1, 1), (1, 1),
                                  padding=((kernel_sizes[1] - 1) // 2, 0), bias=bias)
        if use_weight_norm and use_spectral_norm:
            raise ValueError("Either use use_weight_norm or use_spectral_norm.")
        if use_weight_norm:
            self.apply_weight_norm()
        if use_spectral_norm:
            self.apply_spectral_norm()

    def forward(self, x):
        """x: (B, in_channels, T) -> list of tensors (5 feature maps (B,C,H,p) + logits (B, H'*p))."""
        b, c, t = x.shape
        if t % self.period != 0:
            n_pad = self.period - (t % self.period)
            x = Fn.pad1d(x, 0, n_pad, "reflect")
            t += n_pad
        x = x.reshape(b, c, t // self.period, self.period)
        outs = []
        acts = [layer[1] for layer in self.convs]
        if self.deferred_activation and deferrable(acts):
            # deferred form (layers.activation.PreActivated): every LeakyReLU rides on the NEXT convolution's operand load
            slope, pre = acts[0].slope, None
            for layer in self.convs:
                x = layer[0](x, pre_act=pre, pre_slope=slope)
                pre = "leaky_relu"
                outs.append(x)
            x = self.output_conv(x, pre_act=pre, pre_slope=slope)
            outs.append(torch.flatten(x, 1, -1))
            return PreActivated(outs, slope)
        for layer in self.convs:
            conv, act = layer[0], layer[1]
            x = conv(x, post_act=act.kind, post_slope=act.slope)
            outs.append(x)
        x = self.output_conv(x)
        outs.append(torch.flatten(x, 1, -1))
        return outs

    deferred_activation = False  # set by the trainer (layers.activation.set_deferred_activation)

    def apply_weight_norm(self):
        for m in self.modules():
            if isinstance(m, Conv2d):
                m.apply_weight_norm()
                logging.debug(f"Weight norm is applied to {m}.")

    def apply_spectral_norm(self):
        for m in self.modules():
            if isinstance(m, Conv2d):
                m.apply_spectral_norm()
                logging.debug(f"Spectral norm is applied to {m}.")


class HiFiGANMultiPeriodDiscriminator(torch.nn.Module):
    """HiFi-GAN multi-period discriminator (reference: models/hifigan.py:404-453)."""

    def __init__(self, periods=[2, 3, 5, 7, 11],
                 discriminator_params={"in_channels": 1, "out_channels": 1, "kernel_sizes": [5, 3], "channels": 32,
                                       "downsample_scales": [3, 3, 3, 3, 1], "max_downsample_channels": 1024,
                                       "bias": True, "nonlinear_activation": "LeakyReLU",
                                       "nonlinear_activation_params": {"negative_slope": 0.1},
                                       "use_weight_norm": True, "use_spectral_norm": False}):
        super().__init__()
        self.discriminators = torch.nn.ModuleList()
        for period in periods:
            params = copy.deepcopy(discriminator_params)
            params["period"] = period
            self.discriminators.append(HiFiGANPeriodDiscriminator(**params))

    branch_streams = False  # set True (e.g. by the trainer's hipGraph mode) to fork one stream per period

    def forward(self, x):
        return run_branches([(lambda d=d: d(x)) for d in self.discriminators], x.device, self.branch_streams, inputs=x)


class HiFiGANScaleDiscriminator(torch.nn.Module):
    """HiFi-GAN scale discriminator (reference: models/hifigan.py:456-702): conv k15, grouped
    strided convs k41, conv k5, conv k3; LeakyReLU fused into each producing kernel."""

    def __init__(self, in_channels=1, out_channels=1, kernel_sizes=[15, 41, 5, 3], channels=128,
                 max_downsample_channels=1024, max_groups=16, bias=True, downsample_scales=[2, 2, 4, 4, 1],
                 nonlinear_activation="LeakyReLU", nonlinear_activation_params={"negative_slope": 0.1},
                 use_weight_norm=True, use_spectral_norm=False):
        super().__init__()
        self.layers = torch.nn.ModuleList()
        assert len(kernel_sizes) == 4
        for ks in kernel_sizes:
            assert ks % 2 == 1

        def act():
            return FusedActivation(nonlinear_activation, **nonlinear_activation_params)

        self.layers.append(torch.nn.Sequential(
            Conv1d(in_channels, channels, kernel_sizes[0], bias=bias, padding=(kernel_sizes[0] - 1) // 2), act()))
        in_chs, out_chs, groups = channels, channels, 4
        for scale in downsample_scales:
            self.layers.append(torch.nn.Sequential(
                Conv1d(in_chs, out_chs, kernel_size=kernel_sizes[1], stride=scale, padding=(kernel_sizes[1] - 1) // 2,
                       groups=groups, bias=bias), act()))
            in_chs = out_chs
            out_chs = min(in_chs * 2, max_downsample_channels)
            groups = min(groups * 4, max_groups)
        out_chs = min(in_chs * 2, max_downsample_channels)
        self.layers.append(torch.nn.Sequential(
            Conv1d(in_chs, out_chs, kernel_size=kernel_sizes[2], stride=1, padding=(kernel_sizes[2] - 1) // 2,
                   bias=bias), act()))
        self.layers.append(Conv1d(out_chs, out_channels, kernel_size=kernel_sizes[3], stride=1,
                                  padding=(kernel_sizes[3] - 1) // 2, bias=bias))
        if use_weight_norm and use_spectral_norm:
            raise ValueError("Either use use_weight_norm or use_spectral_norm.")
        self.use_weight_norm = use_weight_norm
        if use_weight_norm:
            self.apply_weight_norm()
        self.use_spectral_norm = use_spectral_norm
        if use_spectral_norm:
            self.apply_spectral_norm()
        self._register_load_state_dict_pre_hook(self._load_state_dict_pre_hook)

    def forward(self, x):
        outs = []
        acts = [f[1] for f in self.layers if isinstance(f, torch.nn.Sequential)]
        if (self.deferred_activation and deferrable(acts) and isinstance(self.layers[0], torch.nn.Sequential)
                and all(isinstance(f, torch.nn.Sequential) for f in self.layers[:-1])):
            slope, pre = acts[0].slope, None
            for f in self.layers:
                conv = f[0] if isinstance(f, torch.nn.Sequential) else f
                x = conv(x, pre_act=pre, pre_slope=slope)
                pre = "leaky_relu"
                outs.append(x)
            return PreActivated(outs, slope)
        for f in self.layers:
            if isinstance(f, torch.nn.Sequential):
                conv, act = f[0], f[1]
                x = conv(x, post_act=act.kind, post_slope=act.slope)
            else:
                x = f(x)
            outs.append(x)
        return outs

    deferred_activation = False  # set by the trainer (layers.activation.set_deferred_activation)

    def _convs(self):
        return [m for m in self.modules() if isinstance(m, Conv1d)]

    def apply_weight_norm(self):
        for m in self._convs():
            m.apply_weight_norm()

    def apply_spectral_norm(self):
        for m in self._convs():
            m.apply_spectral_norm()

    def remove_weight_norm(self):
        for m in self._convs():
            if m.has_weight_norm:
                m.remove_weight_norm()

    def remove_spectral_norm(self):
        for m in self._convs():
            if m.has_spectral_norm:
                m.remove_spectral_norm()

    def _load_state_dict_pre_hook(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                  error_msgs):
        """Checkpoint compatibility (reference: models/hifigan.py:647-702): some published models
        were trained with a config that asks for weight/spectral norm although none was applied;
        when the incoming keys show that, drop the norm from this module so the keys match."""
        keys = [k for k in state_dict.keys() if k.startswith(prefix)]
        if self.use_weight_norm and not any("weight_g" in k for k in keys):
            logging.warning("The checkpoint has no weight-norm parameters for the scale discriminator although the "
                            "config enables it; removing weight norm from the current model to stay compatible "
                            "(set discriminator_params.follow_official_norm / use_weight_norm to false to silence).")
            self.remove_weight_norm()
            self.use_weight_norm = False
        if self.use_spectral_norm and not any("weight_u" in k for k in keys):
            logging.warning("The checkpoint has no spectral-norm buffers for the scale discriminator although the "
                            "config enables it; removing spectral norm from the current model to stay compatible.")
            self.remove_spectral_norm()
            self.use_spectral_norm = False


class HiFiGANMultiScaleDiscriminator(torch.nn.Module):
    """HiFi-GAN multi-scale discriminator (reference: models/hifigan.py:705-777)."""

    def __init__(self, scales=3, downsample_pooling="AvgPool1d",
                 downsample_pooling_params={"kernel_size": 4, "stride": 2, "padding": 2},
                 discriminator_params={"in_channels": 1, "out_channels": 1, "kernel_sizes": [15, 41, 5, 3],
                                       "channels": 128, "max_downsample_channels": 1024, "max_groups": 16,
                                       "bias": True, "downsample_scales": [2, 2, 4, 4, 1],
                                       "nonlinear_activation": "LeakyReLU",
                                       "nonlinear_activation_params": {"negative_slope": 0.1}},
                 follow_official_norm=False):
        super().__init__()
        self.discriminators = torch.nn.ModuleList()
        for i in range(scales):
            params = copy.deepcopy(discriminator_params)
            if follow_official_norm:
                # first scale: spectral norm; the others: weight norm (official implementation)
                params["use_weight_norm"] = i != 0
                params["use_spectral_norm"] = i == 0
            self.discriminators.append(HiFiGANScaleDiscriminator(**params))
        self.pooling = get_pooling(downsample_pooling, **downsample_pooling_params)

    branch_streams = False

    def forward(self, x):
        xs = [x]
        for _ in range(len(self.discriminators) - 1):  # the pooled inputs first (cheap, sequential) ...
            xs.append(self.pooling(xs[-1]))
        # ... then the scale discriminators as independent branches (the trailing pooling of the
        # reference's loop, whose result is never used, is dropped)
        return run_branches([(lambda f=f, xi=xi: f(xi)) for f, xi in zip(self.discriminators, xs)],
                            xs[0].device, self.branch_streams, inputs=xs)


class HiFiGANMultiScaleMultiPeriodDiscriminator(torch.nn.Module):
    """MSD + MPD (reference: models/hifigan.py:780-864); returns msd_outs + mpd_outs."""

    def __init__(self, scales=3, scale_downsample_pooling="AvgPool1d",
                 scale_downsample_pooling_params={"kernel_size": 4, "stride": 2, "padding": 2},
                 scale_discriminator_params={"in_channels": 1, "out_channels": 1, "kernel_sizes": [15, 41, 5, 3],
                                             "channels": 128, "max_downsample_channels": 1024, "max_groups": 16,
                                             "bias": True, "downsample_scales": [2, 2, 4, 4, 1],
                                             "nonlinear_activation": "LeakyReLU",
                                             "nonlinear_activation_params": {"negative_slope": 0.1}},
                 follow_official_norm=True, periods=[2, 3, 5, 7, 11],
                 period_discriminator_params={"in_channels": 1, "out_channels": 1, "kernel_sizes": [5, 3],
                                              "channels": 32, "downsample_scales": [3, 3, 3, 3, 1],
                                              "max_downsample_channels": 1024, "bias": True,
                                              "nonlinear_activation": "LeakyReLU",
                                              "nonlinear_activation_params": {"negative_slope": 0.1},
                                              "use_weight_norm": True, "use_spectral_norm": False}):
        super().__init__()
        self.msd = HiFiGANMultiScaleDiscriminator(
            scales=scales, downsample_pooling=scale_downsample_pooling,
            downsample_pooling_params=scale_downsample_pooling_params,
            discriminator_params=scale_discriminator_params, follow_official_norm=follow_official_norm)
        self.mpd = HiFiGANMultiPeriodDiscriminator(periods=periods, discriminator_params=period_discriminator_params)

    @property
    def branch_streams(self):
        return self.msd.branch_streams

    @branch_streams.setter
    def branch_streams(self, value):
        self.msd.branch_streams = bool(value)
        self.mpd.branch_streams = bool(value)

    def grad_groups(self, n_groups):
        """Exchange groups for data-parallel training (distributed.GradReducer): the 3 + 5 sub-discriminators
        share no parameters, so their backward passes can be issued -- and their gradients exchanged --
        group by group.  Periods first: the scale discriminators' 1/2 and 1/4-rate passes are the
        cheapest, which keeps the last (not overlappable) exchange next to the least compute."""
        from ..distributed import partition_modules

        return partition_modules(list(self.mpd.discriminators) + list(self.msd.discriminators), n_groups)

    def stateful_outputs(self):
        """Indices (into the returned list) of the sub-discriminators whose training-mode forward has a
        side effect -- the spectral-norm power iteration of the first scale discriminator.  The outputs
        of all the others depend only on (weights, input), so a trainer may reuse them while the weights
        are unchanged (bin/train.py: the real-signal pass of the generator phase feeds the discriminator
        phase)."""
        subs = list(self.msd.discriminators) + list(self.mpd.discriminators)
        return [i for i, d in enumerate(subs)
                if any(getattr(m, "has_spectral_norm", False) for m in d.modules())]

    def forward(self, x, only=None):
        """``only``: optional list of sub-discriminator indices to evaluate (the other entries of the
        returned list are None)."""
        xs = [x]
        for _ in range(len(self.msd.discriminators) - 1):  # (no pooling after the last scale: its result is unused)
            xs.append(self.msd.pooling(xs[-1]))
        if only is not None:
            subs = list(zip(self.msd.discriminators, xs)) + [(d, x) for d in self.mpd.discriminators]
            outs = [None] * len(subs)
            res = run_branches([(lambda f=subs[i][0], v=subs[i][1]: f(v)) for i in only], x.device,
                               self.msd.branch_streams and len(only) > 1, inputs=xs)
            for i, r in zip(only, res):
                outs[i] = r
            return outs
        if self.msd.branch_streams:  # one fork over all 3 + 5 sub-discriminators
            fns = [(lambda f=f, v=v: f(v)) for f, v in zip(self.msd.discriminators, xs)]
            fns += [(lambda d=d: d(x)) for d in self.mpd.discriminators]
            return run_branches(fns, x.device, True, inputs=xs)
        return self.msd(x) + self.mpd(x)
