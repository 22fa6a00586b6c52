"""Convolution modules whose arithmetic runs in libpwgkernels.so.

They replace ``torch.nn.Conv1d`` / ``torch.nn.ConvTranspose1d`` / ``torch.nn.Conv2d`` with
``(k, 1)`` kernels, together with the old-style ``torch.nn.utils.weight_norm`` /
``spectral_norm`` hooks, at the reference's call sites.  Parameter and buffer names and shapes
are the reference's, so its checkpoints load unchanged:
``weight``/``bias``; with weight norm ``weight_g``/``weight_v``/``bias``; with spectral norm
``weight_orig``/``weight_u``/``weight_v``/``bias``
(e.g. /root/reference/parallel_wavegan/models/hifigan.py:221-231, :383-402, :603-621).

Inference (no grad): the packed kernel weight image is derived once on the device and cached
until a parameter changes.  Training: the effective weight is a differentiable HIP node
(functional.WeightNormFn / SpectralNormFn) feeding functional.FusedConvFn.
"""
import math
import os

import torch

from .. import functional as Fn
from .. import ops


class _ConvNd(torch.nn.Module):
    transposed = False
    width_mode = False  # True for the (k, 1) Conv2d: input is (B, C, H, W)
    explicit_pad_min_elems = 1 << 20  # no-grad forward with reflect / replicate padding: see forward()
    # Batch folding (forward() / _fold_batch): on by default (C3 49.36 -> 48.29 ms, C5 45.52 -> 44.29 ms per captured
    # step, C4 / C2 unchanged; 64 or 128 columns per item measured slower: profiles/r05_fold_batch_ab.txt);
    # PWG_FOLD_BATCH=0 switches it off for A/B runs
    fold_batch = {"0": False, "1": True}.get(os.environ.get("PWG_FOLD_BATCH", ""), True)
    fold_max_cols = int(os.environ.get("PWG_FOLD_MAX_COLS", 40))  # output columns per item up to which a layer is folded
    fold_min_weight_bytes = 4 << 20  # ... if its weight is at least this large (the launch streams it once per item)

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, output_padding=0, pad_mode="zero"):
        super().__init__()
        if in_channels % groups or out_channels % groups:
            raise ValueError("in_channels and out_channels must be divisible by groups")
        self.in_channels, self.out_channels = in_channels, out_channels
        # padding: int (both sides) or (left, right) -- causal layers pad on the left only
        if isinstance(padding, (tuple, list)):
            padding, self.padding_right = int(padding[0]), int(padding[1])
        else:
            self.padding_right = int(padding)
        self.kernel_size, self.stride, self.padding = int(kernel_size), int(stride), int(padding)
        self.dilation, self.groups, self.output_padding = int(dilation), int(groups), int(output_padding)
        self.pad_mode = pad_mode
        self.weight = torch.nn.Parameter(torch.empty(self._weight_shape()))
        if bias:
            self.bias = torch.nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self._cache_key = None
        self._cache_packed = None
        self.spectral_eps = 1e-12
        self.reset_parameters()

    def _weight_shape(self):
        if self.transposed:
            return (self.in_channels, self.out_channels // self.groups, self.kernel_size)
        return (self.out_channels, self.in_channels // self.groups, self.kernel_size)

    # -- initialisation identical in distribution to torch.nn.Conv1d's default
    def reset_parameters(self):
        w = self.raw_weight
        fan_in = int(w[0].numel())
        bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
        with torch.no_grad():
            w.uniform_(-bound, bound)  # kaiming_uniform_(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
            if self.bias is not None:
                self.bias.uniform_(-bound, bound)

    @property
    def raw_weight(self):
        """The directly trainable weight-shaped parameter (weight / weight_v / weight_orig)."""
        if self.has_weight_norm:
            return self.weight_v
        if self.has_spectral_norm:
            return self.weight_orig
        return self.weight

    # -- old-style weight norm (dim=0): w = g * v / ||v||
    @property
    def has_weight_norm(self):
        return "weight_g" in self._parameters

    @property
    def has_spectral_norm(self):
        return "weight_orig" in self._parameters

    def apply_weight_norm(self):
        if self.has_weight_norm:
            return self
        if self.has_spectral_norm:
            raise ValueError("spectral norm is already applied")
        w = self._parameters.pop("weight")
        with torch.no_grad():
            g = w.detach().reshape(w.shape[0], -1).norm(dim=1).reshape((-1,) + (1,) * (w.dim() - 1))
        self.weight_g = torch.nn.Parameter(g.clone())
        self.weight_v = torch.nn.Parameter(w.detach().clone())
        self._cache_key = None
        return self

    def remove_weight_norm(self):
        if not self.has_weight_norm:
            raise ValueError(f"weight norm is not applied to {self.__class__.__name__}")
        g, v = self._parameters.pop("weight_g"), self._parameters.pop("weight_v")
        self.weight = torch.nn.Parameter(self._bake_weight_norm(g.detach(), v.detach()))
        self._cache_key = None
        return self

    @staticmethod
    def _bake_weight_norm(g, v):
        """One-off parameter conversion for ``remove_weight_norm`` (checkpoint bookkeeping; the
        reference calls it on CPU before ``.to(device)``, bin/decode.py:147-149).  On the device
        it runs on the HIP kernels; on the host it is plain parameter arithmetic, never a
        substitute for the compute path."""
        if v.is_cuda:
            vc = v.contiguous()
            return ops.scale_rows(vc, ops.weight_norm_scale(vc, g.reshape(-1).contiguous()))
        n = v.reshape(v.shape[0], -1).norm(dim=1).reshape(g.shape)
        return v * (g / n)

    # -- old-style spectral norm (dim=0, one power iteration per training forward)
    def apply_spectral_norm(self):
        if self.has_spectral_norm:
            return self
        if self.has_weight_norm:
            raise ValueError("weight norm is already applied")
        w = self._parameters.pop("weight")
        rows, cols = w.shape[0], w[0].numel()
        with torch.no_grad():
            # torch.nn.utils.spectral_norm: u ~ N(0,1) normalised, v likewise, then 15 warm-up
            # iterations at construction time (host-side initialisation only)
            u = torch.nn.functional.normalize(torch.randn(rows), dim=0, eps=self.spectral_eps)
            v = torch.nn.functional.normalize(torch.randn(cols), dim=0, eps=self.spectral_eps)
            wm = w.detach().reshape(rows, cols)
            for _ in range(15):
                v = torch.nn.functional.normalize(torch.mv(wm.t(), u), dim=0, eps=self.spectral_eps)
                u = torch.nn.functional.normalize(torch.mv(wm, v), dim=0, eps=self.spectral_eps)
        self.weight_orig = torch.nn.Parameter(w.detach().clone())
        self.register_buffer("weight_u", u)
        self.register_buffer("weight_v", v)
        self._cache_key = None
        return self

    def remove_spectral_norm(self):
        if not self.has_spectral_norm:
            raise ValueError(f"spectral norm is not applied to {self.__class__.__name__}")
        w = self._parameters.pop("weight_orig")
        u, v = self._buffers.pop("weight_u"), self._buffers.pop("weight_v")
        with torch.no_grad():
            wm = w.detach().reshape(w.shape[0], -1)
            sigma = torch.dot(u, torch.mv(wm, v))
        self.weight = torch.nn.Parameter(w.detach() / sigma)
        self._cache_key = None
        return self

    # -- effective weight
    def weight_tensor(self):
        """Differentiable effective weight (torch layout) computed by HIP kernels."""
        if self.has_weight_norm:
            return Fn.WeightNormFn.apply(self.weight_v, self.weight_g)
        if self.has_spectral_norm:
            return Fn.SpectralNormFn.apply(self.weight_orig, self.weight_u, self.weight_v, self.training,
                                           self.spectral_eps)
        return self.weight

    def effective_weight(self):
        with torch.no_grad():
            return self.weight_tensor().detach()

    def _params_key(self):
        ps = [self.raw_weight] + ([self.weight_g] if self.has_weight_norm else [])
        if self.has_spectral_norm:
            ps += [self.weight_u, self.weight_v]
        # the fused optimizers update parameters through raw pointers: they bump a per-parameter
        # epoch (ops.param_epoch) instead of torch's version counter
        return (ops.PARAM_EPOCH[0],) + tuple((p.data_ptr(), ops.tensor_version(p), ops.param_epoch(p), str(p.device)) for p in ps)

    def prepared(self):
        """:class:`functional.PreparedWeights` for the current parameter values (weight or weight-norm
        layers): weight-norm scale, packed forward image and, lazily, the packed data-gradient image.
        Shared by every forward / backward until a parameter changes."""
        key = self._params_key()
        if self._cache_key != key or self._cache_packed is None:
            desc = self.make_desc(1, self._probe_len())
            with torch.no_grad():
                if self.has_weight_norm:
                    v = self._w3(self.weight_v.detach())
                    scale = ops.weight_norm_scale(v, self.weight_g.detach().reshape(-1).contiguous())
                    self._cache_packed = Fn.PreparedWeights(key, v, scale, None, desc)
                else:
                    w = self._w3(self.effective_weight())
                    self._cache_packed = Fn.PreparedWeights(key, w, None, None, desc)
            self._cache_key = key
        return self._cache_packed

    def packed_weight(self):
        """Cached forward weight image (no-grad path)."""
        if self.has_spectral_norm and self.training:
            # every training-mode forward performs a power iteration: nothing to cache
            return ops.pack_weight(self.make_desc(1, self._probe_len()), self._w3(self.effective_weight()))
        return self.prepared().fwd

    @staticmethod
    def _w3(w):
        return w.reshape(w.shape[0], w.shape[1], -1).contiguous()

    def _probe_len(self):
        # any length with at least one output column (packing does not depend on it)
        return self.kernel_size * self.dilation + self.stride + max(0, -self.padding_right)

    def geom(self):
        return dict(kernel=self.kernel_size, stride=self.stride, dilation=self.dilation, padding=self.padding,
                    padding_right=self.padding_right, groups=self.groups, transposed=self.transposed,
                    output_padding=self.output_padding, width=1, pad_mode=self.pad_mode)

    def out_length(self, t_in):
        raise NotImplementedError

    def make_desc(self, batch, t_in, width=1, **fused):
        raise NotImplementedError

    def _needs_grad(self, *tensors):
        if not torch.is_grad_enabled():
            return False
        if any(p.requires_grad for p in self.parameters(recurse=False)):
            return True
        return any(t is not None and t.requires_grad for t in tensors)

    def _fold_batch(self, x, add1, add2, precomputed):
        """Few columns per item under a large weight (the 1024-channel tail of the scale discriminators at T = 9 .. 32,
        reference: models/hifigan.py:529-601): every item's workgroups stream the whole weight image for a handful of
        columns, and the launch takes the same ~80 us at T = 9, 17 and 32 (profiles/r05_tile_order_ab.txt).  A
        convolution along T treats the items exactly like the independent width columns of the period discriminators'
        (k, 1) layers, so such a layer runs as ONE item of width B over the activations transposed to (C, T, B):
        B x fewer passes over the weights, full column tiles, and the weight gradient's reduction chunks are full too.
        The transposing copies (2 x the activation bytes, a few MB) are the price."""
        if not self.fold_batch or self.width_mode or self.transposed or x.dim() != 3 or self.pad_mode != "zero":
            return False
        if add1 is not None or add2 is not None or precomputed is not None:
            return False
        b, t_out = x.shape[0], self.out_length(x.shape[-1])
        # (the layer's whole weight, groups included: restricting the rule to >= 4 MB PER GROUP -- i.e. to the k = 5 layer
        # -- because the grouped layers' forward measured 87 instead of 80 us stand-alone gave C3 -0.3 ms instead of
        # -1.1 ms and C5 -0.8 instead of -1.2 ms per captured step: their weight gradients gain more than that)
        w_bytes = 4 * self.out_channels * (self.in_channels // self.groups) * self.kernel_size
        return b >= 4 and 0 < t_out <= self.fold_max_cols and w_bytes >= self.fold_min_weight_bytes

    def forward(self, x, pre_act=None, pre_slope=0.0, post_act=None, post_slope=0.0, add1=None, add2=None,
                out_mul=1.0, out_div=1.0, precomputed=None):
        """Fused ``post((conv(pre(x)) + bias + add1 + add2) * out_mul / out_div)``.  ``precomputed`` (autograd path
        only): this layer's output as produced by a multi-layer kernel -- no launch, the node is recorded for backward."""
        if self._fold_batch(x, add1, add2, precomputed):
            b, c, t = x.shape
            xf = x.permute(1, 2, 0).contiguous().view(1, c, t, b)
            y = self._forward(xf, b, pre_act, pre_slope, post_act, post_slope, None, None, out_mul, out_div, None)
            return y.reshape(self.out_channels, -1, b).permute(2, 0, 1).contiguous()
        return self._forward(x, 0, pre_act, pre_slope, post_act, post_slope, add1, add2, out_mul, out_div, precomputed)

    def _forward(self, x, folded, pre_act, pre_slope, post_act, post_slope, add1, add2, out_mul, out_div, precomputed):
        """``folded``: 0, or the width of a batch-folded input (1, C, T, B) of a layer that is not in width mode."""
        fused = dict(pre_act=pre_act, pre_slope=pre_slope, post_act=post_act, post_slope=post_slope,
                     out_mul=out_mul, out_div=out_div)
        width_mode = self.width_mode or folded > 0
        if self._needs_grad(x, add1, add2):
            geom = self.geom()
            if width_mode:
                geom["width"] = x.shape[-1]
            if self.pad_mode != "zero" and (self.padding > 0 or self.padding_right > 0):
                # the backward kernels implement zero padding: pad explicitly (HIP kernel with its own
                # backward); pad and the element-wise pre-activation commute
                x = Fn.pad1d(x, self.padding, self.padding_right, self.pad_mode)
                geom["padding"], geom["padding_right"], geom["pad_mode"] = 0, 0, "zero"
            if self.has_spectral_norm:
                # one power iteration per training forward: the weight is a fresh autograd node
                assert precomputed is None
                return Fn.FusedConvFn.apply(x, self.weight_tensor(), self.bias, add1, add2, geom, fused, None)
            if self.has_weight_norm:
                return Fn.FusedConvFn.apply(x, self.weight_v, self.bias, add1, add2, geom, fused, self.prepared(),
                                            self.weight_g, precomputed)
            return Fn.FusedConvFn.apply(x, self.weight, self.bias, add1, add2, geom, fused, self.prepared(), None,
                                        precomputed)
        assert precomputed is None, "precomputed outputs only make sense on the autograd path"
        with torch.no_grad():
            b = x.shape[0]
            if width_mode:
                h, width = x.shape[2], x.shape[3]
                desc = self.make_desc(b, h, width=width, **fused)
                y = ops.conv1d_forward(desc, x.reshape(b, x.shape[1], -1).contiguous(), self.packed_weight(),
                                       None if self.bias is None else self.bias.detach(),
                                       None if add1 is None else add1.reshape(b, self.out_channels, -1),
                                       None if add2 is None else add2.reshape(b, self.out_channels, -1))
                return y.reshape(b, self.out_channels, desc.t_out, width)
            if (self.pad_mode != "zero" and (self.padding > 0 or self.padding_right > 0) and not self.transposed
                    and x.numel() >= self.explicit_pad_min_elems):
                # Large inputs: pad explicitly (one streaming launch) and run the zero-padding LDS-DMA kernel -- what the
                # training path does anyway.  The register-staged kernel that maps reflected indices itself runs at
                # 33 - 41 TFLOP/s on MelGAN's residual stacks (96 channels, k = 3, B64 x T2048: 176 us) against 103 us
                # for the DMA kernel + 25 us for the pad (profiles/r04_k1_sweep.txt, r04_b_train_shapes_c4.txt).
                xp = Fn.pad1d(x, self.padding, self.padding_right, self.pad_mode)
                t_out = self.out_length(x.shape[-1])
                desc = ops.make_conv_desc(b, self.in_channels, self.out_channels, xp.shape[-1], t_out, self.kernel_size,
                                          self.stride, self.dilation, 0, self.groups, transposed=False, **fused)
                return ops.conv1d_forward(desc, xp, self.packed_weight(),
                                          None if self.bias is None else self.bias.detach(), add1, add2)
            desc = self.make_desc(b, x.shape[-1], **fused)
            return ops.conv1d_forward(desc, x.contiguous(), self.packed_weight(),
                                      None if self.bias is None else self.bias.detach(), add1, add2)

    def extra_repr(self):
        norm = "weight_norm" if self.has_weight_norm else ("spectral_norm" if self.has_spectral_norm else "none")
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, "
                f"padding={self.padding}, dilation={self.dilation}, groups={self.groups}, norm={norm}")


class Conv1d(_ConvNd):
    """Drop-in for ``torch.nn.Conv1d`` (zero / reflect / replicate implicit padding)."""

    transposed = False

    def out_length(self, t_in):
        return ops.conv_out_length(t_in, self.kernel_size, self.stride, self.dilation, self.padding, self.padding_right)

    def make_desc(self, batch, t_in, width=1, **fused):
        return ops.make_conv_desc(batch, self.in_channels, self.out_channels, t_in, self.out_length(t_in),
                                  self.kernel_size, self.stride, self.dilation, self.padding, self.groups,
                                  transposed=False, width=width, pad_mode=self.pad_mode, **fused)


class Conv2d(Conv1d):
    """Drop-in for ``torch.nn.Conv2d`` restricted to ``(k, 1)`` kernels, ``(s, 1)`` strides and
    ``(p, 0)`` padding -- the only 2-D convolutions on the hot path (period discriminator,
    /root/reference/parallel_wavegan/models/hifigan.py:314-341).  Input (B, C, H, W); the
    weight keeps torch's (C_out, C_in, k, 1) shape."""

    width_mode = True

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        def _first(v, name):
            if isinstance(v, (tuple, list)):
                if len(v) != 2 or (name == "kernel_size" and v[1] != 1) or (name == "stride" and v[1] != 1) or (
                        name == "padding" and v[1] != 0):
                    raise NotImplementedError(f"Conv2d: only (k,1) kernels / (s,1) strides / (p,0) padding, got {name}={v}")
                return v[0]
            if name == "stride":
                if v != 1:
                    raise NotImplementedError("Conv2d: an int stride > 1 would also stride the width axis")
                return 1
            raise NotImplementedError(f"Conv2d: {name} must be given as a (k, 1)-style tuple")

        super().__init__(in_channels, out_channels, _first(kernel_size, "kernel_size"), _first(stride, "stride"),
                         _first(padding, "padding"), bias=bias)

    def _weight_shape(self):
        return (self.out_channels, self.in_channels // self.groups, self.kernel_size, 1)


class ConvTranspose1d(_ConvNd):
    """Drop-in for ``torch.nn.ConvTranspose1d`` (polyphase; weight (C_in, C_out/groups, k))."""

    transposed = True

    def out_length(self, t_in):
        return ops.conv_transpose_out_length(t_in, self.kernel_size, self.stride, self.padding, self.output_padding)

    def make_desc(self, batch, t_in, width=1, **fused):
        return ops.make_conv_desc(batch, self.in_channels, self.out_channels, t_in, self.out_length(t_in),
                                  self.kernel_size, self.stride, 1, self.padding, self.groups, transposed=True,
                                  width=width, **fused)
