"""Differentiable wrappers: ``torch.autograd.Function`` nodes whose forward AND backward are
HIP kernels from libpwgkernels.so.  torch's autograd engine only orders the nodes and sums
gradients of tensors used more than once; it performs no convolution/loss arithmetic here.
"""
import ctypes
import math
import os

import torch

from . import _lib, ops
from .ops import _ptr, _require_device, _stream

_L = _lib.lib


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------------------------------------
# weight reparametrisations
# ---------------------------------------------------------------------------------------------
class WeightNormFn(torch.autograd.Function):
    """w = g * v / ||v||  (old-style torch.nn.utils.weight_norm, dim=0)."""

    @staticmethod
    def forward(ctx, v, g):
        v = _c(v)
        g = _c(g)
        scale = ops.weight_norm_scale(v, g.reshape(-1))
        ctx.save_for_backward(v, g)
        return ops.scale_rows(v, scale)

    @staticmethod
    def backward(ctx, dw):
        v, g = ctx.saved_tensors
        dw = _c(dw)
        _require_device(dw)
        dv = torch.empty_like(v)
        dg = torch.empty_like(g)
        n0 = v.shape[0]
        _lib.check(_L().pwg_weight_norm_backward(_ptr(dw), _ptr(v), _ptr(g), _ptr(dv), _ptr(dg), n0,
                                                 v.numel() // n0, _stream()), "weight_norm_backward")
        return dv, dg


class Unpack2Fn(torch.autograd.Function):
    """(v[0], v[1]) of a 2-vector as two 0-dim tensors whose backward is ONE launch (``v[0], v[1]`` by indexing costs two
    zero fills, two copies and an accumulation in the backward pass)."""

    @staticmethod
    def forward(ctx, v):
        out = v.clone()
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g0, g1):
        if g0 is None and g1 is None:
            return None
        z = (g0 if g0 is not None else g1).new_zeros(())
        return torch.stack([g0 if g0 is not None else z, g1 if g1 is not None else z])


class SpectralNormFn(torch.autograd.Function):
    """w = w_orig / sigma, sigma = u^T W v after (optionally) one power iteration that updates the
    ``u``/``v`` buffers in place, as torch.nn.utils.spectral_norm does in training mode."""

    @staticmethod
    def forward(ctx, w_orig, u, v, do_iter, eps):
        w_orig = _c(w_orig)
        _require_device(w_orig, u, v)
        rows = w_orig.shape[0]
        cols = w_orig.numel() // rows
        sigma = torch.empty(1, device=w_orig.device, dtype=torch.float32)
        w = torch.empty_like(w_orig)
        tmp = torch.empty(max(rows, 32 * cols), device=w_orig.device, dtype=torch.float32)
        # torch clones u, v after the iteration so that later in-place updates do not alter this graph: the iteration's
        # kernels write those copies themselves (without an iteration: cloned here, as before)
        u_saved, v_saved = (torch.empty_like(u), torch.empty_like(v)) if do_iter else (u.clone(), v.clone())
        _lib.check(_L().pwg_spectral_norm_forward_saved(_ptr(w_orig), _ptr(u), _ptr(v), _ptr(sigma), _ptr(w), _ptr(tmp),
                                                        _ptr(u_saved), _ptr(v_saved), rows, cols, int(bool(do_iter)),
                                                        float(eps), _stream()), "spectral_norm_forward_saved")
        ctx.save_for_backward(w_orig, u_saved, v_saved, sigma)
        return w

    @staticmethod
    def backward(ctx, dw):
        w_orig, u, v, sigma = ctx.saved_tensors
        dw = _c(dw)
        rows = w_orig.shape[0]
        cols = w_orig.numel() // rows
        dwo = torch.empty_like(w_orig)
        scratch = torch.empty(_lib.SPECTRAL_NORM_SCRATCH_FLOATS, device=dw.device, dtype=torch.float32)
        _lib.check(_L().pwg_spectral_norm_backward(_ptr(dw), _ptr(w_orig), _ptr(u), _ptr(v), _ptr(sigma), _ptr(dwo),
                                                   _ptr(scratch), rows, cols, _stream()), "spectral_norm_backward")
        return dwo, None, None, None, None


# ---------------------------------------------------------------------------------------------
# fused convolution
# ---------------------------------------------------------------------------------------------
class PreparedWeights:
    """Kernel-side images of one layer's weight for the current parameter values: the weight-norm row
    scale, the packed forward image and (built on first use) the packed data-gradient image.  A layer
    keeps one instance per parameter epoch, so every forward / backward of that epoch -- e.g. D(y) and
    D(G(c)) of a discriminator phase -- shares a single scale + pack + pack launch sequence."""

    __slots__ = ("key", "w", "_scale", "_fwd", "_fwd_desc", "_bwd", "_res", "_stale")

    def __init__(self, key, w, scale, fwd=None, fwd_desc=None):
        """``fwd``: the packed forward image, or None with ``fwd_desc`` (any descriptor of the layer) to build it
        on first use -- layers that only run inside a fused multi-layer kernel never need it."""
        self.key, self.w, self._scale, self._fwd, self._fwd_desc, self._bwd, self._res = key, w, scale, fwd, fwd_desc, None, None
        self._stale = False  # set by weight_bank.WeightBank when it overwrites the (shared, persistent) images

    @property
    def scale(self):
        """Weight-norm row scale g / |v| (None without weight norm); a bank-owned buffer like the images: checked."""
        self._check()
        return self._scale

    def _check(self):
        if self._stale:
            raise RuntimeError("these prepared weights were overwritten by a later WeightBank.ensure(): a backward pass "
                               "through a graph built before the last parameter update is not supported with a "
                               "weight bank (call backward before the optimizer step, or disable the bank)")

    @property
    def fwd(self):
        self._check()
        if self._fwd is None:
            with torch.no_grad():
                self._fwd = ops.pack_weight(self._fwd_desc, self.w, self._scale)
        return self._fwd

    def res(self):
        """MFMA A-operand image for the one-launch residual unit (csrc/resunit.hip), built on first use."""
        self._check()
        if self._res is None:
            with torch.no_grad():
                self._res = ops.resunit_pack_weight(self.w, self._scale)
        return self._res

    def bwd(self, desc):
        self._check()
        if self._bwd is None:
            with torch.no_grad():
                self._bwd = ops.pack_weight_bwd(desc, self.w, self._scale)
        return self._bwd


class FusedConvFn(torch.autograd.Function):
    """y = post_act((conv(pre_act(x), w) + bias + add1 + add2) * out_mul / out_div).

    ``geom`` = dict(kernel, stride, dilation, padding, groups, transposed, output_padding, width,
    pad_mode); ``fused`` = dict(pre_act, pre_slope, post_act, post_slope, out_mul, out_div).
    ``packed``: None (pack here), a packed forward image, or a :class:`PreparedWeights`.
    ``g``: when given, ``w`` is the weight-norm direction ``v`` and the effective weight is
    ``g * v / ||v||`` (folded into the packed images); backward returns (dv, dg).
    """

    @staticmethod
    def forward(ctx, x, w, bias, add1, add2, geom, fused, packed, g=None, precomputed=None):
        """``precomputed``: the value of this convolution, already produced by a kernel that fuses several layers
        (csrc/resstack.hip): nothing is launched here, the node only records what its backward needs."""
        x = _c(x)
        b = x.shape[0]
        width = geom.get("width", 1)
        t_in = x.shape[-1] // width if x.dim() == 3 else x.shape[2]
        x3 = x.reshape(b, x.shape[1], -1)
        c_in = x3.shape[1]
        transposed = geom["transposed"]
        k = geom["kernel"]
        if transposed:
            c_out = w.shape[1] * geom["groups"]
            t_out = ops.conv_transpose_out_length(t_in, k, geom["stride"], geom["padding"], geom["output_padding"])
        else:
            c_out = w.shape[0]
            t_out = geom.get("t_out")
            if t_out is None:
                t_out = ops.conv_out_length(t_in, k, geom["stride"], geom["dilation"], geom["padding"],
                                            geom.get("padding_right", geom["padding"]))
        desc = ops.make_conv_desc(b, c_in, c_out, t_in, t_out, k, geom["stride"], geom["dilation"], geom["padding"],
                                  geom["groups"], transposed=transposed, width=width,
                                  pad_mode=geom.get("pad_mode", "zero"), **fused)
        wc = _c(w.reshape(w.shape[0], w.shape[1], -1))
        holder = packed if isinstance(packed, PreparedWeights) else None
        if holder is None:
            scale = None
            if g is not None:
                scale = ops.weight_norm_scale(wc, _c(g).reshape(-1))
            fwd = packed if packed is not None else ops.pack_weight(desc, wc, scale)
            holder = PreparedWeights(None, wc, scale, fwd)
        add1c = None if add1 is None else _c(add1).reshape(b, c_out, -1)
        add2c = None if add2 is None else _c(add2).reshape(b, c_out, -1)
        if precomputed is None:
            y = ops.conv1d_forward(desc, x3, holder.fwd, None if bias is None else _c(bias), add1c, add2c)
        else:
            if tuple(precomputed.shape) != (b, c_out, t_out * width) or not precomputed.is_contiguous():
                raise ValueError(f"FusedConvFn: precomputed output has shape {tuple(precomputed.shape)}, "
                                 f"expected {(b, c_out, t_out * width)}")
            y = precomputed.view(b, c_out, t_out * width)  # (a new tensor object: the output of this node)
        ctx.desc = desc
        ctx.has = (bias is not None, add1 is not None, add2 is not None)
        ctx.fused = fused
        ctx.w_shape = tuple(wc.shape)
        ctx.w_orig_shape = tuple(w.shape)
        ctx.x_shape = tuple(x.shape)
        ctx.holder = holder
        ctx.has_g = g is not None
        # addresses of the parameters whose gradients this node produces (data-parallel gradient slots, ops.GRAD_SLOTS)
        ctx.slot_keys = (w.data_ptr(), None if bias is None else bias.data_ptr(), None if g is None else g.data_ptr()) \
            if ops.GRAD_SLOTS else None
        need_y = fused.get("post_act") not in (None, "none")
        ctx.save_for_backward(x3, y if need_y else None, wc if g is not None else None,
                              _c(g) if g is not None else None)
        if width > 1:
            return y.reshape(b, c_out, t_out, width)
        return y

    @staticmethod
    def backward(ctx, dy):
        x3, y, v, g = ctx.saved_tensors
        desc, fused = ctx.desc, ctx.fused
        dy = _c(dy).reshape(desc.batch, desc.c_out, -1)
        _require_device(dy)
        has_bias, has_add1, has_add2 = ctx.has
        # gradient w.r.t. the pre-post_act, pre-scale sum
        scale = float(fused.get("out_mul", 1.0)) / float(fused.get("out_div", 1.0))
        post = fused.get("post_act")
        if post not in (None, "none") or scale != 1.0:
            gsum = torch.empty_like(dy)
            _lib.check(_L().pwg_act_backward(_ptr(dy), _ptr(y), _ptr(gsum), dy.numel(), ops.ACT[post],
                                             float(fused.get("post_slope", 0.0)), scale, _stream()), "act_backward")
        else:
            gsum = dy
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        need_g = ctx.has_g and ctx.needs_input_grad[8]
        dx = dw = db = dg = None
        if need_x:
            dx = ops.conv1d_backward_data(desc, gsum, ctx.holder.bwd(desc), x3).reshape(ctx.x_shape)
        wn_row_bytes = 4 * (ctx.w_shape[1] * ctx.w_shape[2])
        # data parallel: results go straight into the parameters' bucket slots where those are free (ops.claim_grad_slot)
        out_w = out_b = out_g = None
        acc_w = acc_b = acc_g = None  # slots that already hold this pass's first contribution: add into them
        later_w = later_b = False     # (what claim_grad_slots said: True or the stream of the first contribution)
        if ctx.slot_keys is not None and ops.GRAD_SLOTS:
            kw, kb, kg = ctx.slot_keys
            if need_b and has_bias:
                views, later = ops.claim_grad_slots([(kb, (desc.c_out,))])
                if views is not None:
                    out_b, acc_b = (None, views[0]) if later else (views[0], None)
                    later_b = later
            if ctx.has_g and need_w and need_g:
                views, later = ops.claim_grad_slots([(kw, ctx.w_shape), (kg, tuple(g.reshape(-1).shape))])
                if views is not None:
                    (out_w, out_g), (acc_w, acc_g) = ((None, None), views) if later else (views, (None, None))
                    later_w = later
            elif not ctx.has_g and need_w:
                views, later = ops.claim_grad_slots([(kw, ctx.w_shape)])
                if views is not None:
                    out_w, acc_w = (None, views[0]) if later else (views[0], None)
                    later_w = later
        if ctx.has_g and (need_w or need_g) and wn_row_bytes + 256 <= 64 * 1024:
            # weight-normalised layer: slabs -> (dv, dg) in one fused finishing kernel
            dv, dg, db = ops.conv1d_backward_weight_wn(desc, x3, gsum, v, g.reshape(-1),
                                                       need_db=need_b and has_bias, out_dv=out_w, out_dg=out_g,
                                                       out_db=out_b)
            dw = dv.reshape(ctx.w_orig_shape)
            dg = dg.reshape(g.shape)
        elif need_w or need_g or (need_b and has_bias):
            dw, db = ops.conv1d_backward_weight(desc, x3, gsum, ctx.w_shape, need_dw=need_w or need_g,
                                                need_db=need_b and has_bias,
                                                out_dw=None if ctx.has_g else out_w, out_db=out_b)
            if dw is not None and ctx.has_g:
                # weight norm backward: (dv, dg) from the gradient w.r.t. the effective weight
                dv = out_w if out_w is not None else torch.empty_like(v)
                dg = out_g.view(g.shape) if out_g is not None else torch.empty_like(g)
                n0 = v.shape[0]
                _lib.check(_L().pwg_weight_norm_backward(_ptr(dw), _ptr(v), _ptr(g), _ptr(dv), _ptr(dg), n0,
                                                         v.numel() // n0, _stream()), "weight_norm_backward")
                dw = dv
            if dw is not None:
                dw = dw.reshape(ctx.w_orig_shape)
        # later contributions of a backward pass (the discriminator phase differentiates D(y) and D(G(c)) together):
        # added into the bucket slot here; autograd gets None and has nothing to sum or copy
        if acc_w is not None and dw is not None:
            ops.slot_add(acc_w, dw, later_w)
            dw = None
        if acc_g is not None and dg is not None:
            ops.slot_add(acc_g, dg, later_w)
            dg = None
        if acc_b is not None and db is not None:
            ops.slot_add(acc_b, db, later_b)
            db = None
        gshape = dy.shape if desc.width == 1 else (desc.batch, desc.c_out, desc.t_out, desc.width)
        dadd1 = gsum.reshape(gshape) if has_add1 and ctx.needs_input_grad[3] else None
        dadd2 = gsum.reshape(gshape) if has_add2 and ctx.needs_input_grad[4] else None
        return dx, dw, db, dadd1, dadd2, None, None, None, dg, None


# ---------------------------------------------------------------------------------------------
# weight-gradient launches beside the data path
# ---------------------------------------------------------------------------------------------
# The WaveNet layers' backward is a strict chain (gate -> data gradient -> the previous layer's gate ...) whose
# kernels leave half of the matrix pipes idle, and every layer's three weight-gradient launches depend on the chain
# but nothing in the chain depends on them.  They go to a side stream: in a captured step they become a parallel
# branch of the graph and run beside the data gradient of their layer and the (HBM-bound) gate kernel of the next.
# The caller's stream is joined with the side stream (a) right away when something consumes the parameter gradients
# inside the backward pass (an existing ``.grad`` to add to, tensor or post-accumulate hooks -- the data-parallel
# reducer), else (b) once, by an engine callback at the end of the backward pass.
# Measured on the PWG.v1 step (B6 x 25600, profiles/r04_wavenet_wgrad_variants.txt): eager launches 29.08 -> 28.11 ms;
# inside a captured hipGraph the two branches run concurrently but stretch each other (27.97 vs 28.10 ms), so the
# default ("auto") forks only when the stream is not being captured.  PWG_WAVENET_WGRAD_STREAM=0 / 1: never / always.
WGRAD_SIDE_STREAM = {"0": False, "1": True}.get(os.environ.get("PWG_WAVENET_WGRAD_STREAM", "auto"), "auto")


def _wgrad_fork_now():
    if WGRAD_SIDE_STREAM == "auto":
        return not torch.cuda.is_current_stream_capturing()
    return bool(WGRAD_SIDE_STREAM)


_WGRAD_STREAMS = {}
_WGRAD_JOIN_QUEUED = set()


def _wgrad_side_stream(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    s = _WGRAD_STREAMS.get(key)
    if s is None:
        s = _WGRAD_STREAMS[key] = torch.cuda.Stream(device=device)
    return s


def _grads_consumed_in_pass(params):
    for p in params:
        if p is None:
            continue
        if p.grad is not None or getattr(p, "_post_accumulate_grad_hooks", None) or getattr(p, "_backward_hooks", None):
            return True
    return False


def _join_wgrad_side_stream(device, deferred):
    """Make the current stream wait for the side stream's weight-gradient launches: now, or (``deferred``) when the
    running backward pass ends (the engine runs such callbacks on the caller's streams, before ``backward()`` returns)."""
    side = _wgrad_side_stream(device)
    if not deferred:
        torch.cuda.current_stream(device).wait_stream(side)
        return
    # one callback per (device, backward pass); entries of a pass that died with an exception are harmless (ids are unique)
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch._C._current_graph_task_id())
    if key in _WGRAD_JOIN_QUEUED:
        return
    _WGRAD_JOIN_QUEUED.add(key)
    node_stream = torch.cuda.current_stream(device)  # (= the stream of the layer's forward: the engine set it)

    def join():
        _WGRAD_JOIN_QUEUED.discard(key)
        node_stream.wait_stream(side)
        cur = torch.cuda.current_stream(device)  # the stream around the caller's backward() (see GraphTask post-processing)
        if cur != node_stream:
            cur.wait_stream(side)

    torch.autograd.Variable._execution_engine.queue_callback(join)


def conv_param_grads(desc, x3, gsum, w_shape3, w_orig_shape, v, g, need_w, need_g, need_b):
    """(dw or dv, dg, db) of a convolution from its input ``x3`` and the gradient ``gsum`` w.r.t. its
    pre-activation output: the weight-gradient kernel + the weight-norm finish when ``g`` is given."""
    dw = db = dg = None
    has_g = g is not None
    wn_row_bytes = 4 * (w_shape3[1] * w_shape3[2])
    if has_g and (need_w or need_g) and wn_row_bytes + 256 <= 64 * 1024:
        dv, dg, db = ops.conv1d_backward_weight_wn(desc, x3, gsum, v, g.reshape(-1), need_db=need_b)
        dw = dv.reshape(w_orig_shape)
        dg = dg.reshape(g.shape)
    elif need_w or need_g or need_b:
        dw, db = ops.conv1d_backward_weight(desc, x3, gsum, w_shape3, need_dw=need_w or need_g, need_db=need_b)
        if dw is not None and has_g:
            dv = torch.empty_like(v)
            dg = torch.empty_like(g)
            n0 = v.shape[0]
            _lib.check(_L().pwg_weight_norm_backward(_ptr(dw), _ptr(v), _ptr(g), _ptr(dv), _ptr(dg), n0,
                                                     v.numel() // n0, _stream()), "weight_norm_backward")
            dw = dv
        if dw is not None:
            dw = dw.reshape(w_orig_shape)
    return dw, dg, db


class WaveNetLayerFn(torch.autograd.Function):
    """One gated residual layer of the Parallel WaveGAN generator: forward = ONE launch (csrc/wavenet.hip, also
    writes the gate input z and output g for this backward); backward = the data / weight gradient kernels of
    its four convolutions and the gate (layers/residual_block.py:102-140 of the reference).

    ``block``: the :class:`layers.WaveNetResidualBlock` (weights, descriptors, cached images).  Tensor
    arguments after it are the block's parameters in the order of ``block.fused_params()`` -- they are passed so
    that autograd routes their gradients; values are read through ``block``."""

    @staticmethod
    def forward(ctx, x, c, skips, block, skip_scale, *params):
        x, c = _c(x), _c(c)
        skips = None if skips is None else _c(skips)
        _require_device(x, c, skips)
        desc = block.fused_desc(x.shape[0], x.shape[2], skip_scale)
        convs = block.fused_convs()
        x_out, s_out, z, gt = ops.wavenet_layer_forward(
            desc, x, c, skips, block.fused_image(), *[None if cv.bias is None else cv.bias.detach()
                                                       for cv in (convs[0], convs[2], convs[3])], save=True)
        ctx.block, ctx.desc = block, desc
        ctx.holders = [cv.prepared() for cv in convs]  # (the parameter values this forward used)
        ctx.has_skips = skips is not None
        ctx.param_refs = params
        # the backward reads the weights through ``block`` (packed images, weight-norm tensors), not through
        # saved_tensors: remember their versions so that an update between forward and backward is an error here too
        ctx.param_keys = tuple(cv._params_key()[1:] for cv in convs)
        ctx.save_for_backward(x, c, z, gt)
        ctx.set_materialize_grads(False)
        # third output: c itself, for the NEXT layer -- the gradient of the shared aux features then arrives here
        # already summed over the later layers and is an addend of this layer's data-gradient epilogue (one chain
        # through the 30 layers instead of 29 accumulation launches of autograd)
        c_next = c.view_as(c)
        if not c.requires_grad:
            ctx.mark_non_differentiable(c_next)
        return x_out, s_out, c_next

    @staticmethod
    def backward(ctx, dx_out, ds_out, dc_next=None):
        x, c, z, gt = ctx.saved_tensors
        block, desc = ctx.block, ctx.desc
        conv_d, conv_a, conv_s, conv_o = block.fused_convs()
        if tuple(cv._params_key()[1:] for cv in (conv_d, conv_a, conv_s, conv_o)) != ctx.param_keys:
            raise RuntimeError("WaveNetLayerFn.backward: a parameter of the layer was modified after the forward pass "
                               "(in-place update or optimizer step between forward and backward)")
        h_d, h_a, h_s, h_o = ctx.holders
        b, t = x.shape[0], x.shape[2]
        need = ctx.needs_input_grad
        d_o, d_s = conv_o.make_desc(b, t), conv_s.make_desc(b, t)
        d_d, d_a = conv_d.make_desc(b, t), conv_a.make_desc(b, t)
        dx_out = None if dx_out is None else _c(dx_out)
        if ds_out is None:  # (cannot happen in the generator: every layer's skip sum reaches the loss)
            ds_out = torch.zeros_like(x)
        ds_out = _c(ds_out)
        # data path: two launches (csrc/wavenet.hip): dz and go = out_mul * dx_out, then (below, after the weight
        # path has been forked off) dx (+ go) and dc
        img = block.fused_image_bwd(desc.skip_mul)
        dz, go = ops.wavenet_gate_backward(desc, z, dx_out, ds_out, img)
        # gradient w.r.t. the pre-scale sum of the skip convolution (its weight-gradient operand / the incoming skips)
        gs = ds_out
        if desc.skip_mul != 1.0:
            gs = torch.empty_like(ds_out)
            _lib.check(_L().pwg_act_backward(_ptr(ds_out), None, _ptr(gs), gs.numel(), 0, 0.0, float(desc.skip_mul),
                                             _stream()), "act_backward")
        grads = []
        pi = 5
        fused_w = None
        forked = False
        if os.environ.get("PWG_NO_WAVENET_WGRAD", "0") != "1":
            # weight path: every parameter gradient of the layer in three launches (csrc/wavenet.hip), issued on the
            # side stream right after the gate kernel (see _join_wgrad_side_stream)
            wn = [(hd.w if cv.has_weight_norm else None,
                   cv.weight_g.detach().reshape(-1) if cv.has_weight_norm else None, cv.bias is not None)
                  for cv, hd in ((conv_d, h_d), (conv_a, h_a), (conv_s, h_s), (conv_o, h_o))]
            forked = _wgrad_fork_now()
            if forked:
                dev = dz.device
                cur, side = torch.cuda.current_stream(dev), _wgrad_side_stream(dev)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    fused_w = ops.wavenet_weight_backward(desc, dz, x, c, gs, go, gt, convs=wn)
                for t_ in (dz, x, c, gs, go, gt):  # allocated on the caller's stream, read on the side stream
                    if t_ is not None:
                        t_.record_stream(side)
            else:
                fused_w = ops.wavenet_weight_backward(desc, dz, x, c, gs, go, gt, convs=wn)
        dx, dc = ops.wavenet_data_backward(desc, dz, go, img, need_dx=need[0], need_dc=need[1],
                                           dc_accum=None if dc_next is None else _c(dc_next))
        if fused_w is not None and forked:
            _join_wgrad_side_stream(dz.device, deferred=not _grads_consumed_in_pass(ctx.param_refs))
        for i, (cv, hd, dsc, xin, gsum) in enumerate(((conv_d, h_d, d_d, x, dz), (conv_a, h_a, d_a, c, dz),
                                                      (conv_s, h_s, d_s, gt, gs), (conv_o, h_o, d_o, gt, go))):
            has_g = cv.has_weight_norm
            n_par = 1 + int(has_g) + int(cv.bias is not None)
            if gsum is None:
                grads += [None] * n_par
                pi += n_par
                continue
            need_w = need[pi]
            need_g = has_g and need[pi + 1]
            need_b = cv.bias is not None and need[pi + 1 + int(has_g)]
            v = hd.w
            g = cv.weight_g.detach() if has_g else None
            if fused_w is not None:
                dw, dg, db = fused_w[i]
                dw = dw.reshape(cv.raw_weight.shape)
                dg = None if dg is None else dg.reshape(g.shape)
            else:
                dw, dg, db = conv_param_grads(dsc, xin, gsum, tuple(v.shape), tuple(cv.raw_weight.shape), v, g, need_w,
                                              need_g, need_b)
            grads.append(dw)
            if has_g:
                grads.append(dg)
            if cv.bias is not None:
                grads.append(db)
            pi += n_par
        return (dx, dc, (gs if ctx.has_skips else None), None, None) + tuple(grads)


class ResStackFn(torch.autograd.Function):
    """One MelGAN residual stack (layers/residual_stack.py:75-85 of the reference): forward = ONE launch
    (csrc/resstack.hip, also writes the dilated convolution's output h for this backward); backward = the unit's data
    gradient in one launch (dh and the padded-domain dx), the reflection's adjoint, and the three layers' own
    weight-gradient kernels.

    ``stack``: the :class:`layers.ResidualStack`.  Tensor arguments after it are its parameters in the order of
    ``stack.unit_params()`` -- passed so that autograd routes their gradients; values are read through ``stack``."""

    @staticmethod
    def forward(ctx, x, stack, *params):
        x = _c(x)
        _require_device(x)
        convs = stack.unit_convs()
        d, slope = convs[0].dilation, stack.unit_slope()
        bias = [None if cv.bias is None else cv.bias.detach() for cv in convs]
        y, h = ops.resstack_forward(x, stack.unit_image(), d, slope, *bias, save_h=True)
        ctx.stack, ctx.geom = stack, (d, slope)
        ctx.holders = [cv.prepared() for cv in convs]  # (the parameter values this forward used)
        ctx.param_keys = tuple(cv._params_key()[1:] for cv in convs)
        ctx.save_for_backward(x, h)
        ctx.set_materialize_grads(False)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, h = ctx.saved_tensors
        stack = ctx.stack
        convs = stack.unit_convs()
        n_par = [1 + int(cv.has_weight_norm) + int(cv.bias is not None) for cv in convs]
        if dy is None:
            return (None, None) + (None,) * sum(n_par)
        if tuple(cv._params_key()[1:] for cv in convs) != ctx.param_keys:
            raise RuntimeError("ResStackFn.backward: a parameter of the unit was modified after the forward pass")
        d, slope = ctx.geom
        dy = _c(dy)
        _require_device(dy)
        b, c, t = x.shape
        need = ctx.needs_input_grad
        dh, dxp = ops.resstack_backward_data(dy, h, x, stack.unit_image_bwd(), d, slope)
        dx = None
        if need[0]:
            dx = torch.empty_like(x)
            _lib.check(_L().pwg_pad1d_backward(_ptr(dxp), _ptr(dx), b * c, t, d, d, ops.PAD["reflect"], _stream()),
                       "pad1d_backward")
        # weight path: the three layers' own kernels (dilated layer: on the explicitly padded input, as its un-fused
        # autograd node does)
        xp = torch.empty((b, c, t + 2 * d), device=x.device, dtype=torch.float32)
        _lib.check(_L().pwg_pad1d_forward(_ptr(x), _ptr(xp), b * c, t, d, d, ops.PAD["reflect"], _stream()), "pad1d_forward")
        descs = (ops.make_conv_desc(b, c, c, t + 2 * d, t, 3, dilation=d, pre_act="leaky_relu", pre_slope=slope),
                 ops.make_conv_desc(b, c, c, t, t, 1, pre_act="leaky_relu", pre_slope=slope),
                 ops.make_conv_desc(b, c, c, t, t, 1))
        grads = []
        pi = 2
        for cv, hd, dsc, xin, gsum, n in zip(convs, ctx.holders, descs, (xp, h, x), (dh, dy, dy), n_par):
            has_g = cv.has_weight_norm
            need_w = need[pi]
            need_g = has_g and need[pi + 1]
            need_b = cv.bias is not None and need[pi + 1 + int(has_g)]
            v = hd.w
            g = cv.weight_g.detach() if has_g else None
            dw, dg, db = conv_param_grads(dsc, xin, gsum, tuple(v.shape), tuple(cv.raw_weight.shape), v, g, need_w,
                                          need_g, need_b)
            grads.append(dw)
            if has_g:
                grads.append(dg)
            if cv.bias is not None:
                grads.append(db)
            pi += n
        return (dx, None) + tuple(grads)


class Add3DivFn(torch.autograd.Function):
    """((a + b) + c) / div with c optional -- the MRF combine when blocks run as parallel branches."""

    @staticmethod
    def forward(ctx, a, b, c, div):
        a, b = _c(a), _c(b)
        c = None if c is None else _c(c)
        _require_device(a, b, c)
        y = torch.empty_like(a)
        _lib.check(_L().pwg_add3_div(_ptr(a), _ptr(b), _ptr(c), _ptr(y), a.numel(), float(div), _stream()), "add3_div")
        ctx.div = float(div)
        ctx.has_c = c is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        g = torch.empty_like(dy)
        _lib.check(_L().pwg_act_backward(_ptr(dy), None, _ptr(g), dy.numel(), 0, 0.0, 1.0 / ctx.div, _stream()),
                   "act_backward")
        return g, g, (g if ctx.has_c else None), None


# ---------------------------------------------------------------------------------------------
# pooling / padding
# ---------------------------------------------------------------------------------------------
class AvgPool1dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kernel, stride, pad, count_include_pad):
        x = _c(x)
        _require_device(x)
        t_in = x.shape[-1]
        t_out = (t_in + 2 * pad - kernel) // stride + 1
        rows = x.numel() // t_in
        y = torch.empty(x.shape[:-1] + (t_out,), device=x.device, dtype=torch.float32)
        _lib.check(_L().pwg_avg_pool1d_forward(_ptr(x), _ptr(y), rows, t_in, t_out, kernel, stride, pad,
                                               int(bool(count_include_pad)), _stream()), "avg_pool1d_forward")
        ctx.cfg = (rows, t_in, t_out, kernel, stride, pad, int(bool(count_include_pad)), tuple(x.shape))
        return y

    @staticmethod
    def backward(ctx, dy):
        rows, t_in, t_out, kernel, stride, pad, cip, shape = ctx.cfg
        dy = _c(dy)
        dx = torch.empty(shape, device=dy.device, dtype=torch.float32)
        _lib.check(_L().pwg_avg_pool1d_backward(_ptr(dy), _ptr(dx), rows, t_in, t_out, kernel, stride, pad, cip,
                                                _stream()), "avg_pool1d_backward")
        return dx, None, None, None, None


class Pad1dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pad_left, pad_right, mode):
        x = _c(x)
        _require_device(x)
        t_in = x.shape[-1]
        rows = x.numel() // t_in
        y = torch.empty(x.shape[:-1] + (t_in + pad_left + pad_right,), device=x.device, dtype=torch.float32)
        _lib.check(_L().pwg_pad1d_forward(_ptr(x), _ptr(y), rows, t_in, pad_left, pad_right, ops.PAD[mode], _stream()),
                   "pad1d_forward")
        ctx.cfg = (rows, t_in, pad_left, pad_right, ops.PAD[mode], tuple(x.shape))
        return y

    @staticmethod
    def backward(ctx, dy):
        rows, t_in, pl, pr, mode, shape = ctx.cfg
        dy = _c(dy)
        dx = torch.empty(shape, device=dy.device, dtype=torch.float32)
        _lib.check(_L().pwg_pad1d_backward(_ptr(dy), _ptr(dx), rows, t_in, pl, pr, mode, _stream()), "pad1d_backward")
        return dx, None, None, None


# ---------------------------------------------------------------------------------------------
# Parallel WaveGAN element-wise stages
# ---------------------------------------------------------------------------------------------
class GateFn(torch.autograd.Function):
    """(B, 2C, T) -> (B, C, T): tanh(first half) * sigmoid(second half)."""

    @staticmethod
    def forward(ctx, z):
        z = _c(z)
        _require_device(z)
        b, c2, t = z.shape
        out = torch.empty(b, c2 // 2, t, device=z.device, dtype=torch.float32)
        _lib.check(_L().pwg_gate_forward(_ptr(z), _ptr(out), b, c2 // 2, t, _stream()), "gate_forward")
        ctx.save_for_backward(z)
        return out

    @staticmethod
    def backward(ctx, dout):
        (z,) = ctx.saved_tensors
        dout = _c(dout)
        b, c2, t = z.shape
        dz = torch.empty_like(z)
        _lib.check(_L().pwg_gate_backward(_ptr(z), _ptr(dout), _ptr(dz), b, c2 // 2, t, _stream()), "gate_backward")
        return dz


class StretchConvFn(torch.autograd.Function):
    """Nearest stretch by ``scale`` along time + (F, 2*scale+1) smoothing conv over (mel channel, time); x (B, C, T),
    w (..., F, k) with F = freq_axis_kernel_size (1 in the shipped recipes).  ``pad_left`` = scale (centred, default)
    or 2*scale (causal).  ``act`` (None / "leaky_relu" / "relu" / "tanh"): the stage's optional nonlinearity, applied in
    the kernel's epilogue; its gradient mask is taken from the stage output."""

    @staticmethod
    def forward(ctx, x, w, scale, pad_left=None, act=None, slope=0.0):
        x = _c(x)
        ctx.w_shape = tuple(w.shape)
        k = w.shape[-1]
        fk = w.numel() // k
        w = _c(w.reshape(-1))
        _require_device(x, w)
        t_in = x.shape[-1]
        rows = x.numel() // t_in
        channels = x.shape[-2] if x.dim() >= 2 else 1
        y = torch.empty(x.shape[:-1] + (t_in * scale,), device=x.device, dtype=torch.float32)
        pad_left = (k - 1) // 2 if pad_left is None else int(pad_left)
        _lib.check(_L().pwg_stretch_conv_forward(_ptr(x), _ptr(w), _ptr(y), rows, t_in, scale, k, pad_left, channels, fk,
                                                 ops.ACT[act], float(slope), _stream()), "stretch_conv_forward")
        if act is None:
            ctx.save_for_backward(x, w)
        else:
            ctx.save_for_backward(x, w, y)
        ctx.scale, ctx.pad_left, ctx.geom, ctx.act = scale, pad_left, (k, fk, channels), (act, float(slope))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors[:2]
        dy = _c(dy)
        if ctx.act[0] is not None:
            y, g = ctx.saved_tensors[2], torch.empty_like(dy)
            _lib.check(_L().pwg_act_backward(_ptr(dy), _ptr(y), _ptr(g), dy.numel(), ops.ACT[ctx.act[0]], ctx.act[1], 1.0,
                                             _stream()), "act_backward")
            dy = g
        k, fk, channels = ctx.geom
        t_in = x.shape[-1]
        rows = x.numel() // t_in
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        ws, ws_n = None, 0
        if dw is not None:
            ws_n = int(_L().pwg_stretch_conv_backward_workspace_floats(k, fk))
            ws = torch.empty(ws_n, device=x.device, dtype=torch.float32)
        _lib.check(_L().pwg_stretch_conv_backward(_ptr(dy), _ptr(x), _ptr(w), _ptr(dx), _ptr(dw), rows, t_in, ctx.scale,
                                                  k, ctx.pad_left, channels, fk, _ptr(ws), ws_n, _stream()),
                   "stretch_conv_backward")
        return dx, (None if dw is None else dw.reshape(ctx.w_shape)), None, None, None, None


# ---------------------------------------------------------------------------------------------
# pseudo-QMF filterbank: two polyphase kernels, each the adjoint of the other (csrc/pqmf.hip)
# ---------------------------------------------------------------------------------------------
def _pqmf_down(x, h, n_out, pad):
    b, t = x.shape[0], x.shape[-1]
    k, length = h.shape
    y = torch.empty(b, k, n_out, device=x.device, dtype=torch.float32)
    _lib.check(_L().pwg_pqmf_down(_ptr(x), _ptr(h), _ptr(y), b, t, n_out, k, length, pad, _stream()), "pqmf_down")
    return y


def _pqmf_up(y, g, t_out, pad):
    b, k, n = y.shape
    x = torch.empty(b, 1, t_out, device=y.device, dtype=torch.float32)
    _lib.check(_L().pwg_pqmf_up(_ptr(y), _ptr(g), _ptr(x), b, n, t_out, k, g.shape[1], pad, _stream()), "pqmf_up")
    return x


class PQMFDownFn(torch.autograd.Function):
    """y[b, k, i] = sum_j h[k, j] x[b, 0, i K + j - pad]; x (B, 1, T), h (K, L) a buffer (no filter gradient)."""

    @staticmethod
    def forward(ctx, x, h, n_out, pad):
        x, h = _c(x), _c(h)
        _require_device(x, h)
        ctx.save_for_backward(h)
        ctx.t, ctx.pad = x.shape[-1], pad
        return _pqmf_down(x, h, n_out, pad)

    @staticmethod
    def backward(ctx, dy):
        (h,) = ctx.saved_tensors
        return _pqmf_up(_c(dy), h, ctx.t, ctx.pad), None, None, None


class PQMFUpFn(torch.autograd.Function):
    """x[b, 0, t] = sum_k sum_i g[k, t + pad - i K] y[b, k, i]; y (B, K, n), g (K, L) a buffer."""

    @staticmethod
    def forward(ctx, y, g, t_out, pad):
        y, g = _c(y), _c(g)
        _require_device(y, g)
        ctx.save_for_backward(g)
        ctx.n, ctx.pad = y.shape[-1], pad
        return _pqmf_up(y, g, t_out, pad)

    @staticmethod
    def backward(ctx, dx):
        (g,) = ctx.saved_tensors
        return _pqmf_down(_c(dx), g, ctx.n, ctx.pad), None, None, None


# ---------------------------------------------------------------------------------------------
# StyleMelGAN element-wise stages (layers/tade_res_block.py of the reference)
# ---------------------------------------------------------------------------------------------
class InstanceNormFn(torch.autograd.Function):
    """torch.nn.InstanceNorm1d(affine=False): per (batch, channel) row statistics over time."""

    @staticmethod
    def forward(ctx, x, eps):
        x = _c(x)
        _require_device(x)
        t = x.shape[-1]
        rows = x.numel() // t
        y = torch.empty_like(x)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
        _lib.check(_L().pwg_instance_norm_forward(_ptr(x), _ptr(y), _ptr(mean), _ptr(rstd), rows, t, float(eps),
                                                  _stream()), "instance_norm_forward")
        ctx.save_for_backward(y, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, rstd = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(dy)
        t = y.shape[-1]
        _lib.check(_L().pwg_instance_norm_backward(_ptr(dy), _ptr(y), _ptr(rstd), _ptr(dx), y.numel() // t, t,
                                                   _stream()), "instance_norm_backward")
        return dx, None


class UpsampleNearestFn(torch.autograd.Function):
    """y = nearest_upsample(x, scale) (+ add): x (..., T) -> (..., T * scale)."""

    @staticmethod
    def forward(ctx, x, scale, add=None):
        x = _c(x)
        add = None if add is None else _c(add)
        _require_device(x, add)
        t = x.shape[-1]
        rows = x.numel() // t
        y = torch.empty(x.shape[:-1] + (t * scale,), device=x.device, dtype=torch.float32)
        _lib.check(_L().pwg_upsample_nearest_forward(_ptr(x), _ptr(add), _ptr(y), rows, t, int(scale), _stream()),
                   "upsample_nearest_forward")
        ctx.cfg = (rows, t, int(scale), tuple(x.shape), add is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        rows, t, scale, shape, has_add = ctx.cfg
        dy = _c(dy)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(shape, device=dy.device, dtype=torch.float32)
            _lib.check(_L().pwg_upsample_nearest_backward(_ptr(dy), _ptr(dx), rows, t, scale, _stream()),
                       "upsample_nearest_backward")
        return dx, None, (dy if has_add and ctx.needs_input_grad[2] else None)


class TadeModulateFn(torch.autograd.Function):
    """y = cg[:, :C] * nearest_upsample(xn, scale) + cg[:, C:]   (xn (B, C, T), cg (B, 2C, T * scale))."""

    @staticmethod
    def forward(ctx, xn, cg, scale):
        xn, cg = _c(xn), _c(cg)
        _require_device(xn, cg)
        b, c, t = xn.shape
        assert cg.shape == (b, 2 * c, t * scale), (tuple(cg.shape), (b, 2 * c, t * scale))
        y = torch.empty((b, c, t * scale), device=xn.device, dtype=torch.float32)
        _lib.check(_L().pwg_tade_modulate_forward(_ptr(xn), _ptr(cg), _ptr(y), b, c, t, int(scale), _stream()),
                   "tade_modulate_forward")
        ctx.save_for_backward(xn, cg)
        ctx.scale = int(scale)
        return y

    @staticmethod
    def backward(ctx, dy):
        xn, cg = ctx.saved_tensors
        dy = _c(dy)
        b, c, t = xn.shape
        dxn = torch.empty_like(xn) if ctx.needs_input_grad[0] else None
        dcg = torch.empty_like(cg) if ctx.needs_input_grad[1] else None
        _lib.check(_L().pwg_tade_modulate_backward(_ptr(dy), _ptr(xn), _ptr(cg), _ptr(dxn), _ptr(dcg), b, c, t,
                                                   ctx.scale, _stream()), "tade_modulate_backward")
        return dxn, dcg, None


class SoftmaxGateFn(torch.autograd.Function):
    """(B, 2C, T) -> (B, C, T): softmax over channels (or sigmoid) of the first half * tanh of the second."""

    @staticmethod
    def forward(ctx, z, use_softmax):
        z = _c(z)
        _require_device(z)
        b, c2, t = z.shape
        y = torch.empty((b, c2 // 2, t), device=z.device, dtype=torch.float32)
        _lib.check(_L().pwg_softmax_gate_forward(_ptr(z), _ptr(y), b, c2 // 2, t, int(bool(use_softmax)), _stream()),
                   "softmax_gate_forward")
        ctx.save_for_backward(z)
        ctx.use_softmax = int(bool(use_softmax))
        return y

    @staticmethod
    def backward(ctx, dy):
        (z,) = ctx.saved_tensors
        dy = _c(dy)
        b, c2, t = z.shape
        dz = torch.empty_like(z)
        _lib.check(_L().pwg_softmax_gate_backward(_ptr(z), _ptr(dy), _ptr(dz), b, c2 // 2, t, ctx.use_softmax, _stream()),
                   "softmax_gate_backward")
        return dz, None


# ---------------------------------------------------------------------------------------------
# UHiFiGAN pieces
# ---------------------------------------------------------------------------------------------
class ConcatChannelsFn(torch.autograd.Function):
    """torch.cat((a, b), dim=1) for (B, C, T) tensors as two strided copies."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        _require_device(a, b)
        bsz, ca, t = a.shape
        cb = b.shape[1]
        assert b.shape[0] == bsz and b.shape[2] == t, (tuple(a.shape), tuple(b.shape))
        y = torch.empty((bsz, ca + cb, t), device=a.device, dtype=torch.float32)
        for src, off in ((a, 0), (b, ca)):
            _lib.check(_L().pwg_copy_channels(_ptr(src), _ptr(y), bsz, src.shape[1], ca + cb, off, t, 0, _stream()),
                       "copy_channels")
        ctx.cfg = (bsz, ca, cb, t)
        return y

    @staticmethod
    def backward(ctx, dy):
        bsz, ca, cb, t = ctx.cfg
        dy = _c(dy)
        outs = []
        for i, (c, off) in enumerate(((ca, 0), (cb, ca))):
            if not ctx.needs_input_grad[i]:
                outs.append(None)
                continue
            d = torch.empty((bsz, c, t), device=dy.device, dtype=torch.float32)
            _lib.check(_L().pwg_copy_channels(_ptr(d), _ptr(dy), bsz, c, ca + cb, off, t, 1, _stream()), "copy_channels")
            outs.append(d)
        return tuple(outs)


class DropoutFn(torch.autograd.Function):
    """Training-mode dropout with a counter-based mask (regenerated from the seeds in backward).
    ``seed_dev``: optional int64 device scalar added to the host seed inside the kernel, so that a
    captured hipGraph draws a fresh mask at every replay."""

    @staticmethod
    def forward(ctx, x, p, seed, seed_dev=None):
        x = _c(x)
        _require_device(x)
        y = torch.empty_like(x)
        sp = None if seed_dev is None else ctypes.c_void_p(seed_dev.data_ptr())
        _lib.check(_L().pwg_dropout(_ptr(x), _ptr(y), x.numel(), float(p), int(seed), sp, _stream()), "dropout")
        ctx.cfg = (float(p), int(seed))
        ctx.seed_dev = seed_dev
        return y

    @staticmethod
    def backward(ctx, dy):
        p, seed = ctx.cfg
        dy = _c(dy)
        dx = torch.empty_like(dy)
        sp = None if ctx.seed_dev is None else ctypes.c_void_p(ctx.seed_dev.data_ptr())
        _lib.check(_L().pwg_dropout(_ptr(dy), _ptr(dx), dy.numel(), p, seed, sp, _stream()), "dropout")
        return dx, None, None, None


# ---------------------------------------------------------------------------------------------
# spectral-loss pieces
# ---------------------------------------------------------------------------------------------
class FrameFoldFn(torch.autograd.Function):
    """(B, T) -> (B, hop, n_cols): y[b][c][n] = reflect_pad(x, pad)[b][n*hop + c] (zero past the end)."""

    @staticmethod
    def forward(ctx, x, pad, hop, n_cols):
        x = _c(x)
        _require_device(x)
        b, t = x.shape
        y = torch.empty(b, hop, n_cols, device=x.device, dtype=torch.float32)
        _lib.check(_L().pwg_frame_fold_forward(_ptr(x), _ptr(y), b, t, pad, hop, n_cols, _stream()), "frame_fold_forward")
        ctx.cfg = (b, t, pad, hop, n_cols)
        return y

    @staticmethod
    def backward(ctx, dy):
        b, t, pad, hop, n_cols = ctx.cfg
        dy = _c(dy)
        dx = torch.empty(b, t, device=dy.device, dtype=torch.float32)
        _lib.check(_L().pwg_frame_fold_backward(_ptr(dy), _ptr(dx), b, t, pad, hop, n_cols, _stream()), "frame_fold_backward")
        return dx, None, None, None


class StftMagFn(torch.autograd.Function):
    """(B, 2*bins, frames) [re rows | im rows] -> (B, bins, frames): sqrt(max(re^2+im^2, eps))."""

    @staticmethod
    def forward(ctx, spec, eps):
        spec = _c(spec)
        _require_device(spec)
        b, two_bins, frames = spec.shape
        bins = two_bins // 2
        mag = torch.empty(b, bins, frames, device=spec.device, dtype=torch.float32)
        _lib.check(_L().pwg_stft_mag_forward(_ptr(spec), _ptr(mag), b, bins, frames, float(eps), _stream()), "stft_mag_forward")
        ctx.save_for_backward(spec, mag)
        ctx.eps = float(eps)
        return mag

    @staticmethod
    def backward(ctx, dmag):
        spec, mag = ctx.saved_tensors
        dmag = _c(dmag)
        b, bins, frames = mag.shape
        dspec = torch.empty_like(spec)
        _lib.check(_L().pwg_stft_mag_backward(_ptr(spec), _ptr(mag), _ptr(dmag), _ptr(dspec), b, bins, frames, ctx.eps,
                                              _stream()), "stft_mag_backward")
        return dspec, None


class LogClampFn(torch.autograd.Function):
    """y = log(max(x, eps)) / log_div."""

    @staticmethod
    def forward(ctx, x, eps, log_div):
        x = _c(x)
        _require_device(x)
        y = torch.empty_like(x)
        _lib.check(_L().pwg_log_clamp_forward(_ptr(x), _ptr(y), x.numel(), float(eps), float(log_div), _stream()), "log_clamp_forward")
        ctx.save_for_backward(x)
        ctx.cfg = (float(eps), float(log_div))
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(x)
        _lib.check(_L().pwg_log_clamp_backward(_ptr(x), _ptr(dy), _ptr(dx), x.numel(), ctx.cfg[0], ctx.cfg[1], _stream()), "log_clamp_backward")
        return dx, None, None


# ---------------------------------------------------------------------------------------------
# loss reductions
# ---------------------------------------------------------------------------------------------
RED = {"abs_diff": 0, "sq_diff": 1, "sq": 2, "sq_diff_const": 3, "sum": 4, "hinge_real": 5, "hinge_fake": 6,
       "abs_diff_lrelu": 7}  # (7: |lrelu(a) - lrelu(b)|, slope = const: feature maps in pre-activation form)


class ReduceFn(torch.autograd.Function):
    """0-dim tensor  scale * sum_i term(a_i, b_i | c)  with a deterministic two-stage HIP reduction."""

    @staticmethod
    def forward(ctx, a, b, mode, scale, const):
        a = _c(a)
        _require_device(a)
        if b is not None:
            b = _c(b)
            _require_device(b)
            assert b.shape == a.shape, (a.shape, b.shape)
        out = torch.empty(1, device=a.device, dtype=torch.float32)
        ws = torch.empty(512, device=a.device, dtype=torch.float32)
        _lib.check(_L().pwg_reduce_forward(_ptr(a), _ptr(b), float(const), a.numel(), RED[mode], float(scale), _ptr(out),
                                           _ptr(ws), _stream()), "reduce_forward")
        ctx.save_for_backward(a, b)
        ctx.cfg = (RED[mode], float(scale), float(const))
        return out.reshape(())

    @staticmethod
    def backward(ctx, gout):
        a, b = ctx.saved_tensors
        mode, scale, const = ctx.cfg
        gout = _c(gout).reshape(1)
        need_a, need_b = ctx.needs_input_grad[0], (b is not None and ctx.needs_input_grad[1])
        da = torch.empty_like(a) if need_a else None
        db = torch.empty_like(a) if need_b else None
        if need_a or need_b:
            _lib.check(_L().pwg_reduce_backward(_ptr(a), _ptr(b), const, a.numel(), mode, scale, _ptr(gout), _ptr(da),
                                                _ptr(db), _stream()), "reduce_backward")
        return da, db, None, None, None


class MultiReduceFn(torch.autograd.Function):
    """(n_slots,) tensor  out[slot] = sum_items scale * sum_i term_mode(a_i, b_i | const)  over MANY tensors
    with one launch pair (``pwg_multi_reduce_*``): the loops over discriminators x layers of the
    feature-matching and adversarial losses as a single multi-tensor reduction.

    ``spec``: list of ``(mode, scale, const, slot)`` per item; ``tensors``: a_0, b_0, a_1, b_1, ...
    (``b_i`` None for the unary modes).  Deterministic: per-chunk partial sums, summed in item order."""

    @staticmethod
    def _items(spec, ops_, grads=None):
        arr = (_lib.RedItem * len(spec))()
        for i, (mode, scale, const, slot) in enumerate(spec):
            a, b = ops_[2 * i], ops_[2 * i + 1]
            it = arr[i]
            it.a, it.b = a.data_ptr(), (None if b is None else b.data_ptr())
            if grads is not None:
                da, db = grads[2 * i], grads[2 * i + 1]
                it.da, it.db = (None if da is None else da.data_ptr()), (None if db is None else db.data_ptr())
            it.n, it.mode, it.slot, it.scale, it.c = a.numel(), RED[mode], int(slot), float(scale), float(const)
        return arr

    @staticmethod
    def forward(ctx, spec, n_slots, *tensors):
        assert len(tensors) == 2 * len(spec) and len(spec) > 0
        ts = [None if t is None else _c(t) for t in tensors]
        _require_device(*ts)
        for i in range(len(spec)):
            a, b = ts[2 * i], ts[2 * i + 1]
            assert b is None or b.shape == a.shape, (tuple(a.shape), tuple(b.shape))
        dev = ts[0].device
        out = torch.empty(n_slots, device=dev, dtype=torch.float32)
        for first in range(0, len(spec), _lib.RED_MAX_ITEMS):
            sub = spec[first:first + _lib.RED_MAX_ITEMS]
            items = MultiReduceFn._items(sub, ts[2 * first:2 * (first + len(sub))])
            ws = torch.empty(max(1, _L().pwg_multi_reduce_workspace_floats(items, len(sub))), device=dev,
                             dtype=torch.float32)
            _lib.check(_L().pwg_multi_reduce_forward(items, len(sub), n_slots, _ptr(out), int(first > 0), _ptr(ws),
                                                     _stream()), "multi_reduce_forward")
        ctx.spec, ctx.n_slots = spec, n_slots
        ctx.save_for_backward(*ts)
        return out

    @staticmethod
    def backward(ctx, gout):
        ts = ctx.saved_tensors
        spec = ctx.spec
        gout = _c(gout)
        grads = []
        for i in range(len(spec)):
            a, b = ts[2 * i], ts[2 * i + 1]
            grads.append(torch.empty_like(a) if ctx.needs_input_grad[2 + 2 * i] else None)
            grads.append(torch.empty_like(b) if (b is not None and ctx.needs_input_grad[3 + 2 * i]) else None)
        for first in range(0, len(spec), _lib.RED_MAX_ITEMS):
            sub = spec[first:first + _lib.RED_MAX_ITEMS]
            sl = slice(2 * first, 2 * (first + len(sub)))
            if all(g is None for g in grads[sl]):
                continue
            items = MultiReduceFn._items(sub, ts[sl], grads[sl])
            _lib.check(_L().pwg_multi_reduce_backward(items, len(sub), ctx.n_slots, _ptr(gout), _stream()),
                       "multi_reduce_backward")
        return (None, None) + tuple(grads)


def l1_mean(a, b):
    """F.l1_loss(a, b) (mean)."""
    return ReduceFn.apply(a, b, "abs_diff", 1.0 / a.numel(), 0.0)


def mse_to_const_mean(a, const):
    """F.mse_loss(a, full_like(a, const)) (mean)."""
    return ReduceFn.apply(a, None, "sq_diff_const", 1.0 / a.numel(), float(const))


def sq_diff_sum(a, b):
    return ReduceFn.apply(a, b, "sq_diff", 1.0, 0.0)


def sq_sum(a):
    return ReduceFn.apply(a, None, "sq", 1.0, 0.0)


def avg_pool1d(x, kernel, stride, pad, count_include_pad=True):
    return AvgPool1dFn.apply(x, kernel, stride, pad, count_include_pad)


def pad1d(x, pad_left, pad_right, mode="reflect"):
    return Pad1dFn.apply(x, pad_left, pad_right, mode)
