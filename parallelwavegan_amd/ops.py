"""Tensor-level wrappers over the C ABI (device pointers in, device pointers out).

PyTorch is used here only as the owner of device memory and streams.
"""
import ctypes

import torch

from . import _lib
from ._lib import ConvDesc, ResUnitDesc, WaveNetDesc

ACT = {None: _lib.PWG_ACT_NONE, "none": _lib.PWG_ACT_NONE, "leaky_relu": _lib.PWG_ACT_LEAKY_RELU,
       "tanh": _lib.PWG_ACT_TANH, "relu": _lib.PWG_ACT_RELU}
PAD = {"zero": _lib.PWG_PAD_ZERO, "reflect": _lib.PWG_PAD_REFLECT, "replicate": _lib.PWG_PAD_REPLICATE}


# Parameters updated through raw pointers (fused optimizer kernels) do not bump torch's version
# counters; every such update bumps this epoch instead, and the packed-weight caches key on it.
PARAM_EPOCH = [0]


def bump_param_epoch():
    """Invalidate every cached weight image (e.g. before a hipGraph capture, so that all weight
    preparation kernels of the captured step are recorded inside the graph)."""
    PARAM_EPOCH[0] += 1


def param_epoch(p):
    return getattr(p, "_pwg_epoch", 0)


def tensor_version(t):
    """torch's version counter of ``t``.  Tensors created or loaded under ``torch.inference_mode()`` track none (reading
    it raises) and cannot be written in place outside inference mode either: a constant stands in."""
    return -1 if t.is_inference() else t._version


def bump_params(params):
    """Mark parameters as changed through raw pointers (fused optimizer kernels)."""
    for p in params:
        p._pwg_epoch = getattr(p, "_pwg_epoch", 0) + 1


# Data-parallel gradient slots (distributed.GradReducer): parameter storage address -> [bucket view, weak reference to
# the owner, claim epoch] (the owner drops its entries in remove() and, through weakref.finalize, when it is collected).
# Contract: like ``.grad`` itself, a slot ACCUMULATES every contribution between two ``prepare()`` calls -- a second
# backward() over a retained graph adds to it, as AccumulateGrad would -- and a node writes its parameters' slots
# whenever it runs, so the ``inputs=``-restricted backward calls of one pass must cover disjoint sub-networks (the
# trainer's exchange groups are the independent sub-discriminators).
# A weight-gradient launch whose layer's parameter has a free slot writes its result straight into the bucket:
# autograd's AccumulateGrad adopts the returned alias as ``p.grad`` without a copy and the reducer's hook finds nothing
# left to copy.  A slot is claimed once per backward pass (the owner's epoch); later contributions of the same pass --
# the discriminator phase differentiates D(y) and D(G(c)) in one pass -- are added into the slot by their own node.
GRAD_SLOTS = {}


def claim_grad_slots(keys_shapes):
    """Claim the gradient slots of one layer's parameters together.  ``keys_shapes``: [(parameter address, shape)].
    Returns ``(aliases, accumulate)`` or ``(None, False)`` when any of them has no slot (then nothing is claimed):
    the FIRST claim of a backward pass gets ``accumulate = False`` -- the caller's kernels write the slots and the
    aliases go back to autograd as the gradients; a later claim of the same pass gets ``accumulate = True`` -- the
    caller adds its contribution INTO the slots (same stream, program order) and returns None to autograd, which then
    neither sums nor copies anything."""
    entries = []
    for key, shape in keys_shapes:
        e = GRAD_SLOTS.get(key)
        owner = e[1]() if e is not None else None  # (weak: a dropped reducer must not be kept alive by this table)
        if e is not None and owner is None:
            del GRAD_SLOTS[key]
        if owner is None or not owner.enabled or not owner.direct_slots or e[0].numel() != _numel(shape):
            return None, False
        entries.append((e, shape, owner))
    states = {e[2] == owner.epoch for e, _, owner in entries}
    if len(states) != 1:  # (cannot happen while a layer's parameters live in one reducer; stay on the copying path)
        return None, False
    later = states.pop()
    cur = torch.cuda.current_stream(entries[0][0][0].device) if entries[0][0][0].is_cuda else None
    for e, _, owner in entries:
        e[2] = owner.epoch
        if not later:  # remember where this pass's first contribution is written (see slot_add)
            if len(e) > 3:
                e[3] = cur
            else:
                e.append(cur)
    if later:
        first = entries[0][0][3] if len(entries[0][0]) > 3 else None
        later = first if first is not None else True
    return [e[0].view(shape) for e, shape, _ in entries], later


def slot_add(slot_view, contribution, later):
    """Add a LATER contribution of one backward pass into a gradient slot.  ``later`` is what :func:`claim_grad_slots`
    returned: True, or the stream the pass's FIRST contribution was written on.  The two passes of a layer normally run
    on one stream (program order).  They do not when one of them is the un-forked re-evaluation of a stateful
    sub-discriminator (``only=[...]`` with a single branch: caller's stream) and the other one the forked full pass (side
    stream): an ``add_`` on the caller's stream was then unordered against the first write -- and against the reducer
    hook's event, which autograd places on the first write's stream (round 6: the bias gradients of HiFi-GAN's spectrally
    normalised scale discriminator; found through the eager fork, profiles/r06_eager_nan_bisect.txt; the captured graph
    had the same two unordered nodes).  The addition is issued on the FIRST stream, after this stream's producer."""
    t = contribution.reshape(slot_view.shape)
    if later is True or not t.is_cuda:
        slot_view.add_(t)
        return
    cur = torch.cuda.current_stream(t.device)
    if later == cur:
        slot_view.add_(t)
        return
    done = torch.cuda.Event()
    done.record(cur)
    later.wait_event(done)
    with torch.cuda.stream(later):
        slot_view.add_(t)
    t.record_stream(later)


def _numel(shape):
    n = 1
    for d in shape:
        n *= int(d)
    return n


def _require_device(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "parallelwavegan_amd ops run only on an MI355X device tensor (got a CPU tensor); "
                "there is no CPU fallback path"
            )
        if t.dtype != torch.float32:
            raise RuntimeError(f"expected float32 tensor, got {t.dtype}")
        if not t.is_contiguous():
            raise RuntimeError("expected contiguous tensor")


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def make_conv_desc(batch, c_in, c_out, t_in, t_out, kernel, stride=1, dilation=1, pad_left=0, groups=1,
                   transposed=False, width=1, pad_mode="zero", pre_act=None, pre_slope=0.0, post_act=None,
                   post_slope=0.0, out_mul=1.0, out_div=1.0):
    return ConvDesc(batch, c_in, c_out, t_in, t_out, width, kernel, stride, dilation, pad_left, groups,
                    int(bool(transposed)), PAD[pad_mode], ACT[pre_act], float(pre_slope), ACT[post_act],
                    float(post_slope), float(out_mul), float(out_div))


def conv_out_length(t_in, kernel, stride=1, dilation=1, pad_left=0, pad_right=0):
    return (t_in + pad_left + pad_right - dilation * (kernel - 1) - 1) // stride + 1


def conv_transpose_out_length(t_in, kernel, stride, padding, output_padding):
    return (t_in - 1) * stride - 2 * padding + kernel + output_padding


def packed_weight_floats(desc):
    n = _lib.lib().pwg_conv1d_packed_weight_floats(ctypes.byref(desc))
    if n == 0:
        _lib.check(-1, "packed_weight_floats")
    return n


def pack_weight(desc, w, scale=None):
    """torch-layout weight (+ optional weight_norm row scale) -> packed image."""
    _require_device(w, scale)
    out = torch.empty(packed_weight_floats(desc), device=w.device, dtype=torch.float32)
    _lib.check(_lib.lib().pwg_conv1d_pack_weight(ctypes.byref(desc), _ptr(w), _ptr(scale), _ptr(out), _stream()),
               "conv1d_pack_weight")
    return out


def _workspace(n_floats, device):
    """Scratch for a split reduction (caller-owned per the C ABI: torch's caching allocator here)."""
    if not n_floats:
        return None, 0
    return torch.empty(n_floats, device=device, dtype=torch.float32), n_floats


def conv1d_forward(desc, x, w_packed, bias=None, add1=None, add2=None, out=None):
    _require_device(x, w_packed, bias, add1, add2, out)
    if out is None:
        out = torch.empty((desc.batch, desc.c_out, desc.t_out * desc.width), device=x.device, dtype=torch.float32)
    assert x.numel() == desc.batch * desc.c_in * desc.t_in * desc.width, (tuple(x.shape), desc.batch, desc.c_in, desc.t_in)
    assert out.numel() == desc.batch * desc.c_out * desc.t_out * desc.width
    ws, ws_n = _workspace(_lib.lib().pwg_conv1d_forward_workspace_floats(ctypes.byref(desc)), x.device)
    _lib.check(_lib.lib().pwg_conv1d_forward(ctypes.byref(desc), _ptr(x), _ptr(w_packed), _ptr(bias), _ptr(add1),
                                             _ptr(add2), _ptr(out), _ptr(ws), ws_n, _stream()), "conv1d_forward")
    return out


def make_resunit_desc(batch, channels, t, kernel, dilation, has_conv2=True, slope1=0.1, slope2=0.1, out_div=1.0):
    return ResUnitDesc(int(batch), int(channels), int(t), int(kernel), int(dilation), int(bool(has_conv2)),
                       float(slope1), float(slope2), float(out_div))


def resunit_supported(desc):
    """Does the one-launch residual unit (csrc/resunit.hip) cover this geometry?"""
    return bool(_lib.lib().pwg_resunit_supported(ctypes.byref(desc)))


def resunit_profitable(desc):
    """Supported AND measured faster than the separate convolutions (heuristic lives with the kernel)."""
    return bool(_lib.lib().pwg_resunit_profitable(ctypes.byref(desc)))


def resunit_pack_weight(w, scale=None):
    """torch-layout (C, C, k) weight (+ optional weight-norm row scale) -> MFMA A-operand image."""
    _require_device(w, scale)
    c, k = w.shape[0], w.shape[2]
    assert w.shape[1] == c
    out = torch.empty(_lib.lib().pwg_resunit_packed_weight_floats(c, k), device=w.device, dtype=torch.float32)
    _lib.check(_lib.lib().pwg_resunit_pack_weight(c, k, _ptr(w), _ptr(scale), _ptr(out), _stream()),
               "resunit_pack_weight")
    return out


def resunit_forward(desc, x, w1_packed, b1, w2_packed=None, b2=None, add2=None, out=None):
    """One MRF residual unit: ``(x + conv2(lrelu(conv1(lrelu(x)) + b1)) + b2 [+ add2]) [/ out_div]``."""
    _require_device(x, w1_packed, b1, w2_packed, b2, add2, out)
    assert x.numel() == desc.batch * desc.channels * desc.t
    if out is None:
        out = torch.empty_like(x)
    _lib.check(_lib.lib().pwg_resunit_forward(ctypes.byref(desc), _ptr(x), _ptr(w1_packed), _ptr(b1), _ptr(w2_packed),
                                              _ptr(b2), _ptr(add2), _ptr(out), _stream()), "resunit_forward")
    return out


def resstack_supported(channels, t, dilation):
    """Does the one-launch MelGAN residual stack (csrc/resstack.hip) cover this geometry?"""
    return bool(_lib.lib().pwg_resstack_supported(int(channels), int(t), int(dilation)))


def resstack_pack_weight(w1, s1, w2, s2, ws, ss):
    """(C, C, 3), (C, C, 1), (C, C, 1) weights (+ optional weight-norm row scales) -> the unit's MFMA A-operand image."""
    _require_device(w1, s1, w2, s2, ws, ss)
    c = w1.shape[0]
    n = _lib.lib().pwg_resstack_packed_weight_floats(c)
    if n == 0:
        _lib.check(-1, "resstack_packed_weight_floats")
    out = torch.empty(n, device=w1.device, dtype=torch.float32)
    _lib.check(_lib.lib().pwg_resstack_pack_weight(c, _ptr(w1), _ptr(s1), _ptr(w2), _ptr(s2), _ptr(ws), _ptr(ss), _ptr(out),
                                                   _stream()), "resstack_pack_weight")
    return out


def resstack_forward(x, w_packed, dilation, slope, b1=None, b2=None, bs=None, save_h=False):
    """y (and h, the dilated convolution's pre-activation output, when ``save_h``) of one residual stack."""
    _require_device(x, w_packed, b1, b2, bs)
    b, c, t = x.shape
    y = torch.empty_like(x)
    h = torch.empty_like(x) if save_h else None
    _lib.check(_lib.lib().pwg_resstack_forward(b, c, t, int(dilation), float(slope), _ptr(x), _ptr(w_packed), _ptr(b1), _ptr(b2),
                                               _ptr(bs), _ptr(y), _ptr(h), _stream()), "resstack_forward")
    return y, h


def resstack_pack_weight_bwd(w1, s1, w2, s2, ws, ss):
    """The unit's weights re-laid for its data gradient (transposed; weight-norm scales on the reduction index)."""
    _require_device(w1, s1, w2, s2, ws, ss)
    c = w1.shape[0]
    n = _lib.lib().pwg_resstack_packed_weight_floats(c)
    if n == 0:
        _lib.check(-1, "resstack_packed_weight_floats")
    out = torch.empty(n, device=w1.device, dtype=torch.float32)
    _lib.check(_lib.lib().pwg_resstack_pack_weight_bwd(c, _ptr(w1), _ptr(s1), _ptr(w2), _ptr(s2), _ptr(ws), _ptr(ss),
                                                       _ptr(out), _stream()), "resstack_pack_weight_bwd")
    return out


def resstack_backward_data(dy, h, x, w_packed_bwd, dilation, slope):
    """(dh, dxp): gradient w.r.t. the dilated convolution's output and w.r.t. its reflect-padded input (+ the skip
    branch's contribution at the interior positions), shapes (B, C, T) and (B, C, T + 2 * dilation)."""
    _require_device(dy, h, x, w_packed_bwd)
    b, c, t = x.shape
    dh = torch.empty_like(x)
    dxp = torch.empty((b, c, t + 2 * int(dilation)), device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().pwg_resstack_backward_data(b, c, t, int(dilation), float(slope), _ptr(dy), _ptr(h), _ptr(x),
                                                     _ptr(w_packed_bwd), _ptr(dh), _ptr(dxp), _stream()),
               "resstack_backward_data")
    return dh, dxp


def make_wavenet_desc(batch, t, dilation, residual_channels=64, gate_channels=128, skip_channels=64, aux_channels=80,
                      kernel=3, causal=False, out_mul=1.0, skip_mul=1.0):
    return WaveNetDesc(int(batch), int(t), int(residual_channels), int(gate_channels), int(skip_channels),
                       int(aux_channels), int(kernel), int(dilation), int(bool(causal)), float(out_mul), float(skip_mul))


def wavenet_layer_supported(desc):
    """Does the one-launch WaveNet layer (csrc/wavenet.hip) cover this geometry?"""
    return bool(_lib.lib().pwg_wavenet_layer_supported(ctypes.byref(desc)))


def wavenet_pack_weights(desc, w_dil, s_dil, w_aux, s_aux, w_skip, s_skip, w_out, s_out):
    """The four torch-layout weights of a layer (+ optional weight-norm row scales) -> one MFMA A-operand image."""
    _require_device(w_dil, s_dil, w_aux, s_aux, w_skip, s_skip, w_out, s_out)
    out = torch.empty(_lib.lib().pwg_wavenet_packed_weight_floats(ctypes.byref(desc)), device=w_dil.device,
                      dtype=torch.float32)
    _lib.check(_lib.lib().pwg_wavenet_pack_weights(ctypes.byref(desc), _ptr(w_dil), _ptr(s_dil), _ptr(w_aux), _ptr(s_aux),
                                                   _ptr(w_skip), _ptr(s_skip), _ptr(w_out), _ptr(s_out), _ptr(out),
                                                   _stream()), "wavenet_pack_weights")
    return out


def wavenet_layer_forward(desc, x, c, skips, packed, b_dil, b_skip, b_out, save=False, skips_out=None):
    """One gated residual layer in one launch -> (x_out, skips_out, z, g); ``z`` / ``g`` only with ``save``
    (training: the backward pass needs them).  ``skips_out`` may be ``skips`` itself (in-place running sum)."""
    _require_device(x, c, skips, packed, b_dil, b_skip, b_out, skips_out)
    x_out = torch.empty_like(x)
    if skips_out is None:
        skips_out = torch.empty_like(x)
    z = torch.empty((x.shape[0], desc.gate_channels, x.shape[2]), device=x.device, dtype=torch.float32) if save else None
    g = torch.empty_like(x) if save else None
    _lib.check(_lib.lib().pwg_wavenet_layer_forward(ctypes.byref(desc), _ptr(x), _ptr(c), _ptr(skips), _ptr(packed),
                                                    _ptr(b_dil), _ptr(b_skip), _ptr(b_out), _ptr(x_out), _ptr(skips_out),
                                                    _ptr(z), _ptr(g), _stream()), "wavenet_layer_forward")
    return x_out, skips_out, z, g


def wavenet_pack_weights_bwd(desc, w_dil, s_dil, w_aux, s_aux, w_skip, s_skip, w_out, s_out):
    """Backward-pass image of a layer (gate / dilated / aux data gradients; folds desc.out_mul and desc.skip_mul)."""
    _require_device(w_dil, s_dil, w_aux, s_aux, w_skip, s_skip, w_out, s_out)
    out = torch.empty(_lib.lib().pwg_wavenet_packed_weight_bwd_floats(ctypes.byref(desc)), device=w_dil.device,
                      dtype=torch.float32)
    _lib.check(_lib.lib().pwg_wavenet_pack_weights_bwd(ctypes.byref(desc), _ptr(w_dil), _ptr(s_dil), _ptr(w_aux), _ptr(s_aux),
                                                       _ptr(w_skip), _ptr(s_skip), _ptr(w_out), _ptr(s_out), _ptr(out),
                                                       _stream()), "wavenet_pack_weights_bwd")
    return out


def wavenet_gate_backward(desc, z, dx_out, ds_out, packed_bwd):
    """(dz, go): gradient w.r.t. the gate input and the scaled residual-path gradient ``out_mul * dx_out`` (None
    when ``dx_out`` is None)."""
    _require_device(z, dx_out, ds_out, packed_bwd)
    dz = torch.empty_like(z)
    go = None if dx_out is None else torch.empty_like(dx_out)
    _lib.check(_lib.lib().pwg_wavenet_gate_backward(ctypes.byref(desc), _ptr(z), _ptr(dx_out), _ptr(ds_out),
                                                    _ptr(packed_bwd), _ptr(dz), _ptr(go), _stream()), "wavenet_gate_backward")
    return dz, go


def wavenet_data_backward(desc, dz, go, packed_bwd, need_dx=True, need_dc=True, dc_accum=None):
    """(dx, dc) = data gradients of the dilated convolution (+ go) and of the aux 1x1 convolution (+ ``dc_accum``:
    what the later layers already accumulated for the shared aux features)."""
    _require_device(dz, go, packed_bwd, dc_accum)
    b, t = dz.shape[0], dz.shape[2]
    dx = torch.empty((b, desc.residual_channels, t), device=dz.device, dtype=torch.float32) if need_dx else None
    dc = torch.empty((b, desc.aux_channels, t), device=dz.device, dtype=torch.float32) if need_dc else None
    if dx is None and dc is None:
        return None, None
    if dc_accum is not None and (tuple(dc_accum.shape) != (b, desc.aux_channels, t) or not dc_accum.is_contiguous()):
        raise ValueError("wavenet_data_backward: dc_accum must be a contiguous (B, aux, T) tensor")
    _lib.check(_lib.lib().pwg_wavenet_data_backward(ctypes.byref(desc), _ptr(dz), _ptr(go), _ptr(packed_bwd), _ptr(dc_accum),
                                                    _ptr(dx), _ptr(dc), _stream()), "wavenet_data_backward")
    return dx, dc


def wavenet_weight_backward(desc, dz, x, c, gs, go, g, convs=None):
    """Parameter gradients of the layer's four convolutions (csrc/wavenet.hip: two contraction launches + one finish
    launch).  ``convs`` = four ``(v, g, has_bias)`` entries (dilated, aux, skip, out): ``v``/``g`` the weight-norm
    tensors or None for a plain weight.  Returns four ``(dw_or_dv, dg, db)`` tuples in torch layouts
    ((128,64,3), (128,aux,1), (64,64,1), (64,64,1)); the last is (None, None, None) when ``go`` is None (the
    layer's residual output is unused)."""
    if convs is None:
        convs = ((None, None, True), (None, None, False), (None, None, True), (None, None, True))
    _require_device(dz, x, c, gs, go, g, *[t for cv in convs for t in cv[:2]])
    dev = dz.device
    r, gch, sch, aux = desc.residual_channels, desc.gate_channels, desc.skip_channels, desc.aux_channels
    shapes = ((gch, r, desc.kernel), (gch, aux, 1), (sch, r, 1), (r, r, 1))
    arr = (_lib.WaveNetParamGrad * 4)()
    outs = []
    for i, ((v, gg, has_b), shp) in enumerate(zip(convs, shapes)):
        if i == 3 and go is None:
            outs.append((None, None, None))
            continue
        if v is not None and (tuple(v.shape) != shp or gg is None or gg.numel() != shp[0] or not v.is_contiguous()
                              or not gg.is_contiguous()):
            raise ValueError(f"wavenet_weight_backward: conv {i}: weight-norm tensors do not match {shp}")
        dw = torch.empty(shp, device=dev, dtype=torch.float32)
        dg = torch.empty_like(gg) if v is not None else None
        db = torch.empty(shp[0], device=dev, dtype=torch.float32) if has_b else None
        arr[i].v, arr[i].g, arr[i].dw, arr[i].dg, arr[i].db = _ptr(v), _ptr(gg), _ptr(dw), _ptr(dg), _ptr(db)
        outs.append((dw, dg, db))
    n = _lib.lib().pwg_wavenet_weight_backward_workspace_floats(ctypes.byref(desc))
    if n == 0:
        _lib.check(-1, "wavenet_weight_backward_workspace_floats")
    ws = torch.empty(n, device=dev, dtype=torch.float32)
    _lib.check(_lib.lib().pwg_wavenet_weight_backward(ctypes.byref(desc), _ptr(dz), _ptr(x), _ptr(c), _ptr(gs), _ptr(go),
                                                      _ptr(g), arr, _ptr(ws), n, _stream()), "wavenet_weight_backward")
    return outs


def pack_weight_bwd(desc, w, scale=None):
    """Weight image for the data-gradient direction of ``desc`` (forward descriptor)."""
    _require_device(w, scale)
    n = _lib.lib().pwg_conv1d_packed_weight_bwd_floats(ctypes.byref(desc))
    if n == 0:
        _lib.check(-1, "packed_weight_bwd_floats")
    out = torch.empty(n, device=w.device, dtype=torch.float32)
    _lib.check(_lib.lib().pwg_conv1d_pack_weight_bwd(ctypes.byref(desc), _ptr(w), _ptr(scale), _ptr(out), _stream()),
               "conv1d_pack_weight_bwd")
    return out


def conv1d_backward_data(desc, dy, w_packed_bwd, x=None, accum=None, out=None):
    """dx = pre_act'(x) * data_grad(dy) (+ accum); ``desc`` is the forward descriptor."""
    _require_device(dy, w_packed_bwd, x, accum, out)
    if out is None:
        out = torch.empty((desc.batch, desc.c_in, desc.t_in * desc.width), device=dy.device, dtype=torch.float32)
    ws, ws_n = _workspace(_lib.lib().pwg_conv1d_backward_data_workspace_floats(ctypes.byref(desc)), dy.device)
    _lib.check(_lib.lib().pwg_conv1d_backward_data(ctypes.byref(desc), _ptr(dy), _ptr(w_packed_bwd), _ptr(x),
                                                   _ptr(accum), _ptr(out), _ptr(ws), ws_n, _stream()),
               "conv1d_backward_data")
    return out


def conv1d_backward_weight(desc, x, dy, weight_shape, need_dw=True, need_db=True, out_dw=None, out_db=None):
    """(dw in torch layout, db); deterministic two-stage reduction over (batch, time) slices.  ``out_dw`` / ``out_db``:
    optional destinations (contiguous, the right number of elements), e.g. data-parallel bucket slots."""
    _require_device(x, dy, out_dw, out_db)
    dw = (out_dw if out_dw is not None else torch.empty(weight_shape, device=x.device, dtype=torch.float32)) if need_dw else None
    db = (out_db if out_db is not None else torch.empty(desc.c_out, device=x.device, dtype=torch.float32)) if need_db else None
    ws, ws_n = None, 0
    if need_dw:
        ws_n = _lib.lib().pwg_conv1d_backward_weight_workspace_floats(ctypes.byref(desc))
        if ws_n:
            ws = torch.empty(ws_n, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().pwg_conv1d_backward_weight(ctypes.byref(desc), _ptr(x), _ptr(dy), _ptr(dw), _ptr(db),
                                                     _ptr(ws), ws_n, _stream()), "conv1d_backward_weight")
    return dw, db


def conv1d_backward_weight_wn(desc, x, dy, v, g, need_db=True, out_dv=None, out_dg=None, out_db=None):
    """(dv, dg, db) of a weight-normalised layer: weight-gradient kernel + ONE fused finishing kernel (slab sum
    + weight-norm backward); ``v`` is weight_v in torch layout, ``g`` weight_g.  ``out_*``: optional destinations."""
    _require_device(x, dy, v, g, out_dv, out_dg, out_db)
    dv = out_dv if out_dv is not None else torch.empty_like(v)
    dg = out_dg if out_dg is not None else torch.empty_like(g)
    db = (out_db if out_db is not None else torch.empty(desc.c_out, device=x.device, dtype=torch.float32)) if need_db else None
    ws_n = _lib.lib().pwg_conv1d_backward_weight_wn_workspace_floats(ctypes.byref(desc))
    ws = torch.empty(max(ws_n, 1), device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().pwg_conv1d_backward_weight_wn(ctypes.byref(desc), _ptr(x), _ptr(dy), _ptr(v), _ptr(g), _ptr(dv),
                                                        _ptr(dg), _ptr(db), _ptr(ws), ws_n, _stream()),
               "conv1d_backward_weight_wn")
    return dv, dg, db


def weight_norm_scale(v, g):
    """scale[i] = g[i] / ||v[i]||  (old-style weight_norm, dim=0)."""
    _require_device(v, g)
    n0 = v.shape[0]
    inner = v.numel() // n0
    scale = torch.empty(n0, device=v.device, dtype=torch.float32)
    _lib.check(_lib.lib().pwg_weight_norm_scale(_ptr(v), _ptr(g), _ptr(scale), n0, inner, _stream()), "weight_norm_scale")
    return scale


def scale_rows(v, scale):
    _require_device(v, scale)
    n0 = v.shape[0]
    inner = v.numel() // n0
    w = torch.empty_like(v)
    _lib.check(_lib.lib().pwg_scale_rows(_ptr(v), _ptr(scale), _ptr(w), n0, inner, _stream()), "scale_rows")
    return w


def conv1d_forward_cfg(desc, x, w_packed, bias=None, add1=None, add2=None, out=None, tile_config=0, use_dma=True):
    """Tuning entry: explicit tile configuration / staging path (see include/pwg_kernels.h)."""
    _require_device(x, w_packed, bias, add1, add2, out)
    if out is None:
        out = torch.empty((desc.batch, desc.c_out, desc.t_out * desc.width), device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().pwg_conv1d_forward_cfg(ctypes.byref(desc), _ptr(x), _ptr(w_packed), _ptr(bias), _ptr(add1),
                                                 _ptr(add2), _ptr(out), int(tile_config), int(bool(use_dma)),
                                                 _stream()), "conv1d_forward_cfg")
    return out


def num_tile_configs():
    return _lib.lib().pwg_conv1d_num_tile_configs()


PLAN_FAMILIES = ("mfma", "grouped", "single_input_channel", "few_output_channels")


def conv1d_plan(desc, has_addends=False):
    """The launch plan ``pwg_conv1d_forward`` derives for ``desc`` (host only: no launch, no device needed):
    dict(family, tile_config, ksplit, dma, item_major, grid)."""
    out = (ctypes.c_int32 * 8)()
    _lib.check(_lib.lib().pwg_conv1d_plan(ctypes.byref(desc), int(bool(has_addends)), out), "conv1d_plan")
    return dict(family=PLAN_FAMILIES[out[0]], tile_config=out[1], ksplit=out[2], dma=bool(out[3]),
                item_major=bool(out[4]), grid=(out[5], out[6], out[7]))


def conv_tile_of_workgroup(grid, row_blocks, ksplit, item_major, workgroup):
    """Host evaluation of the convolution kernel's dispatch-id -> logical-tile map (column tile, row block index,
    item * ksplit + slice)."""
    out = (ctypes.c_int32 * 3)()
    _lib.check(_lib.lib().pwg_debug_conv_tile_of_workgroup(int(grid[0]), int(grid[1]), int(grid[2]), int(row_blocks),
                                                           int(ksplit), int(bool(item_major)), int(workgroup), out),
               "conv_tile_of_workgroup")
    return out[0], out[1], out[2]


class profile:
    """Context manager: per-kernel-family HIP-event timing of every launch made through the
    C ABI (``pwg_prof_*``).  ``.results`` -> {kernel: dict(ms, launches, flops, bytes)}."""

    def __enter__(self):
        l = _lib.lib()
        l.pwg_prof_reset()
        l.pwg_prof_enable(1)
        self.results = {}
        return self

    def __exit__(self, *exc):
        l = _lib.lib()
        l.pwg_prof_enable(0)
        n = l.pwg_prof_num_kernels()
        for i in range(n):
            name = ctypes.create_string_buffer(256)
            ms, fl, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
            cnt = ctypes.c_int64()
            _lib.check(l.pwg_prof_get(i, name, 256, ctypes.byref(ms), ctypes.byref(cnt), ctypes.byref(fl),
                                      ctypes.byref(by)), "prof_get")
            self.results[name.value.decode()] = dict(ms=ms.value, launches=cnt.value, flops=fl.value, bytes=by.value)
        l.pwg_prof_reset()
        return False
