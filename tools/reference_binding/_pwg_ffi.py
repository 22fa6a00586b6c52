"""Reference-side binding of libpwgkernels.so: the file a maintainer of kan-bayashi/ParallelWaveGAN would
add as ``parallel_wavegan/layers/_pwg_ffi.py`` to run ``HiFiGANResidualBlock.forward``
(layers/residual_block.py:243-258) on the MI355X kernels.  It uses nothing but ``ctypes`` + the public
header ``include/pwg_kernels.h`` (no import from the ``parallelwavegan_amd`` package), so it documents the
C ABI exactly as an outside caller sees it.  Executed by tests/test_reference_binding_gpu.py.
"""
import ctypes
import os

import torch

_DEFAULT = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                        "parallelwavegan_amd", "libpwgkernels.so")
_lib = ctypes.CDLL(os.environ.get("PWG_KERNELS_SO", _DEFAULT))

PWG_ACT_NONE, PWG_ACT_LEAKY_RELU = 0, 1


class ConvDesc(ctypes.Structure):  # mirrors pwg_conv1d_desc field by field
    _fields_ = [(n, ctypes.c_int32) for n in
                ("batch", "c_in", "c_out", "t_in", "t_out", "width", "kernel", "stride", "dilation",
                 "pad_left", "groups", "transposed", "pad_mode", "pre_act")] + \
               [("pre_slope", ctypes.c_float), ("post_act", ctypes.c_int32), ("post_slope", ctypes.c_float),
                ("out_mul", ctypes.c_float), ("out_div", ctypes.c_float)]


_vp, _dp = ctypes.c_void_p, ctypes.POINTER(ConvDesc)
_lib.pwg_last_error.restype = ctypes.c_char_p
_lib.pwg_abi_version.restype = ctypes.c_int
_lib.pwg_conv1d_packed_weight_floats.restype = ctypes.c_size_t
_lib.pwg_conv1d_packed_weight_floats.argtypes = [_dp]
_lib.pwg_conv1d_pack_weight.restype = ctypes.c_int
_lib.pwg_conv1d_pack_weight.argtypes = [_dp, _vp, _vp, _vp, _vp]  # desc, w, scale, w_packed, stream
_lib.pwg_weight_norm_scale.restype = ctypes.c_int
_lib.pwg_weight_norm_scale.argtypes = [_vp, _vp, _vp, ctypes.c_int32, ctypes.c_int32, _vp]
_lib.pwg_conv1d_forward_workspace_floats.restype = ctypes.c_size_t
_lib.pwg_conv1d_forward_workspace_floats.argtypes = [_dp]
_lib.pwg_conv1d_forward.restype = ctypes.c_int
_lib.pwg_conv1d_forward.argtypes = [_dp] + [_vp] * 7 + [ctypes.c_size_t, _vp]  # ..., workspace, ws_floats, stream


class ResUnitDesc(ctypes.Structure):  # mirrors pwg_resunit_desc field by field
    _fields_ = [(n, ctypes.c_int32) for n in ("batch", "channels", "t", "kernel", "dilation", "has_conv2")] + \
               [("slope1", ctypes.c_float), ("slope2", ctypes.c_float), ("out_div", ctypes.c_float)]


_rp = ctypes.POINTER(ResUnitDesc)
_lib.pwg_resunit_supported.restype = ctypes.c_int
_lib.pwg_resunit_supported.argtypes = [_rp]
_lib.pwg_resunit_packed_weight_floats.restype = ctypes.c_size_t
_lib.pwg_resunit_packed_weight_floats.argtypes = [ctypes.c_int32, ctypes.c_int32]
_lib.pwg_resunit_pack_weight.restype = ctypes.c_int
_lib.pwg_resunit_pack_weight.argtypes = [ctypes.c_int32, ctypes.c_int32, _vp, _vp, _vp, _vp]  # C, k, w, scale, out, stream
_lib.pwg_resunit_forward.restype = ctypes.c_int
_lib.pwg_resunit_forward.argtypes = [_rp] + [_vp] * 8  # desc, x, w1p, b1, w2p, b2, add2, y, stream


def _check(rc):
    if rc != 0:
        raise RuntimeError(_lib.pwg_last_error().decode())


def _stream():
    return _vp(torch.cuda.current_stream().cuda_stream)


def _desc(b, c, t, k, dilation, slope):
    return ConvDesc(b, c, c, t, t, 1, k, 1, dilation, (k - 1) // 2 * dilation, 1, 0, 0, PWG_ACT_LEAKY_RELU, slope,
                    PWG_ACT_NONE, 0.0, 1.0, 1.0)


def pack_weight(weight_v, weight_g, k, dilation):
    """Kernel-side image of a weight-normalised Conv1d weight (weight = g * v / |v| folded into the re-layout)."""
    c_out, c_in, _ = weight_v.shape
    d = _desc(1, c_in, 8 * k * dilation, k, dilation, 0.1)
    scale = torch.empty(c_out, device=weight_v.device)
    _check(_lib.pwg_weight_norm_scale(weight_v.data_ptr(), weight_g.data_ptr(), scale.data_ptr(), c_out, c_in * k,
                                      _stream()))
    packed = torch.empty(_lib.pwg_conv1d_packed_weight_floats(ctypes.byref(d)), device=weight_v.device)
    _check(_lib.pwg_conv1d_pack_weight(ctypes.byref(d), weight_v.data_ptr(), scale.data_ptr(), packed.data_ptr(),
                                       _stream()))
    return packed


def conv1d_lrelu_residual(x, w_packed, bias, residual, k, dilation, slope):
    """``residual + conv_{k,d}(LeakyReLU(x)) + bias`` in one launch (residual_block.py:247-257)."""
    b, c, t = x.shape
    d = _desc(b, c, t, k, dilation, slope)
    y = torch.empty_like(x)
    n_ws = _lib.pwg_conv1d_forward_workspace_floats(ctypes.byref(d))  # > 0 only for split reductions
    ws = torch.empty(n_ws, device=x.device) if n_ws else None
    _check(_lib.pwg_conv1d_forward(ctypes.byref(d), x.data_ptr(), w_packed.data_ptr(), bias.data_ptr(),
                                   residual.data_ptr(), None, y.data_ptr(), ws.data_ptr() if n_ws else None, n_ws,
                                   _stream()))
    return y


def hifigan_residual_block_forward(x, params, kernel_size, dilations, slope=0.1):
    """``HiFiGANResidualBlock.forward`` with ``use_additional_convs=True`` on the library:
    params[i] = (v1, g1, b1, v2, g2, b2) of convs1[i] / convs2[i]."""
    for (v1, g1, b1, v2, g2, b2), dil in zip(params, dilations):
        xt = conv1d_lrelu_residual(x, pack_weight(v1, g1, kernel_size, dil), b1, torch.zeros_like(x), kernel_size, dil,
                                   slope)
        x = conv1d_lrelu_residual(xt, pack_weight(v2, g2, kernel_size, 1), b2, x, kernel_size, 1, slope)
    return x


def resunit_pack_weight(weight_v, weight_g):
    """MFMA A-operand image of a weight-normalised (C, C, k) weight for the one-launch residual unit."""
    c, _, k = weight_v.shape
    scale = torch.empty(c, device=weight_v.device)
    _check(_lib.pwg_weight_norm_scale(weight_v.data_ptr(), weight_g.data_ptr(), scale.data_ptr(), c, c * k, _stream()))
    packed = torch.empty(_lib.pwg_resunit_packed_weight_floats(c, k), device=weight_v.device)
    _check(_lib.pwg_resunit_pack_weight(c, k, weight_v.data_ptr(), scale.data_ptr(), packed.data_ptr(), _stream()))
    return packed


def hifigan_residual_block_forward_one_launch_per_unit(x, params, kernel_size, dilations, slope=0.1):
    """Inference variant: each ``xt = convs1[idx](x); xt = convs2[idx](xt); x = xt + x`` iteration
    (residual_block.py:253-257) is ONE ``pwg_resunit_forward`` launch where the library supports the geometry
    (32 / 64 channels), else the two launches above."""
    b, c, t = x.shape
    for (v1, g1, b1, v2, g2, b2), dil in zip(params, dilations):
        d = ResUnitDesc(b, c, t, kernel_size, dil, 1, slope, slope, 1.0)
        if not _lib.pwg_resunit_supported(ctypes.byref(d)):
            x = hifigan_residual_block_forward(x, [(v1, g1, b1, v2, g2, b2)], kernel_size, [dil], slope)
            continue
        w1, w2 = resunit_pack_weight(v1, g1), resunit_pack_weight(v2, g2)  # (a real module caches these per weight)
        y = torch.empty_like(x)
        _check(_lib.pwg_resunit_forward(ctypes.byref(d), x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                                        b2.data_ptr(), None, y.data_ptr(), _stream()))
        x = y
    return x
