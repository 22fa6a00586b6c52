"""ctypes front-end of oracle/c/libpwgoracle.so (plain-C restatement; TEST INFRASTRUCTURE)."""
import ctypes
import os

import numpy as np

from .c import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_build.build())
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def conv1d(x, w, b, stride=1, dilation=1, padding=0, groups=1, slope=None):
    x, xp = _f(x)
    w, wp = _f(w)
    B, Cin, T = x.shape
    Cout, _, K = w.shape
    Tout = (T + 2 * padding - dilation * (K - 1) - 1) // stride + 1
    y = np.empty((B, Cout, Tout), np.float32)
    bp = None
    if b is not None:
        b, bp = _f(b)
    lib().pwgo_conv1d(xp, wp, bp, y.ctypes.data_as(ctypes.c_void_p), B, Cin, Cout, T, K, stride, dilation, padding,
                      groups, int(slope is not None), ctypes.c_float(slope or 0.0))
    return y


def conv_transpose1d(x, w, b, stride, padding, output_padding=0, slope=None):
    x, xp = _f(x)
    w, wp = _f(w)
    B, Cin, T = x.shape
    _, Cout, K = w.shape
    Tout = (T - 1) * stride - 2 * padding + K + output_padding
    y = np.empty((B, Cout, Tout), np.float32)
    bp = None
    if b is not None:
        b, bp = _f(b)
    lib().pwgo_conv_transpose1d(xp, wp, bp, y.ctypes.data_as(ctypes.c_void_p), B, Cin, Cout, T, K, stride, padding,
                                output_padding, int(slope is not None), ctypes.c_float(slope or 0.0))
    return y


def avg_pool1d(x, kernel, stride, padding, count_include_pad=True):
    x, xp = _f(x)
    T = x.shape[-1]
    rows = x.size // T
    Tout = (T + 2 * padding - kernel) // stride + 1
    y = np.empty(x.shape[:-1] + (Tout,), np.float32)
    lib().pwgo_avg_pool1d(xp, y.ctypes.data_as(ctypes.c_void_p), rows, T, kernel, stride, padding, int(count_include_pad))
    return y


def stft_mag(x, n_fft, hop, win, eps=1e-7):
    x, xp = _f(x)
    T = x.shape[-1]
    frames = 1 + (T + 2 * (n_fft // 2) - n_fft) // hop
    mag = np.empty((frames, n_fft // 2 + 1), np.float32)
    lib().pwgo_stft_mag(xp, mag.ctypes.data_as(ctypes.c_void_p), T, n_fft, hop, win, ctypes.c_float(eps))
    return mag
