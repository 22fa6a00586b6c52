"""torch.stft (center=True, reflect, one-sided, hann/other window) as HIP kernels.

    frames  = frame_fold(x)                       (B, hop, n_cols)      HBM-bound kernel
    spec    = conv1d(frames, windowed DFT basis)  (B, 2*bins, frames)   MFMA conv kernel
    mag     = sqrt(max(re^2 + im^2, eps))         (B, bins, frames)     HBM-bound kernel

A frame of n_fft samples with hop ``hop`` is ``K = ceil(win/hop)`` consecutive columns of the
folded signal, so the DFT is a stride-1 convolution with C_in = hop, C_out = 2*bins, K taps --
also for the non-power-of-two FFT sizes of the sub-band losses (683, 171, 384), where a radix-2
FFT does not apply.  Replaces ``torch.stft`` at
/root/reference/parallel_wavegan/losses/stft_loss.py:30-36 and losses/mel_loss.py:99.
"""
import math

import numpy as np
import torch

from .. import functional as Fn


def _window(name, win_length):
    n = np.arange(win_length, dtype=np.float64)
    name = name.replace("_window", "")
    if name == "hann":  # torch.hann_window is periodic by default
        return 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)
    if name == "hamming":
        return 0.54 - 0.46 * np.cos(2.0 * np.pi * n / win_length)
    if name == "blackman":
        return 0.42 - 0.5 * np.cos(2.0 * np.pi * n / win_length) + 0.08 * np.cos(4.0 * np.pi * n / win_length)
    if name in ("rect", "none", "ones"):
        return np.ones(win_length)
    raise ValueError(f"{name} window is not implemented")


class STFTMagnitude(torch.nn.Module):
    """|STFT| with the clamp of the reference losses; output (B, bins, frames)."""

    def __init__(self, fft_size, hop_size, win_length=None, window="hann", eps=1e-7):
        super().__init__()
        win_length = fft_size if win_length is None else win_length
        assert win_length <= fft_size
        self.fft_size, self.hop_size, self.win_length, self.eps = fft_size, hop_size, win_length, eps
        self.bins = fft_size // 2 + 1
        self.taps = int(math.ceil(win_length / hop_size))
        off = (fft_size - win_length) // 2  # torch pads the window to n_fft, centred
        self.frame_offset = off
        if isinstance(window, (np.ndarray, torch.Tensor)):  # explicit window samples (losses.stft_loss.stft)
            win = np.asarray(window.detach().cpu() if isinstance(window, torch.Tensor) else window, dtype=np.float64)
            assert win.shape == (win_length,), (win.shape, win_length)
        else:
            win = _window(window, win_length) if window is not None else np.ones(win_length)
        n = np.arange(self.taps * hop_size)
        valid = n < win_length
        w = np.where(valid, win[np.minimum(n, win_length - 1)], 0.0)
        phase = 2.0 * np.pi * np.outer(np.arange(self.bins), (n + off)) / fft_size
        basis = np.concatenate([np.cos(phase) * w, -np.sin(phase) * w], axis=0)  # (2*bins, taps*hop)
        # conv weight (C_out, C_in = hop, K = taps): tap j, channel c  <->  sample n = j*hop + c
        weight = basis.reshape(2 * self.bins, self.taps, hop_size).transpose(0, 2, 1)
        self.register_buffer("basis", torch.from_numpy(np.ascontiguousarray(weight, dtype=np.float32)),
                             persistent=False)
        self._geom = dict(kernel=self.taps, stride=1, dilation=1, padding=0, groups=1, transposed=False,
                          output_padding=0, width=1, pad_mode="zero")
        self._fused = dict(pre_act=None, pre_slope=0.0, post_act=None, post_slope=0.0, out_mul=1.0, out_div=1.0)

    def frames(self, t):
        # torch.stft(center=True): the signal is padded by n_fft // 2 on both sides (odd n_fft loses one)
        return 1 + (t + 2 * (self.fft_size // 2) - self.fft_size) // self.hop_size

    def spectrum(self, x):
        """x: (B, T) -> (B, 2*bins, frames) [real rows | imaginary rows]."""
        b, t = x.shape
        n_frames = self.frames(t)
        n_cols = n_frames + self.taps - 1
        folded = Fn.FrameFoldFn.apply(x, self.fft_size // 2 - self.frame_offset, self.hop_size, n_cols)
        return Fn.FusedConvFn.apply(folded, self.basis, None, None, None, self._geom, self._fused, None)

    def forward(self, x):
        return Fn.StftMagFn.apply(self.spectrum(x), self.eps)
