"""Mel-spectrogram loss (drop-in for parallel_wavegan.losses.mel_loss)."""
import math

import torch

from .. import functional as Fn
from .mel_basis import slaney_mel_basis
from .stft import STFTMagnitude


class MelSpectrogram(torch.nn.Module):
    """log-mel spectrogram (B, #mels, #frames) (reference: losses/mel_loss.py:15-110):
    STFT magnitude (clamp eps) -> mel filterbank (1x1 conv on the MFMA kernel) -> clamp -> log."""

    def __init__(self, fs=22050, fft_size=1024, hop_size=256, win_length=None, window="hann", num_mels=80, fmin=80,
                 fmax=7600, center=True, normalized=False, onesided=True, eps=1e-10, log_base=10.0):
        super().__init__()
        if not center or normalized or not onesided:
            raise NotImplementedError("only center=True, normalized=False, onesided=True (the reference defaults)")
        self.fft_size = fft_size
        self.win_length = fft_size if win_length is None else win_length
        self.hop_size = hop_size
        self.window = window
        self.eps = eps
        if window is not None and not hasattr(torch, f"{window}_window"):
            raise ValueError(f"{window} window is not implemented")
        fmin = 0 if fmin is None else fmin
        fmax = fs / 2 if fmax is None else fmax
        melmat = slaney_mel_basis(fs, fft_size, num_mels, fmin, fmax)  # (mels, bins)
        self.register_buffer("melmat", torch.from_numpy(melmat.T.copy()).float())  # (bins, mels) as the reference
        self.register_buffer("mel_weight", torch.from_numpy(melmat[:, :, None].copy()).float(), persistent=False)
        self.log_base = log_base
        if log_base is None:
            self.log_div = 1.0
        elif log_base == 2.0:
            self.log_div = math.log(2.0)
        elif log_base == 10.0:
            self.log_div = math.log(10.0)
        else:
            raise ValueError(f"log_base: {log_base} is not supported.")
        self.stft_magnitude = STFTMagnitude(fft_size, hop_size, self.win_length, window, eps=eps)
        self._geom = dict(kernel=1, stride=1, dilation=1, padding=0, groups=1, transposed=False, output_padding=0,
                          width=1, pad_mode="zero")
        self._fused = dict(pre_act=None, pre_slope=0.0, post_act=None, post_slope=0.0, out_mul=1.0, out_div=1.0)

    def forward(self, x):
        """x: (B, T) or (B, 1, T) -> (B, #mels, #frames)."""
        if x.dim() == 3:
            x = x.reshape(-1, x.size(2))
        amp = self.stft_magnitude(x)  # (B, bins, frames)
        mel = Fn.FusedConvFn.apply(amp, self.mel_weight, None, None, None, self._geom, self._fused, None)
        return Fn.LogClampFn.apply(mel, self.eps, self.log_div)


class MelSpectrogramLoss(torch.nn.Module):
    """L1 between log-mel spectrograms (reference: losses/mel_loss.py:113-165)."""

    def __init__(self, fs=22050, fft_size=1024, hop_size=256, win_length=None, window="hann", num_mels=80, fmin=80,
                 fmax=7600, center=True, normalized=False, onesided=True, eps=1e-10, log_base=10.0):
        super().__init__()
        self.mel_spectrogram = MelSpectrogram(fs=fs, fft_size=fft_size, hop_size=hop_size, win_length=win_length,
                                              window=window, num_mels=num_mels, fmin=fmin, fmax=fmax, center=center,
                                              normalized=normalized, onesided=onesided, eps=eps, log_base=log_base)

    def forward(self, y_hat, y):
        return Fn.l1_mean(self.mel_spectrogram(y_hat), self.mel_spectrogram(y))
