"""Feature extraction on the device: the reference's ``logmelfilterbank``
(/root/reference/parallel_wavegan/bin/preprocess.py:26-89 -- librosa STFT, |.|, Slaney mel basis,
``log10(max(eps, .))``) computed with the same fold + MFMA-convolution STFT kernels that serve the
spectral losses, so that training-time mel losses and offline features come from one code path
(the reference checks its two implementations against each other in test/test_mel_loss.py).

Only the arithmetic is provided here; file formats, F0 extraction and the CLI of the reference's
script are outside the accelerated path.
"""
import math

import numpy as np
import torch

from .. import functional as Fn
from ..losses.mel_basis import slaney_mel_basis
from ..losses.stft import STFTMagnitude


class LogMelFilterbank(torch.nn.Module):
    """(B, T) waveforms -> (B, #frames, num_mels) log-mel features, #frames = 1 + T // hop_size."""

    def __init__(self, sampling_rate, fft_size=1024, hop_size=256, win_length=None, window="hann", num_mels=80,
                 fmin=None, fmax=None, eps=1e-10, log_base=10.0):
        super().__init__()
        fmin = 0 if fmin is None else fmin
        fmax = sampling_rate / 2 if fmax is None else fmax
        if log_base is None:
            self.log_div = 1.0
        elif log_base in (10.0, 2.0):
            self.log_div = math.log(log_base)
        else:
            raise ValueError(f"{log_base} is not supported.")
        self.eps = eps
        # np.abs(stft): no floor on the magnitude (the loss module clamps the power at eps instead)
        self.stft_magnitude = STFTMagnitude(fft_size, hop_size, win_length, window, eps=0.0)
        melmat = slaney_mel_basis(sampling_rate, fft_size, num_mels, fmin, fmax)  # (mels, bins)
        self.register_buffer("mel_weight", torch.from_numpy(melmat[:, :, None].copy()).float(), persistent=False)
        self._geom = dict(kernel=1, stride=1, dilation=1, padding=0, groups=1, transposed=False, output_padding=0,
                          width=1, pad_mode="zero")
        self._fused = dict(pre_act=None, pre_slope=0.0, post_act=None, post_slope=0.0, out_mul=1.0, out_div=1.0)

    @torch.no_grad()
    def forward(self, audio):
        amp = self.stft_magnitude(audio)  # (B, bins, frames)
        mel = Fn.FusedConvFn.apply(amp, self.mel_weight, None, None, None, self._geom, self._fused, None)
        return Fn.LogClampFn.apply(mel, self.eps, self.log_div).transpose(1, 2)


_CACHE = {}


def logmelfilterbank(audio, sampling_rate, fft_size=1024, hop_size=256, win_length=None, window="hann", num_mels=80,
                     fmin=None, fmax=None, eps=1e-10, log_base=10.0, device="cuda"):
    """Drop-in for the reference function: audio (T,) ndarray -> (#frames, num_mels) ndarray."""
    key = (sampling_rate, fft_size, hop_size, win_length, window, num_mels, fmin, fmax, eps, log_base, str(device))
    if key not in _CACHE:
        _CACHE[key] = LogMelFilterbank(sampling_rate, fft_size, hop_size, win_length, window, num_mels, fmin, fmax,
                                       eps, log_base).to(device)
    x = torch.from_numpy(np.ascontiguousarray(audio, dtype=np.float32)).to(device).unsqueeze(0)
    return _CACHE[key](x)[0].cpu().numpy()
