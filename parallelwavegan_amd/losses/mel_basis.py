"""Slaney mel filterbank for the mel-spectrogram loss.

The reference obtains it from ``librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)``
(/root/reference/parallel_wavegan/losses/mel_loss.py:52-59; librosa is an external, unvendored
dependency).  This is a from-scratch implementation of librosa's documented defaults
(``htk=False``, ``norm="slaney"``): a mel axis that is linear (200/3 Hz per mel) below 1 kHz and
logarithmic (ln 6.4 / 27 per mel) above, triangular filters between consecutive band edges,
each scaled by 2 / (band width in Hz) so that all filters have equal area.
"""
import numpy as np

_LIN_HZ_PER_MEL = 200.0 / 3.0
_BREAK_HZ = 1000.0
_BREAK_MEL = _BREAK_HZ / _LIN_HZ_PER_MEL
_LOG_STEP = np.log(6.4) / 27.0


def _to_mel(hz):
    hz = np.atleast_1d(np.asarray(hz, dtype=np.float64))
    mel = hz / _LIN_HZ_PER_MEL
    hi = hz >= _BREAK_HZ
    mel[hi] = _BREAK_MEL + np.log(hz[hi] / _BREAK_HZ) / _LOG_STEP
    return mel


def _to_hz(mel):
    mel = np.atleast_1d(np.asarray(mel, dtype=np.float64))
    hz = mel * _LIN_HZ_PER_MEL
    hi = mel >= _BREAK_MEL
    hz[hi] = _BREAK_HZ * np.exp(_LOG_STEP * (mel[hi] - _BREAK_MEL))
    return hz


def slaney_mel_basis(sr, n_fft, n_mels, fmin=0.0, fmax=None):
    """(n_mels, 1 + n_fft // 2) float32 filterbank."""
    fmax = sr / 2.0 if fmax is None else float(fmax)
    lo, hi = _to_mel(fmin)[0], _to_mel(fmax)[0]
    edges = _to_hz(lo + (hi - lo) * np.arange(n_mels + 2) / (n_mels + 1))
    bins = np.arange(1 + n_fft // 2) * (sr / float(n_fft))
    left, centre, right = edges[:-2, None], edges[1:-1, None], edges[2:, None]
    rising = (bins[None, :] - left) / (centre - left)
    falling = (right - bins[None, :]) / (right - centre)
    tri = np.clip(np.minimum(rising, falling), 0.0, None)
    tri *= 2.0 / (right - left)
    return tri.astype(np.float32)
