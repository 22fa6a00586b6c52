"""numpy restatement of the reference's ``logmelfilterbank``
(/root/reference/parallel_wavegan/bin/preprocess.py:26-89).

TEST INFRASTRUCTURE (see oracle/__init__.py).

Third-party algorithm: the reference calls ``librosa.stft`` and ``librosa.filters.mel``
(``librosa>=0.8.0``, setup.py:29), which are neither vendored nor installed here.  Their published
definitions are restated: ``stft(y, n_fft, hop_length, win_length, window, center=True,
pad_mode="reflect")`` = reflect-pad by n_fft//2, frames of n_fft samples every hop_length, multiplied
by the periodic (``fftbins=True``) window of win_length zero-padded symmetrically to n_fft, rFFT.
Not pinned against librosa itself (absent); pinned against ``torch.stft`` (the formulation of the
reference's own ``MelSpectrogram``, losses/mel_loss.py:81-110, which its test_mel_loss.py asserts
equal to logmelfilterbank) in tests/test_oracle_golden.py AND against an independent third-party
implementation of the librosa definitions, ``transformers.audio_utils.spectrogram`` /
``mel_filter_bank(norm="slaney", mel_scale="slaney")``, in tests/test_mel_third_party.py (<= 1e-6).
"""
import numpy as np
import scipy.signal

from .slaney_mel import mel as mel_filterbank


def stft_magnitude(audio, fft_size, hop_size, win_length=None, window="hann"):
    win_length = fft_size if win_length is None else win_length
    w = scipy.signal.get_window(window, win_length, fftbins=True)
    lpad = (fft_size - win_length) // 2
    w = np.pad(w, (lpad, fft_size - win_length - lpad))
    y = np.pad(np.asarray(audio, dtype=np.float64), fft_size // 2, mode="reflect")
    n_frames = 1 + (len(y) - fft_size) // hop_size
    idx = np.arange(fft_size)[None, :] + hop_size * np.arange(n_frames)[:, None]
    return np.abs(np.fft.rfft(y[idx] * w[None, :], axis=1))  # (#frames, #bins)


def logmelfilterbank(audio, sampling_rate, fft_size=1024, hop_size=256, win_length=None, window="hann", num_mels=80,
                     fmin=None, fmax=None, eps=1e-10, log_base=10.0):
    spc = stft_magnitude(audio, fft_size, hop_size, win_length, window)
    fmin = 0 if fmin is None else fmin
    fmax = sampling_rate / 2 if fmax is None else fmax
    basis = mel_filterbank(sampling_rate, fft_size, num_mels, fmin, fmax).astype(np.float64)
    mel = np.maximum(eps, spc @ basis.T)
    if log_base is None:
        return np.log(mel)
    if log_base == 10.0:
        return np.log10(mel)
    if log_base == 2.0:
        return np.log2(mel)
    raise ValueError(f"{log_base} is not supported.")
