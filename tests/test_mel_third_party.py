"""The reference takes its mel filterbank and STFT from librosa (losses/mel_loss.py:52-59,
bin/preprocess.py:58-77), which is neither vendored nor installed here.  This pins the restatements --
the oracle's (oracle/slaney_mel.py, oracle/logmel_numpy.py) AND the product's
(parallelwavegan_amd/losses/mel_basis.py) -- against an independent third-party implementation of the same
published definitions: HuggingFace ``transformers.audio_utils`` (``mel_filter_bank(norm="slaney",
mel_scale="slaney")`` and ``spectrogram(...)``, written and tested against librosa upstream).  Not librosa
itself, but no longer two restatements by the same author agreeing with each other."""
import numpy as np
import pytest

audio_utils = pytest.importorskip("transformers.audio_utils")

CASES = [(22050, 1024, 80, 80, 7600), (22050, 1024, 80, 0, 11025), (24000, 2048, 80, 0, 12000), (16000, 512, 40, 50, 8000)]


@pytest.mark.parametrize("sr,n_fft,n_mels,fmin,fmax", CASES)
def test_mel_filterbanks_match_third_party(sr, n_fft, n_mels, fmin, fmax):
    from oracle.slaney_mel import mel as oracle_mel
    from parallelwavegan_amd.losses.mel_basis import slaney_mel_basis

    want = audio_utils.mel_filter_bank(num_frequency_bins=n_fft // 2 + 1, num_mel_filters=n_mels, min_frequency=fmin,
                                       max_frequency=fmax, sampling_rate=sr, norm="slaney", mel_scale="slaney").T
    for got in (slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax), oracle_mel(sr, n_fft, n_mels, fmin, fmax)):
        assert got.shape == want.shape == (n_mels, n_fft // 2 + 1)
        assert np.abs(got - want).max() <= 1e-7 * np.abs(want).max() + 1e-9


@pytest.mark.parametrize("fft_size,hop,win,fmin,fmax,fs", [(1024, 256, None, 80, 7600, 22050), (2048, 300, 1200, 0, 12000, 24000)])
def test_logmelfilterbank_oracle_matches_third_party_spectrogram(fft_size, hop, win, fmin, fmax, fs):
    from oracle import logmel_numpy

    rng = np.random.RandomState(3)
    t = np.arange(20000) / fs
    audio = (0.4 * np.sin(2 * np.pi * 440 * t) + 0.05 * rng.randn(len(t))).astype(np.float64)
    win_length = fft_size if win is None else win
    window = audio_utils.window_function(win_length, "hann", periodic=True, frame_length=fft_size)
    filters = audio_utils.mel_filter_bank(num_frequency_bins=fft_size // 2 + 1, num_mel_filters=80, min_frequency=fmin,
                                          max_frequency=fmax, sampling_rate=fs, norm="slaney", mel_scale="slaney")
    want = audio_utils.spectrogram(audio, window, frame_length=fft_size, hop_length=hop, fft_length=fft_size, power=1.0,
                                   center=True, pad_mode="reflect", mel_filters=filters, mel_floor=1e-10,
                                   log_mel="log10").T
    got = logmel_numpy.logmelfilterbank(audio, fs, fft_size, hop, win, "hann", 80, fmin, fmax)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-6
