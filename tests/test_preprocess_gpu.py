"""GPU: device logmelfilterbank vs the numpy restatement of the reference function."""
import numpy as np
import pytest

from oracle import logmel_numpy
from parallelwavegan_amd.bin.preprocess import logmelfilterbank

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fft_size,hop,win,fmin,fmax,fs", [(1024, 256, None, 80, 7600, 22050),
                                                           (2048, 300, 1200, None, None, 24000)])
def test_logmelfilterbank_matches_oracle(fft_size, hop, win, fmin, fmax, fs, device):
    rng = np.random.RandomState(1)
    t = np.arange(30000) / fs
    audio = (0.4 * np.sin(2 * np.pi * 440 * t) + 0.05 * rng.randn(len(t))).astype(np.float32)
    got = logmelfilterbank(audio, fs, fft_size, hop, win, "hann", 80, fmin, fmax, device=device)
    want = logmel_numpy.logmelfilterbank(audio, fs, fft_size, hop, win, "hann", 80, fmin, fmax)
    assert got.shape == want.shape == (1 + len(audio) // hop, 80)
    # log10 of a direct fp32 DFT against float64: absolute error of the log is the relative error
    # of the mel energy / ln(10); 2e-4 covers the weakest bands of this signal
    assert np.abs(got - want).max() < 2e-4
