"""CPU: Collater crop algebra (the reference's test-free contract, bin/train.py:711-798)."""
import numpy as np
import torch

from parallelwavegan_amd.bin.train import Collater


def _utt(frames, hop, rng):
    c = rng.standard_normal((frames, 80)).astype(np.float32)
    x = np.repeat(np.arange(frames, dtype=np.float32), hop) + 0.001 * np.tile(np.arange(hop, dtype=np.float32), frames)
    c[:, 0] = np.arange(frames)  # channel 0 carries the frame index: lets the test check alignment
    return x, c


def test_collater_aligns_audio_and_mel_windows():
    rng = np.random.default_rng(0)
    np.random.seed(0)
    hop, acw = 256, 2
    col = Collater(batch_max_steps=8192 + 100, hop_size=hop, aux_context_window=acw, use_noise_input=True)
    assert col.batch_max_steps == 8192 and col.batch_max_frames == 32
    batch = [_utt(f, hop, rng) for f in (40, 100, 36, 37, 300)]  # 36 frames == threshold -> dropped
    (z, c), y = col(batch)
    assert y.shape == (4, 1, 8192) and c.shape == (4, 80, 32 + 2 * acw) and z.shape == y.shape
    assert y.dtype == c.dtype == z.dtype == torch.float32
    for b in range(4):
        first_frame = int(c[b, 0, acw].item())           # frame index of the first non-context frame
        assert int(c[b, 0, 0].item()) == first_frame - acw
        assert int(y[b, 0, 0].item()) == first_frame      # audio sample start = frame * hop
        assert int(y[b, 0, -1].item()) == first_frame + 31


def test_collater_without_noise_and_length_fix():
    rng = np.random.default_rng(1)
    np.random.seed(1)
    col = Collater(batch_max_steps=2560, hop_size=256, aux_context_window=0)
    x, c = _utt(20, 256, rng)
    (cb,), y = col([(x[:-5], c)])  # short audio is edge-padded to frames * hop
    assert cb.shape == (1, 80, 10) and y.shape == (1, 1, 2560)


def test_collater_matches_reference_collater():
    """Same numpy seed -> the same batches as the reference's Collater (bin/train.py:646-896), for the
    plain, the noise-input (PWG) and the f0/excitation (UHiFiGAN) variants.  Build container only."""
    import pytest

    from oracle import ref_shim

    if not ref_shim.available():
        pytest.skip("/root/reference is not present")
    ref_shim.install()
    from parallel_wavegan.bin.train import Collater as RefCollater

    rng = np.random.default_rng(2)
    hop = 64
    items = []
    for frames in (50, 33, 90, 41):
        x, c = _utt(frames, hop, rng)
        f0 = rng.standard_normal(frames).astype(np.float32)
        ex = rng.standard_normal((frames, hop)).astype(np.float32)
        items.append((x, c, f0, ex))
    for kw, n in ((dict(), 2), (dict(use_noise_input=True), 2), (dict(use_f0_and_excitation=True), 4)):
        args = dict(batch_max_steps=32 * hop, hop_size=hop, aux_context_window=2, **kw)
        batch = [it[:n] for it in items]
        np.random.seed(5)
        torch.manual_seed(5)
        want_in, want_y = RefCollater(**args)(batch)
        np.random.seed(5)
        torch.manual_seed(5)
        got_in, got_y = Collater(**args)(batch)
        assert torch.equal(got_y, want_y)
        assert len(got_in) == len(want_in)
        for a, b in zip(got_in, want_in):
            assert a.shape == b.shape and torch.equal(a, b), kw
