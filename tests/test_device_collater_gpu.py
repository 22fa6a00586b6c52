"""GPU: the HBM-resident collater yields exactly the batches of the host Collater."""
import numpy as np
import pytest
import torch

from parallelwavegan_amd.bin.train import Collater, DeviceCollater

pytestmark = pytest.mark.gpu


def _corpus(n=7, hop=256, seed=0):
    rng = np.random.RandomState(seed)
    pairs = []
    for i in range(n):
        frames = int(rng.randint(20, 90))
        # some waveforms are a little shorter than frames * hop (the reference edge-pads them)
        short = int(rng.randint(0, hop)) if i % 2 else 0
        pairs.append((rng.randn(frames * hop - short).astype(np.float32), rng.randn(frames, 80).astype(np.float32)))
    return pairs


@pytest.mark.parametrize("noise", [False, True])
def test_device_collater_matches_host_collater(noise, device):
    pairs = _corpus()
    kw = dict(batch_max_steps=8192, hop_size=256, aux_context_window=2, use_noise_input=noise)
    host = Collater(**kw)
    dev = DeviceCollater(pairs, device, **kw)
    idx = [3, 0, 6, 2, 5, 1, 4]
    np.random.seed(123)
    inputs_h, y_h = host([pairs[i] for i in idx])
    np.random.seed(123)
    inputs_d, y_d = dev(idx)
    assert y_d.is_cuda and inputs_d[-1].is_cuda
    assert torch.equal(y_d.cpu(), y_h)
    assert torch.equal(inputs_d[-1].cpu(), inputs_h[-1])
    if noise:
        assert inputs_d[0].shape == y_h.shape and inputs_d[0].is_cuda
    # utterances not longer than the crop are dropped by both
    n_long = sum(len(m) > 32 + 4 for _, m in pairs)
    assert y_d.shape[0] == y_h.shape[0] == n_long
