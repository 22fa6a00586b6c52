"""GPU: chunked / batched synthesis equals whole-utterance inference; PCM16 and the
fused normalise+transpose helpers equal their torch definitions."""
import numpy as np
import pytest
import torch

from parallelwavegan_amd import models
from parallelwavegan_amd.utils import streaming
from tests.golden import synth
from tests.util import synth_for

pytestmark = pytest.mark.gpu


def _hifigan(device):
    g = models.HiFiGANGenerator(**synth.HIFIGAN_V1)
    g.load_state_dict(synth_for(g, 11, 1.25))
    g.remove_weight_norm()
    return g.to(device).eval()


def test_receptive_field_probe(device):
    g = _hifigan(device)
    left, right = streaming.receptive_field_frames(g)
    # analytic bound for V1: output conv 3 + per stage 60 samples of MRF k=11 reach, folded back
    # through the transposed convs (k = 2s) and the k=7 input conv: 16 frames
    assert 1 <= left <= 16 and 1 <= right <= 16, (left, right)
    causal = models.HiFiGANGenerator(**synth.HIFIGAN_CAUSAL).to(device).eval()
    left_c, right_c = streaming.receptive_field_frames(causal)
    assert right_c == 0 and left_c >= 1


def test_chunked_synthesis_equals_full_inference(device):
    g = _hifigan(device)
    gen = torch.Generator().manual_seed(3)
    feats = [torch.randn(n, 80, generator=gen) for n in (300, 77, 190)]
    syn = streaming.ChunkedSynthesizer(g, chunk_frames=64, max_batch=4)
    outs = syn.synthesize_many(feats)
    for f, y in zip(feats, outs):
        full = g.inference(f.to(device)).reshape(-1)
        assert y.shape == full.shape == (f.shape[0] * 256,)
        assert (y - full).abs().max().item() <= 2e-5
    # a halo that is too short must show up (guards the test against a trivially passing comparison)
    bad = streaming.ChunkedSynthesizer(g, chunk_frames=64, max_batch=4, halo=(1, 1), use_graph=False)
    y_bad = bad.synthesize(feats[0])
    assert (y_bad - g.inference(feats[0].to(device)).reshape(-1)).abs().max().item() > 1e-3


def test_normalize_before_and_pcm16(device):
    g = _hifigan(device)
    gen = torch.Generator().manual_seed(4)
    mean, scale = torch.randn(80, generator=gen), torch.rand(80, generator=gen) + 0.5
    g.register_buffer("mean", mean.to(device))
    g.register_buffer("scale", scale.to(device))
    f = torch.randn(120, 80, generator=gen) * scale + mean
    syn = streaming.ChunkedSynthesizer(g, chunk_frames=48, max_batch=8)
    y = syn.synthesize(f, normalize_before=True)
    full = g.inference(f.to(device), normalize_before=True).reshape(-1)
    assert (y - full).abs().max().item() <= 2e-5
    c = torch.randn(3, 50, 80, generator=gen)
    got = streaming.normalize_transpose(c.to(device), mean.to(device), scale.to(device)).cpu()
    assert torch.allclose(got, ((c - mean) / scale).transpose(1, 2), atol=1e-6)
    w = torch.cat([torch.linspace(-1.5, 1.5, 1001), torch.tensor([0.5 / 32767, -0.5 / 32767, 1.0, -1.0])])
    pcm = streaming.to_pcm16(w.to(device)).cpu().numpy()
    want = np.rint(np.clip(w.numpy().astype(np.float32), -1, 1) * np.float32(32767)).astype(np.int16)
    assert pcm.dtype == np.int16 and np.array_equal(pcm, want)
