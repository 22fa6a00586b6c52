"""Chunked / batched synthesis around ``model.inference`` (reference call site
/root/reference/parallel_wavegan/bin/decode.py:214-243).

The reference synthesises one whole utterance per call.  A convolutional vocoder only looks at a
bounded window of mel frames around each output sample, so a long utterance (or many utterances)
can be cut into equal-length chunks that carry ``halo`` frames of real context on both sides, run
as ONE batched forward (one hipGraph replay per chunk shape) and stitched by dropping the halos.
With ``halo >= receptive field`` every kept sample is computed from exactly the inputs the
full-length forward would use -- the result is exact, not an overlap-add approximation (it differs
from the one-shot forward only by the fp32 summation order of whichever tile configuration the
convolution kernel picks for the two shapes, <= 1e-5).

The first / last chunk of an utterance see the model's own zero padding on their outer side, so they
run without a halo there (they are batched across utterances when their lengths agree).

Everything on the device runs in libpwgkernels.so: feature normalisation + (T', C) -> (C, T')
transpose, the generator, and the float -> PCM16 conversion.
"""
import ctypes

import torch

from .. import _lib
from ..graphs import GraphedInference
from ..ops import _ptr, _require_device, _stream


def normalize_transpose(c, mean=None, scale=None):
    """(B, T', C) features -> (B, C, T') with optional ``(c - mean) / scale`` (one HIP launch)."""
    c = c.contiguous()
    _require_device(c, mean, scale)
    b, t, ch = c.shape
    y = torch.empty((b, ch, t), device=c.device, dtype=torch.float32)
    _lib.check(_lib.lib().pwg_normalize_transpose(_ptr(c), _ptr(mean), _ptr(scale), _ptr(y), b, t, ch, _stream()),
               "normalize_transpose")
    return y


def to_pcm16(wave):
    """float waveform -> int16 PCM (clip to [-1, 1], scale 32767, round to nearest) on the device."""
    wave = wave.contiguous()
    _require_device(wave)
    pcm = torch.empty(wave.shape, device=wave.device, dtype=torch.int16)
    _lib.check(_lib.lib().pwg_wave_to_pcm16(_ptr(wave), ctypes.c_void_p(pcm.data_ptr()), wave.numel(), _stream()),
               "wave_to_pcm16")
    return pcm


@torch.no_grad()
def receptive_field_frames(model, in_channels=None, probe_frames=96):
    """Measured reach of the generator in mel frames: (left, right) = how many frames before / after
    frame t can influence the samples generated for frame t.  Found by perturbing one half of a random
    input and locating the first / last output sample that changes (exact for a convolutional model;
    the probe is doubled until the reach fits inside it)."""
    dev = next(model.parameters()).device
    ch = in_channels or getattr(model, "in_channels", None) or 80
    up = model.upsample_factor
    while True:
        n = probe_frames
        gen = torch.Generator(device="cpu").manual_seed(1234)
        c1 = torch.randn(1, ch, n, generator=gen).to(dev)
        half = n // 2
        c_future, c_past = c1.clone(), c1.clone()
        c_future[..., half:] = torch.randn(1, ch, n - half, generator=gen).to(dev)
        c_past[..., :half] = torch.randn(1, ch, half, generator=gen).to(dev)
        y = model(c1)
        d_future = (model(c_future) != y).flatten().nonzero()
        d_past = (model(c_past) != y).flatten().nonzero()
        # frames >= half changed: earliest affected sample tells how far the future reaches back
        first = int(d_future.min()) if d_future.numel() else half * up
        last = int(d_past.max()) if d_past.numel() else half * up - 1
        right = -(-(half * up - first) // up)          # frames of look-ahead
        left = -(-(last + 1 - half * up) // up)        # frames of look-back
        right, left = max(right, 0), max(left, 0)
        if max(left, right) < half - 1:
            return left, right
        probe_frames *= 2


class ChunkedSynthesizer:
    """Exact chunked synthesis for generators that map (B, C, T') mel -> (B, 1, T' * upsample_factor)
    from the mel alone (HiFi-GAN, MelGAN).  ``chunk_frames`` frames of output per chunk;
    ``max_batch`` chunks per forward."""

    def __init__(self, model, chunk_frames=256, max_batch=16, halo=None, use_graph=True):
        self.model = model.eval()
        self.chunk = int(chunk_frames)
        self.max_batch = int(max_batch)
        self.left, self.right = halo if halo is not None else receptive_field_frames(model)
        self.up = model.upsample_factor
        self._run = GraphedInference(self.model) if use_graph else self.model

    def _plan(self, n_frames):
        """[(start, end, ctx_start, ctx_end)] frame ranges of one utterance."""
        if n_frames <= self.chunk + self.left + self.right:
            return [(0, n_frames, 0, n_frames)]
        out = []
        for s in range(0, n_frames, self.chunk):
            e = min(n_frames, s + self.chunk)
            out.append((s, e, max(0, s - self.left), min(n_frames, e + self.right)))
        return out

    @torch.no_grad()
    def synthesize_many(self, feats, normalize_before=False):
        """feats: list of (T'_i, C) tensors/arrays -> list of (T'_i * upsample_factor,) float waveforms."""
        dev = next(self.model.parameters()).device
        mean = getattr(self.model, "mean", None) if normalize_before else None
        scale = getattr(self.model, "scale", None) if normalize_before else None
        mels = []
        for f in feats:
            f = torch.as_tensor(f, dtype=torch.float32).to(dev)
            mels.append(normalize_transpose(f.unsqueeze(0), mean, scale)[0])  # (C, T')
        outs = [torch.empty(m.shape[-1] * self.up, device=dev) for m in mels]
        # group chunks by (context length, whether the model's own padding is on the left / right):
        # only equal-shaped chunks share a batch, and edge chunks keep their true zero-padded side
        groups = {}
        for ui, m in enumerate(mels):
            n = m.shape[-1]
            for (s, e, cs, ce) in self._plan(n):
                groups.setdefault((ce - cs, cs == 0, ce == n), []).append((ui, s, e, cs, ce))
        for (length, _, _), items in groups.items():
            for i in range(0, len(items), self.max_batch):
                part = items[i:i + self.max_batch]
                batch = torch.stack([mels[ui][:, cs:ce] for (ui, s, e, cs, ce) in part])
                if len(part) < self.max_batch and len(items) > self.max_batch:
                    # keep one graph per chunk shape: pad the last partial batch with copies
                    pad = self.max_batch - len(part)
                    batch = torch.cat([batch, batch[:1].expand(pad, -1, -1)], 0)
                y = self._run(batch.contiguous())
                for j, (ui, s, e, cs, ce) in enumerate(part):
                    outs[ui][s * self.up:e * self.up] = y[j, 0, (s - cs) * self.up:(e - cs) * self.up]
        return outs

    def synthesize(self, feat, normalize_before=False):
        """(T', C) -> (T' * upsample_factor,)"""
        return self.synthesize_many([feat], normalize_before)[0]
