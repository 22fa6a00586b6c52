"""torch.stft (center=True, reflect, one-sided, hann/other window) as HIP kernels.

    frames  = frame_fold(x)                       (B, hop, n_cols)      HBM-bound kernel
    spec    = conv1d(frames, windowed DFT basis)  (B, 2*bins, frames)   MFMA conv kernel
    mag     = sqrt(max(re^2 + im^2, eps))         (B, bins, frames)     HBM-bound kernel

A frame of n_fft samples with hop ``hop`` is ``K = ceil(win/hop)`` consecutive columns of the
folded signal, so the DFT is a stride-1 convolution with C_in = hop, C_out = 2*bins, K taps --
also for the non-power-of-two FFT sizes of the sub-band losses (683, 171, 384), where a radix-2
FFT does not apply.  Replaces ``torch.stft`` at
/root/reference/parallel_wavegan/losses/stft_loss.py:30-36 and losses/mel_loss.py:99.
"""
import math
import os

import numpy as np
import torch

from .. import functional as Fn


def _window(name, win_length):
    n = np.arange(win_length, dtype=np.float64)
    name = name.replace("_window", "")
    if name == "hann":  # torch.hann_window is periodic by default
        return 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)
    if name == "hamming":
        return 0.54 - 0.46 * np.cos(2.0 * np.pi * n / win_length)
    if name == "blackman":
        return 0.42 - 0.5 * np.cos(2.0 * np.pi * n / win_length) + 0.08 * np.cos(4.0 * np.pi * n / win_length)
    if name in ("rect", "none", "ones"):
        return np.ones(win_length)
    raise ValueError(f"{name} window is not implemented")


class STFTMagnitude(torch.nn.Module):
    """|STFT| with the clamp of the reference losses; output (B, bins, frames)."""

    def __init__(self, fft_size, hop_size, win_length=None, window="hann", eps=1e-7, center=True):
        super().__init__()
        win_length = fft_size if win_length is None else win_length
        assert win_length <= fft_size
        self.fft_size, self.hop_size, self.win_length, self.eps = fft_size, hop_size, win_length, eps
        self.center = bool(center)  # False: frames start at n * hop of the un-padded signal (torch.stft(center=False))
        self.bins = fft_size // 2 + 1
        self.taps = int(math.ceil(win_length / hop_size))
        off = (fft_size - win_length) // 2  # torch pads the window to n_fft, centred
        self.frame_offset = off
        if isinstance(window, (np.ndarray, torch.Tensor)):  # explicit window samples (losses.stft_loss.stft)
            win = np.asarray(window.detach().cpu() if isinstance(window, torch.Tensor) else window, dtype=np.float64)
            assert win.shape == (win_length,), (win.shape, win_length)
        else:
            win = _window(window, win_length) if window is not None else np.ones(win_length)
        self._win64 = np.asarray(win, dtype=np.float64)  # (the FFT path multiplies by the window itself)
        n = np.arange(self.taps * hop_size)
        valid = n < win_length
        w = np.where(valid, win[np.minimum(n, win_length - 1)], 0.0)
        phase = 2.0 * np.pi * np.outer(np.arange(self.bins), (n + off)) / fft_size
        basis = np.concatenate([np.cos(phase) * w, -np.sin(phase) * w], axis=0)  # (2*bins, taps*hop)
        # conv weight (C_out, C_in = hop, K = taps): tap j, channel c  <->  sample n = j*hop + c
        weight = basis.reshape(2 * self.bins, self.taps, hop_size).transpose(0, 2, 1)
        self.register_buffer("basis", torch.from_numpy(np.ascontiguousarray(weight, dtype=np.float32)),
                             persistent=False)
        # image for the fused pair-loss kernel (csrc/stft_loss.hip): [tap][c][m_pad], 64-row groups of
        # 32 cosine rows + the 32 matching -sine rows, zero rows past `bins`
        ngroups = (self.bins + 31) // 32
        pair = np.zeros((self.taps, hop_size, ngroups * 64), dtype=np.float32)
        re = basis[: self.bins].reshape(self.bins, self.taps, hop_size)   # [bin][tap][c]
        im = basis[self.bins:].reshape(self.bins, self.taps, hop_size)
        for g in range(ngroups):
            lo, hi = g * 32, min(self.bins, g * 32 + 32)
            pair[:, :, g * 64: g * 64 + (hi - lo)] = re[lo:hi].transpose(1, 2, 0)
            pair[:, :, g * 64 + 32: g * 64 + 32 + (hi - lo)] = im[lo:hi].transpose(1, 2, 0)
        self.register_buffer("pair_basis", torch.from_numpy(pair), persistent=False)
        self._geom = dict(kernel=self.taps, stride=1, dilation=1, padding=0, groups=1, transposed=False,
                          output_padding=0, width=1, pad_mode="zero")
        self._fused = dict(pre_act=None, pre_slope=0.0, post_act=None, post_slope=0.0, out_mul=1.0, out_div=1.0)

    def frames(self, t):
        # torch.stft(center=True): the signal is padded by n_fft // 2 on both sides (odd n_fft loses one)
        if not self.center:
            if t < self.fft_size:
                raise ValueError(f"STFT(center=False): signal of {t} samples is shorter than n_fft = {self.fft_size}")
            return 1 + (t - self.fft_size) // self.hop_size
        return 1 + (t + 2 * (self.fft_size // 2) - self.fft_size) // self.hop_size

    @property
    def fold_pad(self):
        """Left shift of the folded signal: the reflected n_fft // 2 samples of center=True minus the offset of a
        window shorter than n_fft (negative without centring: the first windowed sample is x[frame_offset])."""
        return (self.fft_size // 2 if self.center else 0) - self.frame_offset

    def spectrum(self, x):
        """x: (B, T) -> (B, 2*bins, frames) [real rows | imaginary rows]."""
        b, t = x.shape
        n_frames = self.frames(t)
        n_cols = n_frames + self.taps - 1
        folded = Fn.FrameFoldFn.apply(x, self.fold_pad, self.hop_size, n_cols)
        return Fn.FusedConvFn.apply(folded, self.basis, None, None, None, self._geom, self._fused, None)

    def forward(self, x):
        return Fn.StftMagFn.apply(self.spectrum(x), self.eps)

    use_fft = os.environ.get("PWG_STFT_FFT", "1") == "1"  # power-of-two sizes: radix-2 FFT in LDS (csrc/stft_fft.hip)

    def pair_losses(self, x, y):
        """Fused single-launch losses of the pair: 2-element tensor [spectral convergence ||Y| - |X||_F / ||Y||_F,
        mean |log|Y| - log|X||] over (B, bins, frames); differentiable w.r.t. ``x`` (``y`` is a constant).
        Power-of-two FFT sizes (256 .. 2048) go through the FFT kernel, every other size (the sub-band losses' 171 /
        384 / 683) through the dense windowed DFT on MFMA."""
        if self.use_fft and self.center and x.shape[-1] > self.fft_size // 2 and self._fft_tables(x.device) is not None:
            return StftFftPairFn.apply(x, y.detach(), self)
        return StftPairSumsFn.apply(x, y.detach(), self)

    def _fft_tables(self, device):
        """(window, twiddle) device tensors for the FFT path, or None when the geometry is not covered."""
        from .. import _lib

        tabs = getattr(self, "_fft_tabs", None)
        if tabs is None or tabs[0] != str(device):
            if not _lib.lib().pwg_stft_fft_supported(self.fft_size, self.win_length, self.hop_size):
                tabs = (str(device), None)
            else:
                t = np.arange(self.fft_size // 2, dtype=np.float64) * (2.0 * np.pi / self.fft_size)
                tw = np.stack([np.cos(t), -np.sin(t)], axis=1).astype(np.float32)
                tabs = (str(device), (torch.from_numpy(self._win64.astype(np.float32)).to(device),
                                      torch.from_numpy(np.ascontiguousarray(tw)).to(device)))
            self._fft_tabs = tabs
        return tabs[1]


class StftPairSumsFn(torch.autograd.Function):
    """``pwg_stft_loss_forward`` / ``_backward`` (csrc/stft_loss.hip): no spectrum, magnitude or log tensor
    in HBM in the forward pass; the backward pass recomputes the tile spectra, writes d(re)/d(im) once and
    reuses the data-gradient convolution (transposed windowed DFT) and the fold's adjoint."""

    @staticmethod
    def forward(ctx, x, y, mod):
        import ctypes

        from .. import _lib, ops
        from ..ops import _ptr, _require_device, _stream

        x = x if x.is_contiguous() else x.contiguous()
        y = y if y.is_contiguous() else y.contiguous()
        _require_device(x, y)
        b, t = x.shape
        frames = mod.frames(t)
        n_cols = frames + mod.taps - 1
        pad = mod.fold_pad  # (0 - frame_offset when the module does not centre its frames)
        fx = torch.empty(b, mod.hop_size, n_cols, device=x.device, dtype=torch.float32)
        fy = torch.empty_like(fx)
        L = _lib.lib()
        for src, dst in ((x, fx), (y, fy)):
            _lib.check(L.pwg_frame_fold_forward(_ptr(src), _ptr(dst), b, t, pad, mod.hop_size, n_cols, _stream()),
                       "frame_fold_forward")
        ws = torch.empty(L.pwg_stft_loss_workspace_floats(b, mod.bins, frames), device=x.device, dtype=torch.float32)
        sums = torch.empty(5, device=x.device, dtype=torch.float32)  # [S_d, S_y, S_l, sc, mag]
        _lib.check(L.pwg_stft_loss_forward(_ptr(fx), _ptr(fy), _ptr(mod.pair_basis), b, mod.hop_size, n_cols, mod.taps,
                                           mod.bins, frames, float(mod.eps), _ptr(ws), _ptr(sums), _stream()),
                   "stft_loss_forward")
        ctx.save_for_backward(fx, fy, sums)
        ctx.mod, ctx.dims = mod, (b, t, frames, n_cols, pad)
        return sums[3:]

    @staticmethod
    def backward(ctx, g2):
        from .. import _lib, ops
        from ..ops import _ptr, _stream

        fx, fy, sums = ctx.saved_tensors
        mod = ctx.mod
        b, t, frames, n_cols, pad = ctx.dims
        g2 = g2.contiguous()
        L = _lib.lib()
        dspec = torch.empty(b, 2 * mod.bins, frames, device=fx.device, dtype=torch.float32)
        _lib.check(L.pwg_stft_loss_backward(_ptr(fx), _ptr(fy), _ptr(mod.pair_basis), b, mod.hop_size, n_cols, mod.taps,
                                            mod.bins, frames, float(mod.eps), _ptr(sums), _ptr(g2), _ptr(dspec), _stream()),
                   "stft_loss_backward")
        # d folded = transposed windowed DFT of (d re | d im): the data gradient of the DFT convolution
        desc = ops.make_conv_desc(b, mod.hop_size, 2 * mod.bins, n_cols, frames, mod.taps)
        if getattr(mod, "_bwd_image", None) is None or mod._bwd_image.device != fx.device:
            mod._bwd_image = ops.pack_weight_bwd(desc, mod.basis)
        dfold = ops.conv1d_backward_data(desc, dspec, mod._bwd_image)
        dx = torch.empty(b, t, device=fx.device, dtype=torch.float32)
        _lib.check(L.pwg_frame_fold_backward(_ptr(dfold), _ptr(dx), b, t, pad, mod.hop_size, n_cols, _stream()),
                   "frame_fold_backward")
        return dx, None, None


class StftFftPairFn(torch.autograd.Function):
    """``pwg_stft_fft_loss_forward`` / ``_backward`` (csrc/stft_fft.hip): the pair losses of a power-of-two resolution
    from the raw signals -- one complex radix-2 FFT in LDS per frame carries both signals; the backward pass re-runs
    it, transforms the per-bin gradients back with the same FFT and overlap-adds the windowed frame gradients with a
    deterministic gather.  Same values as :class:`StftPairSumsFn` to fp32 rounding (the FFT's error is the smaller)."""

    @staticmethod
    def forward(ctx, x, y, mod):
        from .. import _lib
        from ..ops import _ptr, _require_device, _stream

        x = x if x.is_contiguous() else x.contiguous()
        y = y if y.is_contiguous() else y.contiguous()
        _require_device(x, y)
        b, t = x.shape
        frames = mod.frames(t)
        window, twiddle = mod._fft_tables(x.device)
        L = _lib.lib()
        ws = torch.empty(L.pwg_stft_fft_workspace_floats(b, frames, mod.fft_size), device=x.device, dtype=torch.float32)
        sums = torch.empty(5, device=x.device, dtype=torch.float32)  # [S_d, S_y, S_l, sc, mag]
        _lib.check(L.pwg_stft_fft_loss_forward(_ptr(x), _ptr(y), _ptr(window), _ptr(twiddle), b, t, mod.fft_size,
                                               mod.hop_size, mod.win_length, float(mod.eps), _ptr(ws), _ptr(sums),
                                               _stream()), "stft_fft_loss_forward")
        ctx.save_for_backward(x, y, sums, window, twiddle)
        ctx.mod, ctx.dims = mod, (b, t, frames)
        return sums[3:]

    @staticmethod
    def backward(ctx, g2):
        from .. import _lib
        from ..ops import _ptr, _stream

        x, y, sums, window, twiddle = ctx.saved_tensors
        mod = ctx.mod
        b, t, frames = ctx.dims
        g2 = g2.contiguous()
        dframes = torch.empty(b, frames, mod.win_length, device=x.device, dtype=torch.float32)
        dx = torch.empty(b, t, device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib().pwg_stft_fft_loss_backward(_ptr(x), _ptr(y), _ptr(window), _ptr(twiddle), b, t, mod.fft_size,
                                                         mod.hop_size, mod.win_length, float(mod.eps), _ptr(sums), _ptr(g2),
                                                         _ptr(dframes), _ptr(dx), _stream()), "stft_fft_loss_backward")
        return dx, None, None
