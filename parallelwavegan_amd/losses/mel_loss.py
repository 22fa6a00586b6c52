"""Mel-spectrogram loss (drop-in for parallel_wavegan.losses.mel_loss)."""
import ctypes
import math

import numpy as np
import torch

from .. import functional as Fn
from .mel_basis import slaney_mel_basis
from .stft import STFTMagnitude


class MelSpectrogram(torch.nn.Module):
    """log-mel spectrogram (B, #mels, #frames) (reference: losses/mel_loss.py:15-110):
    STFT magnitude (clamp eps) -> mel filterbank (1x1 conv on the MFMA kernel) -> clamp -> log."""

    def __init__(self, fs=22050, fft_size=1024, hop_size=256, win_length=None, window="hann", num_mels=80, fmin=80,
                 fmax=7600, center=True, normalized=False, onesided=True, eps=1e-10, log_base=10.0):
        super().__init__()
        if not onesided:
            # the reference accepts the flag but its own forward cannot run with it: torch.stft then returns n_fft
            # bins and the matmul with the (n_fft // 2 + 1, num_mels) filterbank fails (losses/mel_loss.py:99-106)
            raise ValueError("MelSpectrogram(onesided=False): the mel filterbank covers n_fft // 2 + 1 bins "
                             "(the reference's forward raises a shape error for this option as well)")
        self.center, self.normalized, self.onesided = bool(center), bool(normalized), True
        self.fft_size = fft_size
        self.win_length = fft_size if win_length is None else win_length
        self.hop_size = hop_size
        self.window = window
        self.eps = eps
        if window is not None and not hasattr(torch, f"{window}_window"):
            raise ValueError(f"{window} window is not implemented")
        fmin = 0 if fmin is None else fmin
        fmax = fs / 2 if fmax is None else fmax
        melmat = slaney_mel_basis(fs, fft_size, num_mels, fmin, fmax)  # (mels, bins)
        self.register_buffer("melmat", torch.from_numpy(melmat.T.copy()).float())  # (bins, mels) as the reference
        stft_eps = eps
        if self.normalized:
            # torch.stft(normalized=True) scales the spectrum by n_fft ** -0.5 BEFORE the reference clamps the power:
            # sqrt(max(P / n, eps)) = sqrt(max(P, eps * n)) / sqrt(n) -- clamp at eps * n, fold 1 / sqrt(n) into the
            # filterbank the kernels use (the ``melmat`` buffer keeps the reference's values)
            stft_eps = eps * fft_size
            melmat = melmat / math.sqrt(fft_size)
        self.register_buffer("mel_weight", torch.from_numpy(melmat[:, :, None].copy()).float(), persistent=False)
        self.log_base = log_base
        if log_base is None:
            self.log_div = 1.0
        elif log_base == 2.0:
            self.log_div = math.log(2.0)
        elif log_base == 10.0:
            self.log_div = math.log(10.0)
        else:
            raise ValueError(f"log_base: {log_base} is not supported.")
        self.stft_magnitude = STFTMagnitude(fft_size, hop_size, self.win_length, window, eps=stft_eps, center=self.center)
        # filterbank images of the fused pair-loss kernel (csrc/stft_loss.hip): zero padded to whole 32 x 32 tiles
        bins = melmat.shape[1]
        bins_pad, mels_pad = 32 * ((bins + 31) // 32), 32 * ((num_mels + 31) // 32)
        mel_t = np.zeros((bins_pad, mels_pad), dtype=np.float32)
        mel_t[:bins, :num_mels] = melmat.T
        self.register_buffer("mel_t", torch.from_numpy(mel_t), persistent=False)               # [bin][mel]
        self.register_buffer("mel_b", torch.from_numpy(mel_t.T.copy()), persistent=False)      # [mel][bin]
        self.num_mels = num_mels
        self._geom = dict(kernel=1, stride=1, dilation=1, padding=0, groups=1, transposed=False, output_padding=0,
                          width=1, pad_mode="zero")
        self._fused = dict(pre_act=None, pre_slope=0.0, post_act=None, post_slope=0.0, out_mul=1.0, out_div=1.0)

    def forward(self, x):
        """x: (B, T) or (B, 1, T) -> (B, #mels, #frames)."""
        if x.dim() == 3:
            x = x.reshape(-1, x.size(2))
        amp = self.stft_magnitude(x)  # (B, bins, frames)
        mel = Fn.FusedConvFn.apply(amp, self.mel_weight, None, None, None, self._geom, self._fused, None)
        return Fn.LogClampFn.apply(mel, self.eps, self.log_div)


class MelSpectrogramLoss(torch.nn.Module):
    """L1 between log-mel spectrograms (reference: losses/mel_loss.py:113-165)."""

    def __init__(self, fs=22050, fft_size=1024, hop_size=256, win_length=None, window="hann", num_mels=80, fmin=80,
                 fmax=7600, center=True, normalized=False, onesided=True, eps=1e-10, log_base=10.0):
        super().__init__()
        self.mel_spectrogram = MelSpectrogram(fs=fs, fft_size=fft_size, hop_size=hop_size, win_length=win_length,
                                              window=window, num_mels=num_mels, fmin=fmin, fmax=fmax, center=center,
                                              normalized=normalized, onesided=onesided, eps=eps, log_base=log_base)

    fused = True  # one fused launch for both signals (csrc/stft_loss.hip); False: the op-by-op chain

    def forward(self, y_hat, y):
        ms = self.mel_spectrogram
        # the fused pair kernels implement the recipes' form (centred frames, clamp at eps); other STFT options take
        # the op-by-op chain of the same kernels
        if self.fused and ms.center and not ms.normalized and not (torch.is_grad_enabled() and y.requires_grad):
            if y_hat.dim() == 3:
                y_hat = y_hat.reshape(-1, y_hat.size(2))
                y = y.reshape(-1, y.size(2))
            mod = ms.stft_magnitude
            if (mod.use_fft and y_hat.shape[-1] > mod.fft_size // 2 and ms.num_mels <= 128
                    and mod._fft_tables(y_hat.device) is not None):
                total = MelFftPairLossFn.apply(y_hat, y.detach(), ms)  # power-of-two sizes: FFT in LDS (csrc/stft_fft.hip)
            else:
                total = MelPairLossFn.apply(y_hat, y.detach(), ms)
            n = y_hat.shape[0] * ms.num_mels * ms.stft_magnitude.frames(y_hat.shape[1])
            return total / n
        return Fn.l1_mean(self.mel_spectrogram(y_hat), self.mel_spectrogram(y))


class MelPairLossFn(torch.autograd.Function):
    """sum | log mel(x) - log mel(y) | with ``pwg_mel_loss_forward`` / ``_backward``: frames -> windowed DFT
    (MFMA) -> magnitude -> mel filterbank (MFMA) -> clamp -> log -> L1 for both signals in one kernel; the
    backward pass recomputes the spectrum tiles of x, applies the filterbank's transpose on MFMA and hands
    d(re)/d(im) to the data-gradient convolution (transposed DFT) and the fold's adjoint."""

    @staticmethod
    def forward(ctx, x, y, ms):
        from .. import _lib
        from ..ops import _ptr, _require_device, _stream

        mod = ms.stft_magnitude
        x = x if x.is_contiguous() else x.contiguous()
        y = y if y.is_contiguous() else y.contiguous()
        _require_device(x, y)
        b, t = x.shape
        frames = mod.frames(t)
        n_cols = frames + mod.taps - 1
        pad = mod.fold_pad
        fx = torch.empty(b, mod.hop_size, n_cols, device=x.device, dtype=torch.float32)
        fy = torch.empty_like(fx)
        L = _lib.lib()
        for src, dst in ((x, fx), (y, fy)):
            _lib.check(L.pwg_frame_fold_forward(_ptr(src), _ptr(dst), b, t, pad, mod.hop_size, n_cols, _stream()),
                       "frame_fold_forward")
        ws = torch.empty(L.pwg_mel_loss_workspace_floats(b, mod.bins, frames, ms.num_mels), device=x.device,
                         dtype=torch.float32)
        mel_x = torch.empty(b, ms.num_mels, frames, device=x.device, dtype=torch.float32)
        mel_y = torch.empty_like(mel_x)
        total = torch.empty(1, device=x.device, dtype=torch.float32)
        _lib.check(L.pwg_mel_loss_forward(_ptr(fx), _ptr(fy), _ptr(mod.pair_basis), _ptr(ms.mel_t), b, mod.hop_size,
                                          n_cols, mod.taps, mod.bins, frames, ms.num_mels, float(ms.eps),
                                          float(ms.log_div), _ptr(ws), _ptr(mel_x), _ptr(mel_y), _ptr(total),
                                          _stream()), "mel_loss_forward")
        ctx.save_for_backward(fx, mel_x, mel_y)
        ctx.ms, ctx.dims = ms, (b, t, frames, n_cols, pad)
        return total.reshape(())

    @staticmethod
    def backward(ctx, gout):
        from .. import _lib, ops
        from ..ops import _ptr, _stream

        fx, mel_x, mel_y = ctx.saved_tensors
        ms = ctx.ms
        mod = ms.stft_magnitude
        b, t, frames, n_cols, pad = ctx.dims
        gout = gout.contiguous().reshape(1)
        L = _lib.lib()
        dspec = torch.empty(b, 2 * mod.bins, frames, device=fx.device, dtype=torch.float32)
        _lib.check(L.pwg_mel_loss_backward(_ptr(fx), _ptr(mod.pair_basis), _ptr(ms.mel_b), _ptr(mel_x), _ptr(mel_y), b,
                                           mod.hop_size, n_cols, mod.taps, mod.bins, frames, ms.num_mels,
                                           float(ms.eps), float(ms.log_div), _ptr(gout), _ptr(dspec), _stream()),
                   "mel_loss_backward")
        desc = ops.make_conv_desc(b, mod.hop_size, 2 * mod.bins, n_cols, frames, mod.taps)
        if getattr(mod, "_bwd_image", None) is None or mod._bwd_image.device != fx.device:
            mod._bwd_image = ops.pack_weight_bwd(desc, mod.basis)
        dfold = ops.conv1d_backward_data(desc, dspec, mod._bwd_image)
        dx = torch.empty(b, t, device=fx.device, dtype=torch.float32)
        _lib.check(L.pwg_frame_fold_backward(_ptr(dfold), _ptr(dx), b, t, pad, mod.hop_size, n_cols, _stream()),
                   "frame_fold_backward")
        return dx, None, None


class MelFftPairLossFn(torch.autograd.Function):
    """sum | log mel(x) - log mel(y) | with ``pwg_mel_fft_loss_forward`` / ``_backward`` (csrc/stft_fft.hip): frame ->
    radix-2 FFT in LDS -> magnitude -> mel filters over their supports -> clamp -> log -> L1, both signals per frame; the
    backward pass recomputes them, gathers d|X| from the few filters covering each bin and returns through the same FFT."""

    @staticmethod
    def _tables(ms, device):
        tabs = getattr(ms, "_fft_mel_tabs", None)
        if tabs is None or tabs[0] != str(device):
            fb = ms.mel_b[: ms.num_mels].contiguous().to(device)  # [mel][bins_pad]
            nz = fb.cpu().numpy() != 0.0
            bins = ms.stft_magnitude.bins
            mel_range = np.zeros((ms.num_mels, 2), dtype=np.int32)
            for j in range(ms.num_mels):
                idx = np.nonzero(nz[j, :bins])[0]
                mel_range[j] = (idx[0], idx[-1]) if idx.size else (0, -1)
            bin_range = np.zeros((bins, 2), dtype=np.int32)
            for k in range(bins):
                cover = [j for j in range(ms.num_mels) if mel_range[j, 0] <= k <= mel_range[j, 1]]
                bin_range[k] = (cover[0], cover[-1]) if cover else (0, -1)
            tabs = (str(device), fb, torch.from_numpy(mel_range).to(device), torch.from_numpy(bin_range).to(device))
            ms._fft_mel_tabs = tabs
        return tabs[1:]

    @staticmethod
    def forward(ctx, x, y, ms):
        from .. import _lib
        from ..ops import _ptr, _require_device, _stream

        mod = ms.stft_magnitude
        x = x if x.is_contiguous() else x.contiguous()
        y = y if y.is_contiguous() else y.contiguous()
        _require_device(x, y)
        b, t = x.shape
        frames = mod.frames(t)
        window, twiddle = mod._fft_tables(x.device)
        fb, mel_range, bin_range = MelFftPairLossFn._tables(ms, x.device)
        L = _lib.lib()
        ws = torch.empty(L.pwg_stft_fft_workspace_floats(b, frames, mod.fft_size), device=x.device, dtype=torch.float32)
        total = torch.empty(1, device=x.device, dtype=torch.float32)
        ip = lambda z: ctypes.c_void_p(z.data_ptr())  # noqa: E731  (int32 tables)
        _lib.check(L.pwg_mel_fft_loss_forward(_ptr(x), _ptr(y), _ptr(window), _ptr(twiddle), _ptr(fb), ip(mel_range),
                                              ip(bin_range), b, t, mod.fft_size, mod.hop_size, mod.win_length, ms.num_mels,
                                              fb.shape[1], float(ms.eps), float(ms.log_div), _ptr(ws), _ptr(total),
                                              _stream()), "mel_fft_loss_forward")
        ctx.save_for_backward(x, y, window, twiddle, fb, mel_range, bin_range)
        ctx.ms, ctx.dims = ms, (b, t, frames)
        return total.reshape(())

    @staticmethod
    def backward(ctx, gout):
        from .. import _lib
        from ..ops import _ptr, _stream

        x, y, window, twiddle, fb, mel_range, bin_range = ctx.saved_tensors
        ms = ctx.ms
        mod = ms.stft_magnitude
        b, t, frames = ctx.dims
        gout = gout.contiguous().reshape(1)
        dframes = torch.empty(b, frames, mod.win_length, device=x.device, dtype=torch.float32)
        dx = torch.empty(b, t, device=x.device, dtype=torch.float32)
        ip = lambda z: ctypes.c_void_p(z.data_ptr())  # noqa: E731
        _lib.check(_lib.lib().pwg_mel_fft_loss_backward(_ptr(x), _ptr(y), _ptr(window), _ptr(twiddle), _ptr(fb),
                                                        ip(mel_range), ip(bin_range), b, t, mod.fft_size, mod.hop_size,
                                                        mod.win_length, ms.num_mels, fb.shape[1], float(ms.eps),
                                                        float(ms.log_div), _ptr(gout), _ptr(dframes), _ptr(dx), _stream()),
                   "mel_fft_loss_backward")
        return dx, None, None
