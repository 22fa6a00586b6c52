"""STFT-based losses (drop-in for parallel_wavegan.losses.stft_loss)."""
import torch

from .. import functional as Fn
from .stft import STFTMagnitude


_STFT_CACHE = {}


def stft(x, fft_size, hop_size, win_length, window):
    """Magnitude spectrogram (B, #frames, fft_size // 2 + 1) of x (B, T) (losses/stft_loss.py:16-40);
    ``window`` is the window TENSOR of the reference signature (any ``win_length`` samples) or a
    window name.  The windowed DFT basis is built once per (sizes, window values)."""
    if isinstance(window, torch.Tensor):
        wkey = window.detach().cpu().numpy().tobytes()
    else:
        wkey = window
    key = (fft_size, hop_size, win_length, wkey, str(x.device))
    mod = _STFT_CACHE.get(key)
    if mod is None:
        mod = _STFT_CACHE[key] = STFTMagnitude(fft_size, hop_size, win_length, window).to(x.device)
    return mod(x).transpose(2, 1)


class SpectralConvergenceLoss(torch.nn.Module):
    """||y - x||_F / ||y||_F (reference: losses/stft_loss.py:43-61)."""

    def forward(self, x_mag, y_mag):
        return torch.sqrt(Fn.sq_diff_sum(y_mag, x_mag)) / torch.sqrt(Fn.sq_sum(y_mag))


class LogSTFTMagnitudeLoss(torch.nn.Module):
    """mean |log y - log x| (reference: losses/stft_loss.py:64-82)."""

    def forward(self, x_mag, y_mag):
        return Fn.l1_mean(Fn.LogClampFn.apply(y_mag, 0.0, 1.0), Fn.LogClampFn.apply(x_mag, 0.0, 1.0))


class STFTLoss(torch.nn.Module):
    """Single-resolution STFT loss (reference: losses/stft_loss.py:85-118)."""

    def __init__(self, fft_size=1024, shift_size=120, win_length=600, window="hann_window"):
        super().__init__()
        self.fft_size, self.shift_size, self.win_length = fft_size, shift_size, win_length
        self.spectral_convergence_loss = SpectralConvergenceLoss()
        self.log_stft_magnitude_loss = LogSTFTMagnitudeLoss()
        self.register_buffer("window", getattr(torch, window)(win_length))  # state-dict compatibility
        self.stft_magnitude = STFTMagnitude(fft_size, shift_size, win_length, window, eps=1e-7)

    fused = True  # one fused launch per resolution (csrc/stft_loss.hip); False: the op-by-op chain

    def forward(self, x, y):
        """x: predicted (B, T), y: ground truth (B, T) -> (sc_loss, mag_loss)."""
        if self.fused and not (torch.is_grad_enabled() and y.requires_grad):
            # frame -> windowed DFT (MFMA) -> magnitude -> log -> the three sums, for both signals, in one
            # kernel; its finish forms the two losses (stft_loss.py:61, :82) -- no 0-dim ATen arithmetic
            l2 = self.stft_magnitude.pair_losses(x, y)
            return l2[0], l2[1]
        x_mag = self.stft_magnitude(x)
        y_mag = self.stft_magnitude(y)
        return self.spectral_convergence_loss(x_mag, y_mag), self.log_stft_magnitude_loss(x_mag, y_mag)


class MultiResolutionSTFTLoss(torch.nn.Module):
    """Multi-resolution STFT loss (reference: losses/stft_loss.py:121-170)."""

    def __init__(self, fft_sizes=[1024, 2048, 512], hop_sizes=[120, 240, 50], win_lengths=[600, 1200, 240],
                 window="hann_window"):
        super().__init__()
        assert len(fft_sizes) == len(hop_sizes) == len(win_lengths)
        self.stft_losses = torch.nn.ModuleList(
            [STFTLoss(fs, ss, wl, window) for fs, ss, wl in zip(fft_sizes, hop_sizes, win_lengths)])

    def forward(self, x, y):
        """x, y: (B, T) or (B, #subband, T) -> (sc_loss, mag_loss), each averaged over resolutions."""
        if len(x.shape) == 3:
            x = x.reshape(-1, x.size(2))
            y = y.reshape(-1, y.size(2))
        n = len(self.stft_losses)
        if all(f.fused for f in self.stft_losses) and not (torch.is_grad_enabled() and y.requires_grad):
            # The fused resolutions hand over (sc, mag) PAIRS: add the pairs (same left-to-right order as the scalar sums
            # below), divide once, unpack once -- per resolution the scalar form costs two 0-dim adds forward and five
            # launches backward (two select-backward fills + copies and their accumulation), all in the serial chain
            # between the generator's forward and backward pass (profiles/r06_graph_gaps.txt).
            acc = None
            for f in self.stft_losses:
                l2 = f.stft_magnitude.pair_losses(x, y)
                acc = l2 if acc is None else acc + l2
            return Fn.Unpack2Fn.apply(acc / n)
        sc_loss = 0.0
        mag_loss = 0.0
        for f in self.stft_losses:
            sc_l, mag_l = f(x, y)
            sc_loss = sc_loss + sc_l
            mag_loss = mag_loss + mag_l
        return sc_loss / n, mag_loss / n
