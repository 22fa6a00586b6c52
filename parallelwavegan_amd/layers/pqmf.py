"""Pseudo-QMF filterbank (drop-in for parallel_wavegan.layers.pqmf).

The reference runs analysis as a full-rate 63-tap FIR (1 -> K channels) followed by a one-hot
stride-K "pick" convolution, and synthesis as a one-hot zero-stuffing transposed convolution
followed by a K -> 1 FIR (layers/pqmf.py:120-149), i.e. it computes K x the samples it keeps.
Here each direction is ONE launch of a dedicated HBM-bound polyphase kernel (csrc/pqmf.hip, K <= 8):
  analysis : y[b, k, i] = sum_j h_analysis[k, j] x[b, i K + j - taps/2],   i < floor(T / K)  (any T, as the reference)
  synthesis: x[b, t]    = sum_k sum_i K h_synthesis[k, taps/2 + i K - t] y[b, k, i]
which produce exactly the kept samples; each kernel is the other's adjoint, so the backward passes are the same two
launches.  More than 8 sub-bands run the general convolution kernel in the same polyphase form.  Filter design follows
the same published formulas (Kaiser-windowed sinc prototype, cosine modulation).
"""
import numpy as np
import torch

from .. import functional as Fn


def design_prototype_filter(taps=62, cutoff_ratio=0.142, beta=9.0):
    """Kaiser-window prototype low-pass of length taps + 1 (float64)."""
    assert taps % 2 == 0, "The number of taps mush be even number."
    assert 0.0 < cutoff_ratio < 1.0, "Cutoff ratio must be > 0.0 and < 1.0."
    n = np.arange(taps + 1) - 0.5 * taps
    omega_c = np.pi * cutoff_ratio
    with np.errstate(invalid="ignore", divide="ignore"):
        h = np.sin(omega_c * n) / (np.pi * n)
    h[taps // 2] = cutoff_ratio  # limit of sin(w n)/(pi n) at n = 0
    return h * np.kaiser(taps + 1, beta)


class PQMF(torch.nn.Module):
    def __init__(self, subbands=4, taps=62, cutoff_ratio=0.142, beta=9.0):
        super().__init__()
        h_proto = design_prototype_filter(taps, cutoff_ratio, beta)
        n = np.arange(taps + 1) - (taps / 2)
        k = np.arange(subbands)[:, None]
        phase = (2 * k + 1) * (np.pi / (2 * subbands)) * n[None, :]
        sign = ((-1.0) ** k) * np.pi / 4
        h_analysis = 2 * h_proto[None, :] * np.cos(phase + sign)
        h_synthesis = 2 * h_proto[None, :] * np.cos(phase - sign)
        # buffers with the reference's names / shapes (state-dict compatible)
        self.register_buffer("analysis_filter", torch.from_numpy(h_analysis).float().unsqueeze(1))
        self.register_buffer("synthesis_filter", torch.from_numpy(h_synthesis).float().unsqueeze(0))
        updown = torch.zeros((subbands, subbands, subbands)).float()
        for i in range(subbands):
            updown[i, i, 0] = 1.0
        self.register_buffer("updown_filter", updown)
        self.subbands, self.taps = subbands, taps
        # transposed-conv weight (C_in = subbands, C_out = 1, k): K * h_synthesis flipped in time
        syn = (subbands * self.synthesis_filter[0].flip(-1)).unsqueeze(1).contiguous()
        self.register_buffer("_synthesis_weight", syn, persistent=False)
        self._fused = dict(pre_act=None, pre_slope=0.0, post_act=None, post_slope=0.0, out_mul=1.0, out_div=1.0)

    def _geom(self, transposed):
        return dict(kernel=self.taps + 1, stride=self.subbands, dilation=1, padding=self.taps // 2, groups=1,
                    transposed=transposed, output_padding=self.subbands - 1, width=1, pad_mode="zero")

    def analysis(self, x):
        """(B, 1, T) -> (B, subbands, T // subbands); T need not be a multiple of the number of sub-bands (the
        reference filters at full rate and keeps samples 0, K, 2K, ... that have a full stride: pqmf.py:120-131)."""
        n_out = x.shape[-1] // self.subbands
        if n_out < 1:
            raise ValueError("PQMF.analysis: the signal is shorter than one sub-band sample")
        if self.subbands <= 8:
            return Fn.PQMFDownFn.apply(x, self.analysis_filter[:, 0], n_out, self.taps // 2)
        y = Fn.FusedConvFn.apply(x, self.analysis_filter, None, None, None, self._geom(False), self._fused, None)
        return y if y.shape[-1] == n_out else y[..., :n_out].contiguous()

    def synthesis(self, x):
        """(B, subbands, T // subbands) -> (B, 1, T)."""
        if self.subbands <= 8:
            return Fn.PQMFUpFn.apply(x, self._synthesis_weight[:, 0], x.shape[-1] * self.subbands, self.taps // 2)
        return Fn.FusedConvFn.apply(x, self._synthesis_weight, None, None, None, self._geom(True), self._fused, None)
