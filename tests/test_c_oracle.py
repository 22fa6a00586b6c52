"""CPU: the plain-C restatement agrees with the torch-level oracle (two independent statements
of the same ATen semantics)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import c_oracle, torch_cpu


@pytest.mark.parametrize("cfg", [
    dict(B=2, Cin=6, Cout=8, T=50, K=3, stride=1, dilation=3, padding=3, groups=1, slope=0.1),
    dict(B=1, Cin=8, Cout=16, T=70, K=41, stride=4, dilation=1, padding=20, groups=4, slope=0.2),
    dict(B=2, Cin=1, Cout=4, T=64, K=15, stride=1, dilation=1, padding=7, groups=1, slope=None),
])
def test_conv1d(cfg):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(cfg["B"], cfg["Cin"], cfg["T"], generator=g)
    w = torch.randn(cfg["Cout"], cfg["Cin"] // cfg["groups"], cfg["K"], generator=g)
    b = torch.randn(cfg["Cout"], generator=g)
    xin = F.leaky_relu(x, cfg["slope"]) if cfg["slope"] is not None else x
    ref = F.conv1d(xin, w, b, cfg["stride"], cfg["padding"], cfg["dilation"], cfg["groups"]).numpy()
    got = c_oracle.conv1d(x.numpy(), w.numpy(), b.numpy(), cfg["stride"], cfg["dilation"], cfg["padding"],
                          cfg["groups"], cfg["slope"])
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()


@pytest.mark.parametrize("s,k,p,op", [(8, 16, 4, 0), (5, 10, 3, 1), (4, 63, 31, 3)])
def test_conv_transpose1d(s, k, p, op):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 6, 19, generator=g)
    w = torch.randn(6, 4, k, generator=g)
    b = torch.randn(4, generator=g)
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1), w, b, stride=s, padding=p, output_padding=op).numpy()
    got = c_oracle.conv_transpose1d(x.numpy(), w.numpy(), b.numpy(), s, p, op, slope=0.1)
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()


def test_avg_pool_and_stft():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 2, 101, generator=g)
    for k, s, p, cip in [(4, 2, 2, True), (4, 2, 1, False)]:
        ref = F.avg_pool1d(x, k, s, p, count_include_pad=cip).numpy()
        assert np.abs(c_oracle.avg_pool1d(x.numpy(), k, s, p, cip) - ref).max() <= 1e-6
    y = torch.randn(1, 700, generator=g)
    for n_fft, hop, win in [(128, 30, 75), (171, 10, 60), (64, 16, 64)]:
        ref = torch_cpu.stft_magnitude(y, n_fft, hop, win)[0].numpy()
        got = c_oracle.stft_mag(y[0].numpy(), n_fft, hop, win)
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max()
