"""One-launch MRF residual unit (csrc/resunit.hip, pwg_resunit_forward) against a plain torch fp32 CPU
restatement of /root/reference/parallel_wavegan/layers/residual_block.py:253-257
(`xt = convs1[idx](x); xt = convs2[idx](xt); x = xt + x`) and against the two-launch path it replaces."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-5  # fp32, |values| = O(1); summation order differs from the reference (tap-major MFMA chain)


def _ref_unit(x, w1, b1, w2, b2, k, d, slope, add2=None, out_div=1.0):
    h = F.conv1d(F.leaky_relu(x, slope), w1, b1, padding=(k - 1) // 2 * d, dilation=d)
    if w2 is not None:
        h = F.conv1d(F.leaky_relu(h, slope), w2, b2, padding=(k - 1) // 2)
    y = h + x
    if add2 is not None:
        y = y + add2
    return y / out_div if out_div != 1.0 else y


def _where(err):
    i = int(err.argmax())
    b, c, t = err.shape
    return (i // (c * t), (i // t) % c, i % t)


@pytest.mark.parametrize("channels", [32, 64])
@pytest.mark.parametrize("kernel,dilation", [(3, 1), (3, 5), (7, 3), (11, 1), (11, 5)])
@pytest.mark.parametrize("pair", [True, False])
def test_unit_matches_reference(channels, kernel, dilation, pair):
    from parallelwavegan_amd import ops

    torch.manual_seed(channels * 100 + kernel * 10 + dilation + int(pair))
    # lengths: shorter than one tile / a few tiles with a ragged last one / many tiles
    for batch, t in [(2, 100), (3, 1000), (2, 4096 + 52)]:
        x = torch.randn(batch, channels, t)
        w1 = torch.randn(channels, channels, kernel) / (channels * kernel) ** 0.5
        w2 = torch.randn(channels, channels, kernel) / (channels * kernel) ** 0.5 if pair else None
        b1 = torch.randn(channels) * 0.1
        b2 = torch.randn(channels) * 0.1 if pair else None
        add2 = torch.randn(batch, channels, t)
        desc = ops.make_resunit_desc(batch, channels, t, kernel, dilation, pair, 0.1, 0.1, 3.0)
        assert ops.resunit_supported(desc)
        dev = torch.device("cuda:0")
        w1p = ops.resunit_pack_weight(w1.to(dev))
        w2p = ops.resunit_pack_weight(w2.to(dev)) if pair else None
        y = ops.resunit_forward(desc, x.to(dev), w1p, b1.to(dev), w2p, None if b2 is None else b2.to(dev), add2.to(dev))
        ref = _ref_unit(x, w1, b1, w2, b2, kernel, dilation, 0.1, add2, 3.0)
        err = (y.cpu() - ref).abs()
        assert err.max() < TOL, f"T={t}: max err {err.max():.3e} at (b,c,t)={_where(err)}"
        # no bias, no addend, no division
        desc = ops.make_resunit_desc(batch, channels, t, kernel, dilation, pair, 0.1, 0.1, 1.0)
        y = ops.resunit_forward(desc, x.to(dev), w1p, None, w2p, None, None)
        ref = _ref_unit(x, w1, None, w2, None, kernel, dilation, 0.1)
        err = (y.cpu() - ref).abs()
        assert err.max() < TOL, f"T={t} (plain): max err {err.max():.3e} at (b,c,t)={_where(err)}"


def test_weight_norm_scale_folds_into_the_image():
    from parallelwavegan_amd import ops

    torch.manual_seed(5)
    dev = torch.device("cuda:0")
    c, k, d, t = 32, 7, 3, 2048
    x = torch.randn(2, c, t)
    v = torch.randn(c, c, k)
    g = torch.rand(c) + 0.5
    scale = g / v.reshape(c, -1).norm(dim=1)
    w = v * scale[:, None, None]
    desc = ops.make_resunit_desc(2, c, t, k, d, False, 0.1, 0.1, 1.0)
    y = ops.resunit_forward(desc, x.to(dev), ops.resunit_pack_weight(v.to(dev), scale.to(dev)), None)
    ref = _ref_unit(x, w, None, None, None, k, d, 0.1)
    assert (y.cpu() - ref).abs().max() < TOL


def test_block_uses_one_launch_per_unit_and_matches_two_launch_path():
    """HiFiGANResidualBlock under no_grad: 3 launches (one per dilation) instead of 6, same values as the
    separate convolutions; with gradients enabled the block keeps the separate (differentiable) path."""
    from parallelwavegan_amd import ops
    from parallelwavegan_amd.layers import HiFiGANResidualBlock

    torch.manual_seed(11)
    dev = torch.device("cuda:0")
    for channels, kernel in [(64, 7), (32, 11), (32, 3)]:  # (64, 11): supported but not profitable
        blk = HiFiGANResidualBlock(kernel, channels, (1, 3, 5)).to(dev)
        for m in blk.modules():
            if hasattr(m, "apply_weight_norm"):
                m.apply_weight_norm()
        x = torch.randn(2, channels, 8192, device=dev)
        acc = torch.randn_like(x)
        with torch.no_grad():
            with ops.profile() as prof:
                y1 = blk(x, accum=acc, out_div=3.0)
            fams = {k_: v["launches"] for k_, v in prof.results.items()}
            assert fams.get("resunit_kernel", 0) == 3 and not any("conv1d" in k_ for k_ in fams), fams
            blk.fuse_units = False
            y2 = blk(x, accum=acc, out_div=3.0)
            blk.fuse_units = True
        assert (y1 - y2).abs().max() < TOL
        xg = x.clone().requires_grad_(True)
        with ops.profile() as prof:
            blk(xg).sum().backward()
        assert not any("resunit" in k_ for k_ in prof.results), list(prof.results)


def test_unsupported_units_fall_back_to_separate_convolutions():
    from parallelwavegan_amd import ops

    # 128 channels, dilation too wide for the resident tile, length not a multiple of 4
    assert not ops.resunit_supported(ops.make_resunit_desc(2, 128, 4096, 3, 1))
    assert not ops.resunit_supported(ops.make_resunit_desc(2, 64, 4096, 11, 9))
    assert not ops.resunit_supported(ops.make_resunit_desc(2, 32, 4098, 3, 1))
    d = ops.make_resunit_desc(2, 64, 4096, 11, 5)
    assert ops.resunit_supported(d) and not ops.resunit_profitable(d)
    dev = torch.device("cuda:0")
    x = torch.randn(1, 32, 64, device=dev)
    with pytest.raises(RuntimeError):
        ops.resunit_forward(ops.make_resunit_desc(1, 32, 64, 4, 1), x, torch.zeros(4 * 32 * 32, device=dev), None,
                            torch.zeros(4 * 32 * 32, device=dev), None)
