"""GPU parity: HIP HiFi-GAN generator vs the reference's golden vectors and the oracle."""
import pytest
import torch

from oracle import torch_cpu
from parallelwavegan_amd.models import HiFiGANGenerator
from tests.golden import synth
from tests.test_oracle_golden import HIFIGAN_CASES
from tests.util import WAVE_TOL, load_golden, max_abs, synth_for

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,cfg", HIFIGAN_CASES)
def test_generator_matches_reference_golden(name, cfg, device):
    gold = load_golden(name)
    batch, frames, seed = (int(v) for v in gold["meta"])
    g = HiFiGANGenerator(**cfg)
    g.load_state_dict(synth_for(g, seed, float(gold["g_scale"])))
    g = g.to(device).eval()
    c = synth.synth_input("c", (batch, cfg["in_channels"], frames), seed=seed)
    with torch.no_grad():
        y = g(c.to(device))
        assert max_abs(y, gold["y"]) <= WAVE_TOL
        # remove_weight_norm + (T, C) inference API, as bin/decode.py uses it
        g.remove_weight_norm()
        y_inf = g.inference(c[0].transpose(0, 1).numpy())
        assert y_inf.shape == (frames * g.upsample_factor, 1)
        assert max_abs(y_inf, gold["y_inference"]) <= WAVE_TOL


@pytest.mark.parametrize("frames,batch", [(1, 1), (7, 2), (100, 1), (33, 3)])
def test_generator_matches_oracle_various_lengths(frames, batch, device):
    cfg = synth.HIFIGAN_V1
    g = HiFiGANGenerator(**cfg)
    sd = synth_for(g, 5, 1.25)
    g.load_state_dict(sd)
    g = g.to(device).eval()
    c = synth.synth_input("c", (batch, 80, frames), seed=frames)
    with torch.no_grad():
        y = g(c.to(device))
        ref = torch_cpu.hifigan_generator(sd, c, **cfg)
    assert y.shape == ref.shape
    assert max_abs(y, ref) <= WAVE_TOL


def test_generator_at_benchmark_length_matches_oracle(device):
    """bench.py's utterance length (800 frames = 204 800 samples per item): the long-T launches pick the
    wide tile configurations (128 x 128 / 128 x 256 column tiles) that short test inputs never reach."""
    cfg = synth.HIFIGAN_V1
    g = HiFiGANGenerator(**cfg)
    sd = synth_for(g, 9, 1.25)
    g.load_state_dict(sd)
    g.remove_weight_norm()
    sd = {k: v.detach().clone() for k, v in g.state_dict().items()}
    g = g.to(device).eval()
    c = synth.synth_input("c", (2, 80, 800), seed=800)
    with torch.no_grad():
        y = g(c.to(device))
        ref = torch_cpu.hifigan_generator(sd, c, **cfg)
    assert y.shape == ref.shape == (2, 1, 800 * 256)
    assert max_abs(y, ref) <= WAVE_TOL


def test_chained_branch_ends_equal_the_serial_running_sum(device):
    """Inference with MRF branches on parallel streams whose last kernels are chained (streams.run_branches_chained)
    is bit-identical to the serial running sum `cs += block(c); c = cs / num_blocks` (reference
    models/hifigan.py:186-190) and to the fork / combine-kernel variant."""
    from parallelwavegan_amd.models import HiFiGANGenerator

    torch.manual_seed(3)
    g = HiFiGANGenerator(channels=128, upsample_scales=(4, 4), upsample_kernel_sizes=(8, 8)).to(device).eval()
    c = torch.randn(2, 80, 64, device=device)
    with torch.no_grad():
        serial = g(c)
        g.branch_streams = True
        g.chain_min_elems = 0
        chained = g(c)
        g.chain_min_elems = 1 << 62
        forked = g(c)
    torch.cuda.synchronize()
    assert torch.equal(chained, serial)
    assert torch.equal(forked, serial)
