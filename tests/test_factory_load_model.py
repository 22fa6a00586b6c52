"""Drop-in boundary, host side: the reference's YAML schema builds the whole training object set
(``build_from_config`` mirrors /root/reference/parallel_wavegan/bin/train.py:1364-1493), and a
reference-layout checkpoint + ``config.yml`` (+ ``stats.npy``) round-trips through ``load_model``
(utils/utils.py:294-360).  CPU: construction / state-dict / alias / launcher; GPU: the loaded model
reproduces the reference's golden waveform."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import yaml

from tests.golden import synth
from tests.util import WAVE_TOL, load_golden, max_abs, synth_for

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONF = os.path.join(ROOT, "tests", "fixtures", "conf")

EXPECT = {
    "parallel_wavegan.v1": ("ParallelWaveGANGenerator", "ParallelWaveGANDiscriminator", "RAdam", "StepLR",
                            {"gen_adv", "dis_adv", "stft"}),
    "hifigan.v1": ("HiFiGANGenerator", "HiFiGANMultiScaleMultiPeriodDiscriminator", "Adam", "MultiStepLR",
                   {"gen_adv", "dis_adv", "mel", "feat_match"}),
    "multi_band_melgan.v2": ("MelGANGenerator", "MelGANMultiScaleDiscriminator", "Adam", "MultiStepLR",
                             {"gen_adv", "dis_adv", "stft", "sub_stft", "pqmf"}),
    "hifigan.v1.libritts": ("HiFiGANGenerator", "HiFiGANMultiScaleMultiPeriodDiscriminator", "Adam", "MultiStepLR",
                            {"gen_adv", "dis_adv", "mel", "feat_match"}),
    "style_melgan.v1.debug": ("StyleMelGANGenerator", "StyleMelGANDiscriminator", "Adam", "StepLR",
                              {"gen_adv", "dis_adv", "stft"}),
}


def load_conf(name):
    with open(os.path.join(CONF, name + ".yaml")) as f:
        return yaml.load(f, Loader=yaml.Loader)


@pytest.mark.parametrize("name", sorted(EXPECT))
def test_build_from_reference_yaml(name):
    from parallelwavegan_amd import optimizers
    from parallelwavegan_amd.utils import build_from_config

    cfg = load_conf(name)
    model, criterion, optimizer, scheduler = build_from_config(cfg)
    g, d, opt, sch, crit = EXPECT[name]
    assert type(model["generator"]).__name__ == g and type(model["discriminator"]).__name__ == d
    assert set(criterion) == crit
    for k in ("generator", "discriminator"):
        assert type(optimizer[k]).__name__ == opt and isinstance(optimizer[k], optimizers.fused._FusedBase)
        assert type(scheduler[k]).__name__ == sch
        assert optimizer[k].defaults["lr"] == cfg[f"{k}_optimizer_params"]["lr"]
    # the flags the Trainer reads were normalised in place, as the reference's main() does
    for flag in ("use_stft_loss", "use_subband_stft_loss", "use_feat_match_loss", "use_mel_loss"):
        assert isinstance(cfg[flag], bool)
    if name == "style_melgan.v1.debug":
        assert criterion["dis_adv"].loss_type == "hinge" and criterion["gen_adv"].loss_type == "hinge"
    if name == "hifigan.v1":
        assert criterion["feat_match"].average_by_layers is False
        assert criterion["mel"].mel_spectrogram.fft_size == 1024


def test_build_rejects_out_of_scope_types():
    from parallelwavegan_amd.utils import build_from_config

    cfg = load_conf("hifigan.v1")
    cfg["generator_type"] = "VQVAE"
    with pytest.raises(NotImplementedError):
        build_from_config(cfg)


def _write_checkpoint(tmp_path, name, seed, g_scale, with_stats):
    from parallelwavegan_amd.utils import build_from_config

    cfg = load_conf(name)
    model, _, _, _ = build_from_config(cfg, only=("model",))
    sd = synth_for(model["generator"], seed, g_scale)
    ckpt = tmp_path / "checkpoint-1000steps.pkl"
    # the reference's checkpoint layout (bin/train.py:101-125)
    torch.save({"model": {"generator": sd, "discriminator": model["discriminator"].state_dict()},
                "optimizer": {}, "scheduler": {}, "steps": 1000, "epochs": 3}, ckpt)
    cfg["format"] = "npy"
    with open(tmp_path / "config.yml", "w") as f:
        yaml.dump(cfg, f, Dumper=yaml.Dumper)
    if with_stats:
        c = cfg["generator_params"].get("in_channels", 80)
        c = cfg["generator_params"].get("aux_channels", c)
        mean = synth.synth_tensor("mean", (c,), seed).numpy()
        scale = synth.synth_tensor("scale", (c,), seed).numpy()
        np.save(tmp_path / "stats.npy", np.stack([mean, scale]))
    return str(ckpt), sd


@pytest.mark.parametrize("name", ["hifigan.v1", "parallel_wavegan.v1", "multi_band_melgan.v2"])
def test_load_model_roundtrip_cpu(tmp_path, name):
    from parallelwavegan_amd.utils import load_model

    ckpt, sd = _write_checkpoint(tmp_path, name, 3, 1.25, with_stats=True)
    model = load_model(ckpt)  # config.yml and stats.npy are found beside the checkpoint
    got = model.state_dict()
    for k, v in sd.items():
        assert torch.equal(got[k], v), k
    assert torch.allclose(model.mean, synth.synth_tensor("mean", model.mean.shape, 3))
    assert hasattr(model, "pqmf") == (name == "multi_band_melgan.v2")


def test_parallel_wavegan_alias_in_subprocess():
    code = (
        "import parallelwavegan_amd.compat as c; c.install();"
        "from parallel_wavegan.utils import load_model;"
        "from parallel_wavegan.models import HiFiGANGenerator, ParallelWaveGANGenerator, MelGANGenerator;"
        "from parallel_wavegan.losses import MultiResolutionSTFTLoss, MelSpectrogramLoss;"
        "from parallel_wavegan.layers import PQMF;"
        "from parallel_wavegan.optimizers import RAdam;"
        "from parallel_wavegan.bin.train import Trainer, Collater;"
        "import parallel_wavegan.distributed.launch as l;"
        "print(load_model.__module__, l.__name__)")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ["parallelwavegan_amd.utils.utils", "parallelwavegan_amd.distributed.launch"]


def test_launcher_spawns_ranks_and_propagates_failure(tmp_path):
    """parallelwavegan_amd.distributed.launch keeps the reference launcher's CLI
    (distributed/launch.py:15-171): env rendezvous variables, --local_rank unless --use_env."""
    script = tmp_path / "probe.py"
    script.write_text(
        "import os, sys\n"
        "r = os.environ['RANK']; open(os.path.join(sys.argv[-1], 'r' + r), 'w').write(\n"
        "    ' '.join([os.environ['WORLD_SIZE'], os.environ['LOCAL_RANK'], os.environ['MASTER_ADDR'],\n"
        "              os.environ['MASTER_PORT'], os.environ['HSA_ENABLE_IPC_MODE_LEGACY']] + sys.argv[1:-1]))\n"
        "sys.exit(3 if (r == '1' and 'fail' in sys.argv) else 0)\n")
    base = [sys.executable, "-m", "parallelwavegan_amd.distributed.launch", "--nproc_per_node", "2", "--master_port", "0"]
    ok = subprocess.run(base + [str(script), "x", str(tmp_path)], cwd=ROOT, timeout=120)
    assert ok.returncode == 0
    for r in (0, 1):
        f = (tmp_path / f"r{r}").read_text().split()
        assert f[0] == "2" and f[1] == str(r) and f[2] == "127.0.0.1" and int(f[3]) > 0 and f[4] == "0"
        assert f[5] == f"--local_rank={r}" and f[6] == "x"
    env_only = subprocess.run(base + ["--use_env", str(script), "x", str(tmp_path)], cwd=ROOT, timeout=120)
    assert env_only.returncode == 0
    assert (tmp_path / "r1").read_text().split()[5] == "x"
    bad = subprocess.run(base + ["--use_env", str(script), "fail", str(tmp_path)], cwd=ROOT, timeout=120)
    assert bad.returncode == 3


@pytest.mark.gpu
def test_load_model_reproduces_reference_golden(tmp_path, device):
    from parallelwavegan_amd.utils import load_model

    gold = load_golden("hifigan_v1_g")
    batch, frames, seed = (int(v) for v in gold["meta"])
    ckpt, _ = _write_checkpoint(tmp_path, "hifigan.v1", seed, float(gold["g_scale"]), with_stats=False)
    model = load_model(ckpt).to(device).eval()
    c = synth.synth_input("c", (batch, 80, frames), seed=seed)
    with torch.no_grad():
        assert max_abs(model(c.to(device)), gold["y"]) <= WAVE_TOL
        model.remove_weight_norm()
        assert max_abs(model.inference(c[0].transpose(0, 1).numpy()), gold["y_inference"]) <= WAVE_TOL


def test_download_pretrained_model_resolves_the_local_cache_only(tmp_path):
    """`parallel_wavegan.utils.download_pretrained_model` (utils/utils.py:363-421 of the reference) is distribution
    plumbing: the drop-in resolves a tag / Google-Drive URL that is ALREADY in the cache directory the reference's
    downloader fills and raises, instead of going to the network, for anything else."""
    from parallelwavegan_amd.utils import download_pretrained_model

    tag = "ljspeech_hifigan.v1"
    with pytest.raises(FileNotFoundError):
        download_pretrained_model(tag, download_dir=str(tmp_path))
    d = tmp_path / tag
    d.mkdir()
    (d / "config.yml").write_text("generator_type: HiFiGANGenerator\n")
    (d / "checkpoint-2500000steps.pkl").write_bytes(b"x")
    assert download_pretrained_model(tag, download_dir=str(tmp_path)) == str(d / "checkpoint-2500000steps.pkl")
    url = "https://drive.google.com/file/d/10GYvB_mIKzXzSjD67tSnBhknZRoBjsNb/view?usp=sharing"
    with pytest.raises(FileNotFoundError):
        download_pretrained_model(url, download_dir=str(tmp_path))
    d2 = tmp_path / "10GYvB_mIKzXzSjD67tSnBhknZRoBjsNb"
    d2.mkdir()
    (d2 / "checkpoint-400000steps.pkl").write_bytes(b"x")
    assert download_pretrained_model(url, download_dir=str(tmp_path)).endswith("checkpoint-400000steps.pkl")
    with pytest.raises(ValueError):
        download_pretrained_model("https://drive.google.com/short", download_dir=str(tmp_path))


def test_prepared_weights_refuse_every_bank_owned_buffer_once_stale():
    """`.scale` and `.res()` of a holder whose buffers a later WeightBank.ensure() overwrote used to be
    served silently (only `.fwd` / `.bwd()` were guarded)."""
    import torch

    from parallelwavegan_amd.functional import PreparedWeights

    w, scale = torch.zeros(4, 2, 3), torch.ones(4)
    pw = PreparedWeights(("k",), w, scale, fwd=torch.zeros(8))
    assert pw.scale is scale and pw.fwd is not None
    pw._stale = True
    for access in (lambda: pw.scale, lambda: pw.fwd, lambda: pw.res(), lambda: pw.bwd(None)):
        with pytest.raises(RuntimeError, match="overwritten by a later WeightBank.ensure"):
            access()
