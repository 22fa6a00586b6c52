"""``parallel-wavegan-train`` (parallelwavegan_amd.bin.train.main): the reference's command line on a synthetic
``format: npy`` dump directory -- single process, and two ranks through the package's launcher (the data-parallel
path of a real training run: DistributedSampler shards, GradReducer averages, rank 0 writes the checkpoint)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dump(tmp_path, n_utts=6, hop=64):
    rng = np.random.default_rng(3)
    for split in ("train", "dev"):
        d = tmp_path / split
        d.mkdir()
        for i in range(n_utts):
            frames = 30 + 3 * i
            np.save(d / f"utt{i}-feats.npy", rng.standard_normal((frames, 80)).astype(np.float32))
            np.save(d / f"utt{i}-wave.npy", (0.3 * rng.standard_normal(frames * hop)).astype(np.float32))
    conf = dict(
        sampling_rate=16000, hop_size=hop, num_mels=80, format="npy", batch_size=2, batch_max_steps=8 * hop,
        remove_short_samples=True, allow_cache=False, num_workers=0, pin_memory=False,
        generator_type="HiFiGANGenerator",
        generator_params=dict(in_channels=80, out_channels=1, channels=32, kernel_size=7, upsample_scales=[4, 4, 4],
                              upsample_kernel_sizes=[8, 8, 8], resblock_kernel_sizes=[3, 7],
                              resblock_dilations=[[1, 3], [1, 3]], use_additional_convs=True),
        discriminator_type="HiFiGANMultiScaleMultiPeriodDiscriminator",
        discriminator_params=dict(scales=2, periods=[2, 3],
                                  scale_discriminator_params=dict(in_channels=1, out_channels=1, kernel_sizes=[15, 41, 5, 3],
                                                                  channels=16, max_downsample_channels=64, max_groups=4,
                                                                  bias=True, downsample_scales=[2, 2],
                                                                  nonlinear_activation="LeakyReLU",
                                                                  nonlinear_activation_params=dict(negative_slope=0.1)),
                                  period_discriminator_params=dict(in_channels=1, out_channels=1, kernel_sizes=[5, 3],
                                                                   channels=8, downsample_scales=[3, 3],
                                                                   max_downsample_channels=32, bias=True,
                                                                   nonlinear_activation="LeakyReLU",
                                                                   nonlinear_activation_params=dict(negative_slope=0.1),
                                                                   use_weight_norm=True, use_spectral_norm=False)),
        use_stft_loss=False, use_mel_loss=True,
        mel_loss_params=dict(fs=16000, fft_size=256, hop_size=hop, win_length=256, window="hann", num_mels=40, fmin=0,
                             fmax=8000, log_base=None),
        use_feat_match_loss=True, lambda_aux=45.0, lambda_adv=1.0, lambda_feat_match=2.0,
        generator_optimizer_type="Adam", generator_optimizer_params=dict(lr=2e-4, betas=[0.5, 0.9], weight_decay=0.0),
        generator_scheduler_type="MultiStepLR", generator_scheduler_params=dict(gamma=0.5, milestones=[1000]),
        generator_grad_norm=-1, discriminator_optimizer_type="Adam",
        discriminator_optimizer_params=dict(lr=2e-4, betas=[0.5, 0.9], weight_decay=0.0),
        discriminator_scheduler_type="MultiStepLR", discriminator_scheduler_params=dict(gamma=0.5, milestones=[1000]),
        discriminator_grad_norm=-1, generator_train_start_steps=1, discriminator_train_start_steps=0,
        train_max_steps=8, save_interval_steps=1000, eval_interval_steps=1000, log_interval_steps=4,
        graph_warmup_steps=2)
    with open(tmp_path / "conf.yaml", "w") as f:
        yaml.dump(conf, f)
    return conf


def test_datasets_and_argument_errors(tmp_path):
    """(CPU) the npy dump is found, thresholds filter, and the command line refuses what it does not implement."""
    from parallelwavegan_amd.bin import train as T
    from parallelwavegan_amd.datasets import AudioMelDataset

    _dump(tmp_path)
    ds = AudioMelDataset(str(tmp_path / "train"), format="npy", mel_length_threshold=35, return_utt_id=True)
    assert len(ds) == 4 and ds[0][0] == "utt2"  # 30, 33 frames dropped
    utt, audio, mel = ds[1]
    assert audio.dtype == np.float32 and mel.shape == (39, 80) and len(audio) == 39 * 64
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU compute path"):
            T.main(["--train-dumpdir", str(tmp_path / "train"), "--dev-dumpdir", str(tmp_path / "dev"), "--outdir",
                    str(tmp_path / "exp"), "--config", str(tmp_path / "conf.yaml")])


@pytest.mark.gpu
def test_train_main_runs_and_checkpoints(device, tmp_path):
    from parallelwavegan_amd.bin import train as T

    _dump(tmp_path)
    out = tmp_path / "exp"
    T.main(["--train-dumpdir", str(tmp_path / "train"), "--dev-dumpdir", str(tmp_path / "dev"), "--outdir", str(out),
            "--config", str(tmp_path / "conf.yaml"), "--verbose", "0"])
    ck = torch.load(out / "checkpoint-8steps.pkl", map_location="cpu")
    assert ck["steps"] == 8 and set(ck["model"]) == {"generator", "discriminator"}
    assert all(torch.isfinite(v).all() for v in ck["model"]["generator"].values())
    assert os.path.exists(out / "config.yml")
    # resume: two more steps from the checkpoint, host Collater + DataLoader path this time
    conf = yaml.load(open(tmp_path / "conf.yaml"), Loader=yaml.Loader)
    conf.update(train_max_steps=10, device_collater=False, use_hip_graph=False)
    yaml.dump(conf, open(tmp_path / "conf2.yaml", "w"))
    T.main(["--train-dumpdir", str(tmp_path / "train"), "--dev-dumpdir", str(tmp_path / "dev"), "--outdir", str(out),
            "--config", str(tmp_path / "conf2.yaml"), "--resume", str(out / "checkpoint-8steps.pkl"), "--verbose", "0"])
    assert torch.load(out / "checkpoint-10steps.pkl", map_location="cpu")["steps"] == 10


@pytest.mark.gpu
def test_launcher_runs_two_ranks_of_the_training_command(device, tmp_path):
    """``launch.py --nproc_per_node 2 -c parallel-wavegan-train ...`` as the reference's recipes call it (run.sh
    stage 2); the two ranks share the one GPU of the test box, so the collectives go through gloo."""
    _dump(tmp_path)
    out = tmp_path / "exp"
    env = dict(os.environ, PATH=os.path.join(ROOT, "tools", "bin") + os.pathsep + os.environ.get("PATH", ""),
               PYTHONPATH=ROOT, PWG_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "parallelwavegan_amd.distributed.launch", "--nproc_per_node", "2", "--master_port", "0",
           "-c", "parallel-wavegan-train", "--train-dumpdir", str(tmp_path / "train"), "--dev-dumpdir",
           str(tmp_path / "dev"), "--outdir", str(out), "--config", str(tmp_path / "conf.yaml"), "--verbose", "0"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    ck = torch.load(out / "checkpoint-8steps.pkl", map_location="cpu")
    assert ck["steps"] == 8
    assert all(torch.isfinite(v).all() for v in ck["model"]["discriminator"].values())
