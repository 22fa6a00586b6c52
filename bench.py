#!/usr/bin/env python
"""bench.py -- headline benchmark: HiFi-GAN V1 (22.05 kHz) generator inference RTF^-1
(audio samples / second) on MI355X, plus the same-box CPU baseline.

Contract (one JSON line on stdout from rank 0):
  python bench.py --gpus N --steps K --warmup W
A "step" is one forward pass of the generator over one resident batch of
synthetic mel frames (B utterances x F frames, mel ~ N(0,1), random-init weights
of the V1 architecture, weight norm removed as bin/decode.py does).  N > 1 runs
one replica per GPU on its own batch (utterances are independent: "replicas
only", no data-path collective) -> weak scaling; value = total samples / max
time over ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HIFIGAN_V1 = dict(
    in_channels=80, out_channels=1, channels=512, kernel_size=7, upsample_scales=[8, 8, 2, 2],
    upsample_kernel_sizes=[16, 16, 4, 4], resblock_kernel_sizes=[3, 7, 11],
    resblock_dilations=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], use_additional_convs=True, bias=True,
    nonlinear_activation="LeakyReLU", nonlinear_activation_params={"negative_slope": 0.1}, use_weight_norm=True,
)
FP32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
HBM_PEAK_GBS = 8000.0


def hifigan_macs_per_sample(cfg):
    """Algorithmic multiply-accumulates per output sample (1 199 424)."""
    up = 1
    for s in cfg["upsample_scales"]:
        up *= s
    ch = cfg["channels"]
    macs = cfg["in_channels"] * ch * cfg["kernel_size"] / up  # input conv runs at the frame rate
    rate = 1.0 / up
    for s, k in zip(cfg["upsample_scales"], cfg["upsample_kernel_sizes"]):
        macs += ch * (ch // 2) * k * rate  # ConvTranspose1d: Cin*Cout*k per INPUT sample
        ch //= 2
        rate *= s
        for ks, dil in zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilations"]):
            n_convs = len(dil) * (2 if cfg["use_additional_convs"] else 1)
            macs += n_convs * ch * ch * ks * rate
    macs += ch * cfg["out_channels"] * cfg["kernel_size"] * rate
    return macs


def cpu_baseline(budget_s=12.0):
    """The oracle (torch CPU restatement of the reference's ATen call sequence) timed on this
    box's host cores: B=1, 100 mel frames per call (bin/decode.py is utterance-at-a-time)."""
    from oracle import torch_cpu
    from parallelwavegan_amd.models import HiFiGANGenerator

    cores = os.cpu_count() or 1
    g = HiFiGANGenerator(**HIFIGAN_V1)
    g.remove_weight_norm()
    sd = {k: v.detach() for k, v in g.state_dict().items()}
    frames = 100
    c = torch.randn(1, 80, frames)

    def once():
        t0 = time.time()
        y = torch_cpu.hifigan_generator(sd, c, **HIFIGAN_V1)
        return time.time() - t0, y

    with torch.no_grad():
        # torch's default (= all cores) oversubscribes these small convs on a many-core host, so
        # probe a few intra-op thread counts briefly and keep the fastest for the timed sample
        probe = {}
        for nt in sorted({min(cores, n) for n in (8, 16, 32, 64)}):
            torch.set_num_threads(nt)
            once()
            probe[nt] = min(once()[0], once()[0])
        nthreads = min(probe, key=probe.get)
        torch.set_num_threads(nthreads)
        best, n, t_start = float("inf"), 0, time.time()
        while n < 3 or (time.time() - t_start < budget_s and n < 200):
            dt, y = once()
            best = min(best, dt)
            n += 1
    return {
        "value": y.numel() / best,
        "unit": "samples/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"oracle.torch_cpu.hifigan_generator, B=1 x {frames} frames, best of {n} calls, "
                  f"{nthreads} of {cores} host threads (fastest of {sorted(probe)})",
    }



HIFIGAN_V1_D = dict(
    scales=3, scale_downsample_pooling="AvgPool1d",
    scale_downsample_pooling_params=dict(kernel_size=4, stride=2, padding=2),
    scale_discriminator_params=dict(in_channels=1, out_channels=1, kernel_sizes=[15, 41, 5, 3], channels=128,
                                    max_downsample_channels=1024, max_groups=16, bias=True,
                                    downsample_scales=[4, 4, 4, 4, 1], nonlinear_activation="LeakyReLU",
                                    nonlinear_activation_params=dict(negative_slope=0.1)),
    follow_official_norm=True, periods=[2, 3, 5, 7, 11],
    period_discriminator_params=dict(in_channels=1, out_channels=1, kernel_sizes=[5, 3], channels=32,
                                     downsample_scales=[3, 3, 3, 3, 1], max_downsample_channels=1024, bias=True,
                                     nonlinear_activation="LeakyReLU",
                                     nonlinear_activation_params=dict(negative_slope=0.1), use_weight_norm=True,
                                     use_spectral_norm=False),
)
MEL_LOSS = dict(fs=22050, fft_size=1024, hop_size=256, win_length=None, window="hann", num_mels=80, fmin=0,
                fmax=11025, log_base=None)
# fwd = 1x, bwd = 2x, incl. the second no-grad G pass and the no-grad D(real) pass:
# per item 4 * G(32 frames) + 10 * D(8192 samples)
TRAIN_GFLOP_PER_ITEM = 4 * 19.65 + 10 * 12.08


def cpu_train_baseline(budget_steps=2):
    """oracle.train_step (torch CPU restatement of Trainer._train_step) on a B=2 slice of the
    same workload; reported as steps/s scaled to the full batch of 16."""
    from oracle.train_step import HiFiGANTrainState
    from parallelwavegan_amd.models import HiFiGANGenerator, HiFiGANMultiScaleMultiPeriodDiscriminator

    cores = os.cpu_count() or 1
    nthreads = min(cores, 32)
    torch.set_num_threads(nthreads)
    g = HiFiGANGenerator(**HIFIGAN_V1)
    d = HiFiGANMultiScaleMultiPeriodDiscriminator(**HIFIGAN_V1_D)
    st = HiFiGANTrainState({k: v.detach() for k, v in g.state_dict().items()},
                           {k: v.detach() for k, v in d.state_dict().items()}, HIFIGAN_V1, HIFIGAN_V1_D, MEL_LOSS)
    b = 2
    c, y = torch.randn(b, 80, 32), 0.3 * torch.randn(b, 1, 8192)
    st.step(c, y)  # warm-up
    t0 = time.time()
    for _ in range(budget_steps):
        st.step(c, y)
    dt = (time.time() - t0) / budget_steps
    return {
        "value": 1.0 / (dt * 16 / b),
        "unit": "steps/s (B=16 x 8192, extrapolated linearly from B=2)",
        "cores": nthreads,
        "kind": "port",
        "sample": f"oracle.train_step.HiFiGANTrainState.step, B={b} x 8192 samples, {budget_steps} timed steps "
                  f"({dt:.2f} s each), {nthreads} of {cores} host threads",
    }


def bench_train(args, dev, rank, world, dist):
    """HiFi-GAN V1 LJSpeech training step (configs[2] / C3): B=16 x 8192 samples per GPU, both the
    generator and the discriminator phase active, mel loss + adversarial + feature matching, Adam."""
    import tempfile

    from parallelwavegan_amd import losses, ops, optimizers
    from parallelwavegan_amd.bin.train import Trainer
    from parallelwavegan_amd.models import HiFiGANGenerator, HiFiGANMultiScaleMultiPeriodDiscriminator

    torch.manual_seed(4321)
    model = {"generator": HiFiGANGenerator(**HIFIGAN_V1).to(dev),
             "discriminator": HiFiGANMultiScaleMultiPeriodDiscriminator(**HIFIGAN_V1_D).to(dev)}
    criterion = {
        "gen_adv": losses.GeneratorAdversarialLoss(average_by_discriminators=False),
        "dis_adv": losses.DiscriminatorAdversarialLoss(average_by_discriminators=False),
        "mel": losses.MelSpectrogramLoss(**MEL_LOSS).to(dev),
        "feat_match": losses.FeatureMatchLoss(average_by_discriminators=False, average_by_layers=False,
                                              include_final_outputs=False),
    }
    opt = {k: optimizers.Adam(model[k].parameters(), lr=2.0e-4, betas=(0.5, 0.9), weight_decay=0.0)
           for k in ("generator", "discriminator")}
    sched = {k: optimizers.lr_scheduler.MultiStepLR(opt[k], gamma=0.5, milestones=[200000, 400000, 600000, 800000])
             for k in ("generator", "discriminator")}
    steps, warmup = args.train_steps, args.train_warmup
    config = dict(generator_type="HiFiGANGenerator", generator_params=HIFIGAN_V1, use_stft_loss=False,
                  use_subband_stft_loss=False, use_mel_loss=True, use_feat_match_loss=True, lambda_aux=45.0,
                  lambda_adv=1.0, lambda_feat_match=2.0, generator_grad_norm=-1, discriminator_grad_norm=-1,
                  generator_train_start_steps=0, discriminator_train_start_steps=0,
                  train_max_steps=10 ** 9, save_interval_steps=10 ** 9, eval_interval_steps=10 ** 9,
                  log_interval_steps=10 ** 9, distributed=world > 1, rank=rank, outdir=tempfile.mkdtemp(),
                  progress=False, use_hip_graph=not args.no_graph, graph_warmup_steps=2,
                  reuse_real_discriminator_pass=os.environ.get("PWG_REUSE_REAL", "1") == "1")
    gen = torch.Generator(device="cpu").manual_seed(200 + rank)
    b, t = args.train_batch, 8192
    c = torch.randn(b, 80, t // 256, generator=gen).to(dev)
    y = (0.3 * torch.randn(b, 1, t, generator=gen)).to(dev)
    batch = ((c,), y)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # hipGraph replay of the step (data parallel: graphs cut at the gradient exchanges); should the
    # capture fail on some stack, every rank sees the same exception and the run continues eagerly
    for use_graph in ([True, False] if config["use_hip_graph"] else [False]):
        config["use_hip_graph"] = use_graph
        tr = Trainer(steps=1, epochs=0, data_loader={"train": [batch], "dev": [batch]},
                     sampler={"train": None, "dev": None}, model=model, criterion=criterion, optimizer=opt,
                     scheduler=sched, config=config, device=dev)
        tr.tqdm = None
        try:
            for _ in range(warmup):
                tr._train_step(batch)
            break
        except Exception as e:  # noqa: BLE001
            if not use_graph:
                raise
            print(f"[bench] hipGraph training step failed ({type(e).__name__}: {e}); falling back to eager",
                  file=sys.stderr, flush=True)
            for r in (tr.reducers or {}).values():
                r.remove()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr._train_step(batch)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    tr._flush_pending()
    finite = all(v == v and abs(v) != float("inf") for v in tr.total_train_loss.values())
    out = None
    # one more (eager) step with per-kernel event timing; every rank runs it -- a data-parallel step
    # contains collectives -- but only rank 0 reports
    tr.config["use_hip_graph"] = False
    for m in model.values():  # serial launches: concurrent branches would inflate each kernel's event time
        for sub in m.modules():
            if hasattr(sub, "branch_streams"):
                sub.branch_streams = False
    with ops.profile() as prof:
        tr._train_step(batch)
    if rank == 0:
        tot_ms = sum(v["ms"] for v in prof.results.values())
        kern = {k: {"ms_per_step": round(v["ms"], 3), "launches": v["launches"],
                    "TFLOPs": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["flops"] else None,
                    "GBps_algorithmic": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)}
                for k, v in sorted(prof.results.items(), key=lambda kv: -kv[1]["ms"])}
        flops_step = TRAIN_GFLOP_PER_ITEM * 1e9 * b
        out = {
            "metric": "HiFi-GAN V1 LJSpeech training steps/s (G+D phases, B=16 x 8192 per GPU)",
            "value": steps / elapsed,
            "unit": "steps/s",
            "ms_per_step": elapsed / steps * 1e3,
            "steps": steps,
            "warmup": warmup,
            "batch_per_gpu": b,
            "global_batch": b * world,
            "segments_per_s": b * world * steps / elapsed,
            "scaling": "weak",
            "parallelism": f"dp{world}" if world > 1 else "single",
            "hip_graph": bool(tr._graphs),
            "losses_finite": finite,
            "algorithmic_TFLOP_per_step_per_gpu": flops_step / 1e12,
            "achieved_TFLOPs_per_gpu": flops_step / (elapsed / steps) / 1e12,
            "frac_of_fp32_matrix_peak": flops_step / (elapsed / steps) / 1e12 / FP32_MATRIX_PEAK_TFLOPS,
            "kernel_time_ms_per_step": round(tot_ms, 3),
            "kernels": kern,
        }
    for r in (tr.reducers or {}).values():
        r.remove()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="utterances per step per GPU")
    ap.add_argument("--frames", type=int, default=800, help="mel frames per utterance (800 = 9.3 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the training-step measurement")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-utterance latency runs")
    ap.add_argument("--train-steps", type=int, default=50)
    ap.add_argument("--train-warmup", type=int, default=10, help=">= 3 so that the hipGraph capture is not timed")
    ap.add_argument("--no-graph", action="store_true", help="run the training step eagerly (no hipGraph replay)")
    ap.add_argument("--train-batch", type=int, default=16)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm; PWG_DIST_BACKEND=gloo lets two ranks share one GPU for smoke tests
        dist.init_process_group(os.environ.get("PWG_DIST_BACKEND", "nccl"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)

    from parallelwavegan_amd import ops
    from parallelwavegan_amd.models import HiFiGANGenerator

    torch.manual_seed(1234)
    g = HiFiGANGenerator(**HIFIGAN_V1)
    g.remove_weight_norm()  # as bin/decode.py:147
    g = g.to(dev).eval()
    gen = torch.Generator(device="cpu").manual_seed(100 + rank)
    c = torch.randn(args.batch, 80, args.frames, generator=gen).to(dev)
    samples_per_step = args.batch * args.frames * g.upsample_factor

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the forward is replayed as one hipGraph (the 3 MRF blocks of a stage are parallel branches)
    from parallelwavegan_amd.graphs import GraphedInference

    g.branch_streams = not args.no_graph
    run = g if args.no_graph else GraphedInference(g)
    with torch.no_grad():
        for _ in range(args.warmup):
            y = run(c)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = run(c)
        barrier()
        elapsed = time.perf_counter() - t0
    g.branch_streams = False
    assert torch.isfinite(y).all()
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    # ---- dominant-kernel roofline: HIP events recorded inside the library on the launch stream
    # around every kernel of `prof_steps` further steps (pwg_prof_*, include/pwg_kernels.h)
    roofline = None
    if rank == 0:
        prof_steps = min(args.steps, 3)
        with ops.profile() as prof, torch.no_grad():
            for _ in range(prof_steps):
                g(c)
        name, r = max(prof.results.items(), key=lambda kv: kv[1]["ms"])
        achieved = r["flops"] / (r["ms"] * 1e-3) / 1e12
        traffic, traffic_src = None, None
        pmc_file = os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")
        if os.path.exists(pmc_file) and args.batch == 16 and args.frames == 800:
            # HBM bytes per launch of this kernel on THIS workload, from separate rocprofv3 --pmc passes
            # (FETCH_SIZE, WRITE_SIZE; KiB units; read side calibrated on a known byte count) --
            # counters cannot be read from inside the process, so the committed summary is cited
            with open(pmc_file) as f:
                pmc = json.load(f)
            if pmc.get("kernel") == name:
                traffic, traffic_src = pmc["hbm_bytes_per_launch"], "profiles/r01_pmc_hbm_traffic.json"
        roofline = {
            "kernel": name,
            "bound": "mfma",
            "achieved": achieved,
            "peak": FP32_MATRIX_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / FP32_MATRIX_PEAK_TFLOPS,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "launches_per_step": r["launches"] / prof_steps,
            "avg_launch_us": r["ms"] * 1e3 / r["launches"],
            "flops_per_launch": r["flops"] / r["launches"],
            "algorithmic_bytes_per_launch": r["bytes"] / r["launches"],
            "algorithmic_GBps": r["bytes"] / (r["ms"] * 1e-3) / 1e9,
            "kernel_ms_per_step": r["ms"] / prof_steps,
            "share_of_step_kernel_time": r["ms"] / sum(v["ms"] for v in prof.results.values()),
        }

    # single-utterance latency (bin/decode.py's regime: batch 1), same graph-replay path
    latency = None
    if rank == 0 and not args.no_latency:
        latency = {}
        g.branch_streams = not args.no_graph
        for frames in (100, 800):
            c1 = torch.randn(1, 80, frames, generator=gen).to(dev)
            run1 = g if args.no_graph else GraphedInference(g)
            with torch.no_grad():
                for _ in range(3):
                    run1(c1)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(20):
                    run1(c1)
                torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / 20
            latency[f"B1_F{frames}"] = {"ms": dt * 1e3, "samples_per_s": frames * g.upsample_factor / dt,
                                        "rtf": dt / (frames * g.upsample_factor / 22050.0)}
            del run1
        g.branch_streams = False
    del y, run
    torch.cuda.empty_cache()
    train = None if args.no_train else bench_train(args, dev, rank, world, dist)

    if rank == 0:
        value = samples_per_step * world * args.steps / elapsed
        out = {
            "metric": "HiFi-GAN V1 22.05 kHz generator inference RTF^-1 (audio samples/s)",
            "value": value,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"HiFi-GAN V1 generator forward (configs[2], LJSpeech 22.05 kHz), "
                            f"{args.batch} utterances x {args.frames} mel frames per GPU per step, fp32, "
                            f"weight norm removed, random-init weights",
                "batch_per_gpu": args.batch,
                "frames": args.frames,
                "samples_per_step_per_gpu": samples_per_step,
                "parallelism": f"replicas x{world}",
            },
            "rtf": 22050.0 / value,
            "hip_graph": not args.no_graph,
            "latency": latency,
            "roofline": roofline,
        }
        if train is not None:
            out["train"] = train
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            if train is not None:
                train["cpu_baseline"] = cpu_train_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
