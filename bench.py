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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="utterances per step per GPU")
    ap.add_argument("--frames", type=int, default=800, help="mel frames per utterance (800 = 9.3 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank)
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

    with torch.no_grad():
        for _ in range(args.warmup):
            y = g(c)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = g(c)
        barrier()
        elapsed = time.perf_counter() - t0
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
        roofline = {
            "kernel": name,
            "bound": "mfma",
            "achieved": achieved,
            "peak": FP32_MATRIX_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / FP32_MATRIX_PEAK_TFLOPS,
            "traffic": None,
            "launches_per_step": r["launches"] / prof_steps,
            "avg_launch_us": r["ms"] * 1e3 / r["launches"],
            "flops_per_launch": r["flops"] / r["launches"],
            "algorithmic_bytes_per_launch": r["bytes"] / r["launches"],
            "algorithmic_GBps": r["bytes"] / (r["ms"] * 1e-3) / 1e9,
            "kernel_ms_per_step": r["ms"] / prof_steps,
            "share_of_step_kernel_time": r["ms"] / sum(v["ms"] for v in prof.results.values()),
        }

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
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
