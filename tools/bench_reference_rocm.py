"""The reference's OWN package (staged copy oracle/_ref) on the MI355X through stock PyTorch-ROCm (ATen + MIOpen):
the second, non-graded comparison SURVEY.md s8(d) asks for.  HiFi-GAN V1 generator inference at the headline batch
(B16 x 800 frames, weight norm removed, eval, fp32, no grad), timed with HIP events after warm-up.  Progress lines are
flushed as they happen so that a run cut by its time limit still reports how far it got (MIOpen has no precompiled
kernel database for gfx950 in this image: the first call of every distinct convolution compiles its kernel).

usage: python tools/bench_reference_rocm.py [out.txt [pwg]]   (measurement infrastructure, not product code)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
T0 = time.time()
OUT = open(sys.argv[1], "w") if len(sys.argv) > 1 else None


def say(msg):
    line = f"[{time.time() - T0:7.1f} s] {msg}"
    print(line, flush=True)
    if OUT:
        OUT.write(line + "\n")
        OUT.flush()
        os.fsync(OUT.fileno())


def main():
    import torch

    import bench
    from oracle import ref_run

    say(f"torch {torch.__version__}, device {torch.cuda.get_device_name(0)}, reference package available: {ref_run.available()}")
    dev = torch.device("cuda:0")
    conf = bench.load_conf(bench.TRAIN_CONFIGS["c3"])
    g = ref_run.generator("HiFiGANGenerator", conf["generator_params"]).to(dev)
    say("reference HiFiGANGenerator on the device (weight norm removed, eval)")
    gen = torch.Generator(device="cpu").manual_seed(0)
    with torch.no_grad():
        for b, frames, reps in ((1, 100, 3), (16, 800, 5)):
            c = torch.randn(b, 80, frames, generator=gen).to(dev)
            t = time.time()
            y = g(c)
            torch.cuda.synchronize()
            say(f"B{b} x {frames} frames: first call {time.time() - t:.2f} s (kernel selection / compilation included)")
            t = time.time()
            y = g(c)
            torch.cuda.synchronize()
            say(f"B{b} x {frames} frames: second call {(time.time() - t) * 1e3:.2f} ms")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = float("inf")
            for _ in range(reps):
                e0.record()
                y = g(c)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            say(f"B{b} x {frames} frames: best of {reps}: {best:.2f} ms per forward = {y.numel() / best / 1e3:.2f} M samples/s "
                f"(stock PyTorch-ROCm, eager, fp32)")


def main_pwg():
    """Parallel WaveGAN.v1 generator (C1 / C2's generator) at bench.py's PWG inference batch: B16 x 400 frames."""
    import torch

    import bench
    from oracle import ref_run

    dev = torch.device("cuda:0")
    conf = bench.load_conf(bench.TRAIN_CONFIGS["c2"])
    gp = dict(conf["generator_params"])
    g = ref_run.generator("ParallelWaveGANGenerator", gp).to(dev)
    acw, hop = gp["aux_context_window"], conf["hop_size"]
    say("reference ParallelWaveGANGenerator on the device (weight norm removed, eval)")
    gen = torch.Generator(device="cpu").manual_seed(5)
    with torch.no_grad():
        for b, frames, reps in ((1, 100, 3), (16, 400, 5)):
            c = torch.randn(b, 80, frames + 2 * acw, generator=gen).to(dev)
            z = torch.randn(b, 1, frames * hop, generator=gen).to(dev)
            t = time.time()
            y = g(z, c)
            torch.cuda.synchronize()
            say(f"PWG B{b} x {frames} frames: first call {time.time() - t:.2f} s")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = float("inf")
            for _ in range(reps + 1):
                e0.record()
                y = g(z, c)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            say(f"PWG B{b} x {frames} frames: best of {reps + 1}: {best:.2f} ms per forward = {y.numel() / best / 1e3:.2f} M samples/s "
                f"(stock PyTorch-ROCm, eager, fp32)")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "pwg":
        main_pwg()
    else:
        main()
