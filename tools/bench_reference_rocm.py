"""The reference's OWN package (staged copy oracle/_ref) on the MI355X through stock PyTorch-ROCm (ATen + MIOpen):
the second, non-graded comparison SURVEY.md s8(d) asks for.  HiFi-GAN V1 generator inference at the headline batch
(B16 x 800 frames, weight norm removed, eval, fp32, no grad), timed with HIP events after warm-up.  Progress lines are
flushed as they happen so that a run cut by its time limit still reports how far it got (MIOpen has no precompiled
kernel database for gfx950 in this image: the first call of every distinct convolution compiles its kernel).

usage: python tools/bench_reference_rocm.py [out.txt [pwg | train [c3 c2 c4 c5 ...]]]   (measurement infrastructure, not product code)

``train``: the reference's OWN ``Trainer._train_step`` (bin/train.py:189-340) on the device at the recipe's batch, both
phases active, synthetic batch as bench.py's; the MIOpen kernel search of the first steps is excluded (warm-up steps until
two consecutive steps agree within 10 %, at most 6), then >= 20 steps timed with HIP events + synchronisation.
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


def main_train(tags):
    import torch

    import bench
    from oracle import ref_run

    say(f"torch {torch.__version__}, device {torch.cuda.get_device_name(0)}, reference package available: {ref_run.available()}")
    dev = torch.device("cuda:0")
    for tag in tags:
        conf = bench.load_conf(bench.TRAIN_CONFIGS[tag])
        b = conf["batch_size"]
        batch = bench.synthetic_batch(conf, b, dev, 0)
        torch.manual_seed(4321)
        tr = ref_run.trainer(conf, batch, device=dev)
        say(f"{tag}: reference Trainer on the device ({bench.TRAIN_CONFIGS[tag]}, B={b} x {conf['batch_max_steps']})")
        prev = None
        for i in range(6):  # MIOpen searches / compiles a kernel per distinct (fwd, bwd-data, bwd-weight) problem
            t = time.time()
            tr._train_step(batch)
            torch.cuda.synchronize()
            dt = time.time() - t
            say(f"{tag}: warm-up step {i}: {dt * 1e3:.1f} ms")
            if prev is not None and abs(dt - prev) <= 0.1 * prev and i >= 2:
                break
            prev = dt
        n = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t = time.time()
        e0.record()
        for _ in range(n):
            tr._train_step(batch)
        e1.record()
        torch.cuda.synchronize()
        wall = (time.time() - t) / n * 1e3
        ms = e0.elapsed_time(e1) / n
        losses = {k.split("/")[-1]: round(float(v), 5) for k, v in tr.total_train_loss.items()}
        say(f"{tag}: {n} timed steps: {ms:.2f} ms per step (HIP events; host wall {wall:.2f} ms) = {1e3 / ms:.2f} steps/s "
            f"(the reference's Trainer._train_step, stock PyTorch-ROCm / MIOpen, eager, fp32); accumulated losses {losses}")
        # where its time goes: torch's own profiler over 2 further steps, top kernels by device time
        try:
            from torch.profiler import ProfilerActivity, profile

            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                for _ in range(2):
                    tr._train_step(batch)
                torch.cuda.synchronize()
            rows = sorted(prof.key_averages(), key=lambda r: -r.device_time_total)[:14]
            tot = sum(r.device_time_total for r in prof.key_averages())
            say(f"{tag}: device time {tot / 2e3:.2f} ms per step over {sum(r.count for r in prof.key_averages()) // 2} kernels; top:")
            for r in rows:
                say(f"{tag}:   {r.device_time_total / 2e3:8.3f} ms  x{r.count // 2:<5d} {r.key[:110]}")
        except Exception as e:  # noqa: BLE001
            say(f"{tag}: profiler unavailable ({type(e).__name__}: {e})")
        del tr
        torch.cuda.empty_cache()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "train":
        main_train(sys.argv[3:] or ["c3"])
    elif len(sys.argv) > 2 and sys.argv[2] == "pwg":
        main_pwg()
    else:
        main()
