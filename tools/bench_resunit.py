#!/usr/bin/env python3
"""Time the one-launch MRF residual unit (csrc/resunit.hip) against the two convolution launches it replaces,
on the HiFi-GAN V1 inference shapes of bench.py (16 utterances x 800 frames: T = 102400 at C = 64, 204800 at
C = 32).  Usage: python tools/bench_resunit.py [--batch 16] [--frames 800]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parallelwavegan_amd import ops  # noqa: E402
from parallelwavegan_amd.layers import HiFiGANResidualBlock  # noqa: E402


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--frames", type=int, default=800)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    total_f = total_s = 0.0
    for channels, t in [(64, args.frames * 128), (32, args.frames * 256)]:
        x = torch.randn(args.batch, channels, t, device=dev)
        for kernel in (3, 7, 11):
            blk = HiFiGANResidualBlock(kernel, channels, (1, 3, 5)).to(dev)
            with torch.no_grad():
                flops = 3 * 2 * 2.0 * channels * channels * kernel * args.batch * t
                blk.fuse_units = True
                tf = timeit(lambda: blk(x))
                blk.fuse_units = False
                ts = timeit(lambda: blk(x))
                blk.fuse_units = True
                err = (blk(x) - (lambda: (setattr(blk, "fuse_units", False), blk(x))[1])()).abs().max().item()
                blk.fuse_units = True
            total_f += tf
            total_s += ts
            print(f"C={channels} k={kernel} T={t}: block (3 units) one-launch {tf:8.1f} us {flops / tf * 1e-6:6.1f} TF | "
                  f"separate {ts:8.1f} us {flops / ts * 1e-6:6.1f} TF | x{ts / tf:.2f} | max diff {err:.2e}", flush=True)
    print(f"sum over the 6 blocks: one-launch {total_f / 1e3:.2f} ms, separate {total_s / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
