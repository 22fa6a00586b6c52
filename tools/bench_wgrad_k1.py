"""Weight gradient of the 1 x 1 convolutions of MB-MelGAN's residual stacks at the C4 batch (B = 64): per-launch HIP-event
times of the producing kernel and of the finishers, HBM rate on the algorithmic bytes (one read of G and X).  GPU only.
    PWG_WG_K1=0 python tools/bench_wgrad_k1.py    # the general MFMA weight-gradient kernel
    python tools/bench_wgrad_k1.py                # wgrad_k1_kernel (round 6)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from parallelwavegan_amd import ops

SHAPES = [(64, 96, 96, 2048), (64, 48, 48, 4096), (64, 192, 192, 512), (64, 64, 64, 4096), (16, 32, 32, 8192), (16, 128, 128, 2048)]


def main():
    dev = torch.device("cuda:0")
    print(f"PWG_WG_K1={os.environ.get('PWG_WG_K1', '1')}")
    print(f"{'B x C x T':>18s} {'kernel':>28s} {'us':>8s} {'TB/s':>6s} {'TF':>6s} | {'finish us':>9s} {'total us':>9s}")
    for B, ci, co, t in SHAPES:
        desc = ops.make_conv_desc(B, ci, co, t, t, 1, pre_act="leaky_relu", pre_slope=0.2)
        x = torch.randn(B, ci, t, device=dev)
        dy = torch.randn(B, co, t, device=dev)
        v = torch.randn(co, ci, 1, device=dev)
        g = torch.ones(co, device=dev)
        for _ in range(3):
            ops.conv1d_backward_weight_wn(desc, x, dy, v, g)
        reps = 20
        with ops.profile() as p:
            for _ in range(reps):
                ops.conv1d_backward_weight_wn(desc, x, dy, v, g)
        torch.cuda.synchronize()
        main_k = [k for k in p.results if k.startswith("wgrad_k1_kernel") or k.startswith("conv1d_wgrad_kernel")]
        ms = sum(p.results[k]["ms"] for k in main_k) / reps
        tot = sum(r["ms"] for r in p.results.values()) / reps
        byts = 4.0 * B * t * (ci + co)
        fl = 2.0 * B * t * ci * co
        print(f"{B:4d} x {ci:3d}>{co:3d} x {t:5d} {main_k[0].split(' ')[0]:>28s} {ms * 1e3:8.1f} {byts / ms / 1e9:6.2f} {fl / ms / 1e9:6.1f} | "
              f"{(tot - ms) * 1e3:9.1f} {tot * 1e3:9.1f}")


if __name__ == "__main__":
    main()
