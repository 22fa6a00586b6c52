"""Micro-benchmark of the one-launch WaveNet layer (csrc/wavenet.hip) at the PWG.v1 shapes: inference batch
(B16 x 102400) and training batch (B6 x 25600), a few dilations.  PWG_WN_DBG = timing experiments (see wavenet.hip).
usage: bench_wavenet.py"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parallelwavegan_amd import ops


def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


dev = torch.device("cuda:0")
for B, T, save in ((16, 102400, False), (6, 25600, False), (6, 25600, True)):
    x, c, s = torch.randn(B, 64, T, device=dev), torch.randn(B, 80, T, device=dev), torch.randn(B, 64, T, device=dev)
    w = [torch.randn(128, 64, 3, device=dev) * .07, torch.randn(128, 80, 1, device=dev) * .1, torch.randn(64, 64, 1, device=dev) * .1,
         torch.randn(64, 64, 1, device=dev) * .1]
    b = [torch.randn(128, device=dev), torch.randn(64, device=dev), torch.randn(64, device=dev)]
    for dil in (1, 16, 512):
        desc = ops.make_wavenet_desc(B, T, dil, out_mul=math.sqrt(.5))
        img = ops.wavenet_pack_weights(desc, w[0], None, w[1], None, w[2], None, w[3], None)
        so = torch.empty_like(s)
        t = timeit(lambda: ops.wavenet_layer_forward(desc, x, c, s, img, b[0], b[1], b[2], save=save, skips_out=so))
        fl = 2.0 * B * T * (128 * 272 + 128 * 64)
        by = 4.0 * B * T * (64 * 4 + 80 + (192 if save else 0))
        print(f"B{B} T{T} d{dil:3d} save{int(save)}: {t*1e3:7.1f} us {fl/t/1e9:6.1f} TF {by/t/1e6:6.0f} GB/s", flush=True)

# backward, weight path (two contraction launches + two reduce launches; PWG_PROFILE=1 prints each)
B, T = 6, 25600
dz = torch.randn(B, 128, T, device=dev)
x, c, g = torch.randn(B, 64, T, device=dev), torch.randn(B, 80, T, device=dev), torch.randn(B, 64, T, device=dev)
gs, go = torch.randn(B, 64, T, device=dev), torch.randn(B, 64, T, device=dev)
for dil in (1, 16, 512):
    desc = ops.make_wavenet_desc(B, T, dil, out_mul=math.sqrt(.5))
    t = timeit(lambda: ops.wavenet_weight_backward(desc, dz, x, c, gs, go, g))
    fl = 2.0 * B * T * 128 * (272 + 64)
    print(f"weight backward B{B} T{T} d{dil:3d}: {t*1e3:7.1f} us {fl/t/1e9:6.1f} TF", flush=True)
