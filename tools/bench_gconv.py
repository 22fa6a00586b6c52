"""Micro-benchmark of the grouped k = 41 layers of the scale discriminators at the BASELINE training shapes
(C4: multi-band MelGAN B 64 x 16384; C3: HiFi-GAN MSD layer 2, B 16 x 8192): forward, data gradient, weight
gradient; TFLOP/s and GB/s (algorithmic).  PWG_NO_GCONV=1 times the general 32 x 32 x 2 kernels instead.
usage: bench_gconv.py"""
import os, sys
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


ROWS = [("C4 L1 16->64 g4", 64, 16, 64, 4, 16384, 4), ("C4 L1 scale1", 64, 16, 64, 4, 8192, 4), ("C4 L2 64->256 g16", 64, 64, 256, 16, 4096, 4),
        ("C4 L3 256->512 g64", 64, 256, 512, 64, 1024, 4), ("C4 L3 scale2", 64, 256, 512, 64, 256, 4),
        ("C3 MSD L2 128->256 g16 s4", 16, 128, 256, 16, 2048, 4), ("C3 MSD L2 scale2", 16, 128, 256, 16, 513, 4),
        ("(8->16 per group, s2)", 16, 128, 256, 16, 4096, 2)]
dev = torch.device("cuda:0")
for name, B, cin, cout, g, T, s in ROWS:
    t_out = (T + 40 - 41) // s + 1
    desc = ops.make_conv_desc(B, cin, cout, T, t_out, 41, s, 1, 20, g, post_act="leaky_relu", post_slope=0.2)
    x = torch.randn(B, cin, T, device=dev); w = torch.randn(cout, cin // g, 41, device=dev) * 0.1
    b = torch.randn(cout, device=dev); dy = torch.randn(B, cout, t_out, device=dev)
    wp, wpb = ops.pack_weight(desc, w), ops.pack_weight_bwd(desc, w)
    y = torch.empty(B, cout, t_out, device=dev); dx = torch.empty_like(x)
    fl = 2.0 * B * cout * t_out * (cin // g) * 41
    by = 4.0 * (x.numel() + y.numel())
    tf = timeit(lambda: ops.conv1d_forward(desc, x, wp, b, out=y))
    td = timeit(lambda: ops.conv1d_backward_data(desc, dy, wpb, None, out=dx))
    tw = timeit(lambda: ops.conv1d_backward_weight(desc, x, dy, tuple(w.shape)))
    print(f"{name:28s} fwd {tf*1e3:7.1f} us {fl/tf/1e9:6.1f} TF {by/tf/1e6:6.0f} GB/s | dgrad {td*1e3:7.1f} us {fl/td/1e9:6.1f} TF | "
          f"wgrad {tw*1e3:7.1f} us {fl/tw/1e9:6.1f} TF", flush=True)
