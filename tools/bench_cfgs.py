"""Compare explicit tile configurations on a few HiFi-GAN inference layers. usage: bench_cfgs.py cfg[,cfg...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parallelwavegan_amd import ops
cfgs = [int(c) for c in sys.argv[1].split(",")]
dev = torch.device("cuda:0")
def timeit(fn, reps=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
B = 16
for C, T in ((256, 6400), (128, 51200)):
    for k, d in ((3, 1), (7, 1), (11, 1), (11, 5)):
        desc = ops.make_conv_desc(B, C, C, T, T, k, dilation=d, pad_left=(k - 1) // 2 * d, pre_act="leaky_relu", pre_slope=0.1)
        w = torch.randn(C, C, k, device=dev) * 0.05; wp = ops.pack_weight(desc, w)
        x = torch.randn(B, C, T, device=dev); bias = torch.randn(C, device=dev); add1 = torch.randn(B, C, T, device=dev)
        y = torch.empty(B, C, T, device=dev)
        fl = 2.0 * C * C * k * T * B
        ref = ops.conv1d_forward_cfg(desc, x, wp, bias, add1, tile_config=2, use_dma=True).clone()
        line = f"C{C} k{k} d{d}:"
        for c in cfgs:
            out = ops.conv1d_forward_cfg(desc, x, wp, bias, add1, out=y, tile_config=c, use_dma=True)
            err = (out - ref).abs().max().item()
            t = timeit(lambda: ops.conv1d_forward_cfg(desc, x, wp, bias, add1, out=y, tile_config=c, use_dma=True))
            line += f"  c{c} {fl/t/1e9:6.1f}TF (e{err:.0e})"
        print(line, flush=True)
