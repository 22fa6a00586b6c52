"""Tile-configuration sweep over stride-1 dense convolution shapes given on the command line: what the planner picks
against the best of all tile configurations.  GPU box only.
usage: bench_shapes_cfg.py B:Cin:Cout:T:k:d [...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from parallelwavegan_amd import ops
from tools.bench_conv import CFG, timeit

dev = torch.device("cuda:0")
for spec in sys.argv[1:]:
    B, ci, co, T, k, d = (int(v) for v in spec.split(":"))
    desc = ops.make_conv_desc(B, ci, co, T, T, k, dilation=d, pad_left=(k - 1) // 2 * d, pre_act="leaky_relu", pre_slope=0.2)
    w = torch.randn(co, ci, k, device=dev) * 0.03
    wp = ops.pack_weight(desc, w)
    x = torch.randn(B, ci, T, device=dev)
    bias = torch.randn(co, device=dev)
    y = torch.empty(B, co, T, device=dev)
    flops = 2.0 * ci * co * k * T * B
    ms = timeit(lambda: ops.conv1d_forward(desc, x, wp, bias, out=y))
    line = f"{spec:24s} planner {ms * 1e3:7.1f} us {flops / ms / 1e9:6.1f} TF"
    ref = ops.conv1d_forward(desc, x, wp, bias).clone()
    res = []
    for cid in CFG:
        try:
            out = ops.conv1d_forward_cfg(desc, x, wp, bias, out=y, tile_config=cid, use_dma=True)
            torch.cuda.synchronize()
            err = (out - ref).abs().max().item()
            t = timeit(lambda: ops.conv1d_forward_cfg(desc, x, wp, bias, out=y, tile_config=cid, use_dma=True), reps=5)
            res.append((t, cid, err))
        except RuntimeError:
            pass
    res.sort()
    line += " | best: " + "  ".join(f"c{c}({CFG[c][0]}x{CFG[c][1]}x{CFG[c][2]}) {flops / t / 1e9:.0f}TF e{er:.0e}" for t, c, er in res[:4])
    print(line, flush=True)
