"""Tile-configuration sweep over the HiFi-GAN discriminator's convolution shapes at the C3 training batch (B = 16 x 8192):
what the planner picks (alone on the chip / with the concurrency hint of a branch-parallel captured step) against the
best of all tile configurations.  GPU box only.  usage: bench_dshapes.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from parallelwavegan_amd import _lib, ops
from tools.bench_conv import CFG, timeit

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
T = 8192
shapes = []
for p in (2, 3, 5, 7, 11):
    rows = -(-T // p)
    r1 = (rows + 4 - 5) // 3 + 1
    r2 = (r1 + 4 - 5) // 3 + 1
    r3 = (r2 + 4 - 5) // 3 + 1
    r4 = (r3 + 4 - 5) // 3 + 1
    shapes.append((f"mpd p{p} 128->512 k5 s3", dict(c_in=128, c_out=512, t_in=r2, t_out=r3, k=5, stride=3, pad=2, width=p)))
    shapes.append((f"mpd p{p} 512->1024 k5 s3", dict(c_in=512, c_out=1024, t_in=r3, t_out=r4, k=5, stride=3, pad=2, width=p)))
    shapes.append((f"mpd p{p} 1024->1024 k5 s1", dict(c_in=1024, c_out=1024, t_in=r4, t_out=r4, k=5, stride=1, pad=2, width=p)))
    shapes.append((f"mpd p{p} dgrad of 512->1024 s3", dict(c_in=1024, c_out=512, t_in=r4, t_out=r3, k=5, stride=3, pad=2, width=p, transposed=True)))
for t in (32, 17, 9):
    shapes.append((f"msd T{t} 1024->1024 k41 g16", dict(c_in=1024, c_out=1024, t_in=t, t_out=t, k=41, stride=1, pad=20, groups=16)))
    shapes.append((f"msd T{t} 1024->1024 k5", dict(c_in=1024, c_out=1024, t_in=t, t_out=t, k=5, stride=1, pad=2)))
    shapes.append((f"msd T{t} 512->1024 k41 s4 g16", dict(c_in=512, c_out=1024, t_in=4 * t - 3 if t != 32 else 128, t_out=t, k=41, stride=4, pad=20, groups=16)))
lib = _lib.lib()
for name, p in shapes:
    w_ = p.get("width", 1)
    g = p.get("groups", 1)
    tr = p.get("transposed", False)
    desc = ops.make_conv_desc(B, p["c_in"], p["c_out"], p["t_in"], p["t_out"], p["k"], stride=p["stride"], pad_left=p["pad"],
                              groups=g, transposed=tr, width=w_, pre_act="leaky_relu", pre_slope=0.1)
    w = (torch.randn(p["c_in"], p["c_out"] // g, p["k"], device=dev) if tr else torch.randn(p["c_out"], p["c_in"] // g, p["k"], device=dev)) * 0.03
    wp = ops.pack_weight(desc, w)
    x = torch.randn(B, p["c_in"], p["t_in"] * w_, device=dev)
    bias = torch.randn(p["c_out"], device=dev)
    y = torch.empty(B, p["c_out"], p["t_out"] * w_, device=dev)
    flops = 2.0 * p["c_in"] * p["c_out"] // g * p["k"] * (p["t_in"] if tr else p["t_out"]) * w_ * B
    line = f"{name:34s} cols/item {(p['t_in'] if tr else p['t_out']) * w_:5d}"
    for hint in (1.0, 0.5):
        lib.pwg_set_concurrency_hint(hint)
        ms = timeit(lambda: ops.conv1d_forward(desc, x, wp, bias, out=y))
        line += f" | hint {hint}: {ms * 1e3:7.1f} us {flops / ms / 1e9:6.1f} TF"
    lib.pwg_set_concurrency_hint(1.0)
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
    line += " | best unsplit: " + "  ".join(f"c{c}({CFG[c][0]}x{CFG[c][1]}x{CFG[c][2]}) {flops / t / 1e9:.0f}TF e{er:.0e}" for t, c, er in res[:3])
    print(line, flush=True)
