"""Forced (tile configuration, split-K) sweep for the period discriminators' 1024-channel layers (PWG_FORCE_CFG):
conv + split-K finish time per combination.  GPU box only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from parallelwavegan_amd import ops

CFG = {0: (128, 128, 8), 1: (128, 128, 16), 2: (128, 128, 4), 9: (128, 64, 8), 12: (64, 128, 8), 13: (64, 64, 8), 15: (64, 64, 16),
       16: (32, 128, 16), 11: (64, 256, 4), 3: (64, 256, 8), 14: (128, 32, 8), 4: (64, 256, 16), 5: (32, 256, 8), 6: (32, 256, 16),
       7: (32, 512, 8), 8: (32, 512, 16), 10: (32, 128, 8), 17: (64, 64, 4)}


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


dev = torch.device("cuda:0")
B = 16
K1 = len(sys.argv) > 1 and sys.argv[1] == "k1"
FOLD = len(sys.argv) > 1 and sys.argv[1] == "fold"
# round 6: the batch-FOLDED tail layers of the scale discriminators (layers/conv.py: _fold_batch): one item of width 16
fold_shapes = []
for t in (32, 17, 9):
    fold_shapes += [(f"fold T{t} 1024->1024 k41 g16", dict(c_in=1024, c_out=1024, t_in=t, t_out=t, k=41, stride=1, pad=20, groups=16, width=16, batch=1)),
                    (f"fold T{t} 1024->1024 k41 g16 dgrad", dict(c_in=1024, c_out=1024, t_in=t, t_out=t, k=41, stride=1, pad=20, groups=16, width=16, batch=1, transposed=True)),
                    (f"fold T{t} 1024->1024 k5", dict(c_in=1024, c_out=1024, t_in=t, t_out=t, k=5, stride=1, pad=2, width=16, batch=1)),
                    (f"fold T{t} 512->1024 k41 s4 g16", dict(c_in=512, c_out=1024, t_in=4 * t - 3 if t != 32 else 128, t_out=t, k=41, stride=4, pad=20, groups=16, width=16, batch=1)),
                    (f"fold T{t} 512->1024 k41 s4 g16 dgrad", dict(c_in=1024, c_out=512, t_in=t, t_out=4 * t - 3 if t != 32 else 128, k=41, stride=4, pad=20, groups=16, width=16, batch=1, transposed=True))]
# round 6: the C3 categories furthest below 100 TFLOP/s that are not folded (profiles/r06_a_train_shapes_c3.txt)
C3X = len(sys.argv) > 1 and sys.argv[1] == "c3x"
c3x_shapes = [
    ("msd L2 128->128 k41 s4 g4 T8192", dict(c_in=128, c_out=128, t_in=8192, t_out=2048, k=41, stride=4, pad=20, groups=4)),
    ("msd L2 128->128 k41 s4 g4 T4097", dict(c_in=128, c_out=128, t_in=4097, t_out=1025, k=41, stride=4, pad=20, groups=4)),
    ("msd L2 dgrad (convT 128->128 k41 s4 g4) T2048", dict(c_in=128, c_out=128, t_in=2048, t_out=8192, k=41, stride=4, pad=20, groups=4, transposed=True)),
    ("mpd p5 L4 dgrad (convT 1024->512 k5 s3)", dict(c_in=1024, c_out=512, t_in=21, t_out=61, k=5, stride=3, pad=2, width=5, transposed=True)),
    ("mpd p5 L3 dgrad (convT 512->128 k5 s3)", dict(c_in=512, c_out=128, t_in=61, t_out=183, k=5, stride=3, pad=2, width=5, transposed=True)),
    ("mpd p5 L5 1024->1024 k5 s1", dict(c_in=1024, c_out=1024, t_in=21, t_out=21, k=5, stride=1, pad=2, width=5)),
    ("mpd p11 L5 1024->1024 k5 s1", dict(c_in=1024, c_out=1024, t_in=10, t_out=10, k=5, stride=1, pad=2, width=11)),
    ("G res 256 k3 T256", dict(c_in=256, c_out=256, t_in=256, t_out=256, k=3, stride=1, pad=1)),
    ("G res 256 k7 T256", dict(c_in=256, c_out=256, t_in=256, t_out=256, k=7, stride=1, pad=3)),
    ("G res 128 k3 T2048", dict(c_in=128, c_out=128, t_in=2048, t_out=2048, k=3, stride=1, pad=1)),
    ("G convT 512->256 k16 s8 T32", dict(c_in=512, c_out=256, t_in=32, t_out=256, k=16, stride=8, pad=4, transposed=True)),
    ("G convT 256->128 k16 s8 T256", dict(c_in=256, c_out=128, t_in=256, t_out=2048, k=16, stride=8, pad=4, transposed=True)),
]
shapes = fold_shapes if FOLD else c3x_shapes if C3X else [("c4 k1 96->96 T2048 B64", dict(c_in=96, c_out=96, t_in=2048, t_out=2048, k=1, stride=1, pad=0, batch=64)),
          ("c4 k1 48->48 T4096 B64", dict(c_in=48, c_out=48, t_in=4096, t_out=4096, k=1, stride=1, pad=0, batch=64)),
          ("c4 k1 192->192 T512 B64", dict(c_in=192, c_out=192, t_in=512, t_out=512, k=1, stride=1, pad=0, batch=64)),
          ("c4 k3 96->96 d3 T2048 B64", dict(c_in=96, c_out=96, t_in=2048, t_out=2048, k=3, stride=1, pad=3, batch=64, dil=3)),
          ("c2 pwg-D k3 64->64 T25600 B6", dict(c_in=64, c_out=64, t_in=25600, t_out=25600, k=3, stride=1, pad=1, batch=6))] if K1 else [("L5 p5 1024->1024 k5 s1 W5 (21 rows)", dict(c_in=1024, c_out=1024, t_in=21, t_out=21, k=5, stride=1, pad=2, width=5)),
          ("L5 p2 1024->1024 k5 s1 W2 (51 rows)", dict(c_in=1024, c_out=1024, t_in=51, t_out=51, k=5, stride=1, pad=2, width=2)),
          ("L4 p5 512->1024 k5 s3 W5", dict(c_in=512, c_out=1024, t_in=61, t_out=21, k=5, stride=3, pad=2, width=5)),
          ("L4 p5 dgrad (convT 1024->512)", dict(c_in=1024, c_out=512, t_in=21, t_out=61, k=5, stride=3, pad=2, width=5, transposed=True)),
          ("msd T17 1024->1024 k5", dict(c_in=1024, c_out=1024, t_in=17, t_out=17, k=5, stride=1, pad=2)),
          ("L3 p5 128->512 k5 s3 W5", dict(c_in=128, c_out=512, t_in=183, t_out=61, k=5, stride=3, pad=2, width=5))]
for name, p in shapes:
    w_ = p.get("width", 1)
    tr = p.get("transposed", False)
    B = p.get("batch", 16)
    g_ = p.get("groups", 1)
    desc = ops.make_conv_desc(B, p["c_in"], p["c_out"], p["t_in"], p["t_out"], p["k"], stride=p["stride"], pad_left=p["pad"], dilation=p.get("dil", 1),
                              groups=g_, transposed=tr, width=w_, pre_act=None if FOLD else "leaky_relu", pre_slope=0.1)
    w = (torch.randn(p["c_in"], p["c_out"] // g_, p["k"], device=dev) if tr else torch.randn(p["c_out"], p["c_in"] // g_, p["k"], device=dev)) * 0.03
    wp = ops.pack_weight(desc, w)
    x = torch.randn(B, p["c_in"], p["t_in"] * w_, device=dev)
    bias = torch.randn(p["c_out"], device=dev)
    y = torch.empty(B, p["c_out"], p["t_out"] * w_, device=dev)
    flops = 2.0 * p["c_in"] * p["c_out"] // g_ * p["k"] * (p["t_in"] if tr else p["t_out"]) * w_ * B
    os.environ.pop("PWG_FORCE_CFG", None)
    ref = ops.conv1d_forward(desc, x, wp, bias).clone()
    ms = timeit(lambda: ops.conv1d_forward(desc, x, wp, bias, out=y))
    print(f"{name}: planner {ms * 1e3:.1f} us {flops / ms / 1e9:.1f} TF")
    res = []
    for cid, (bm, bn, ck) in CFG.items():
        for ks in ((1,) if K1 else (1, 2, 4, 8, 16) if FOLD else (1, 2, 4, 8)):
            os.environ["PWG_FORCE_CFG"] = f"{cid},{ks}"
            try:
                out = ops.conv1d_forward(desc, x, wp, bias, out=y)
                torch.cuda.synchronize()
                err = (out - ref).abs().max().item()
                t = timeit(lambda: ops.conv1d_forward(desc, x, wp, bias, out=y), reps=6)
                res.append((t, cid, ks, err))
            except RuntimeError as e:
                res.append((9e9, cid, ks, -1))
    res.sort()
    for t, c, ks, er in res[:8]:
        print(f"    c{c} ({CFG[c][0]}x{CFG[c][1]}x{CFG[c][2]}) split {ks}: {t * 1e3:7.1f} us {flops / t / 1e9:6.1f} TF  err {er:.0e}")
os.environ.pop("PWG_FORCE_CFG", None)
