"""Sweep every tile configuration on a few problem shapes taken from the C3 training-step profile
(profiles/r02_train_shapes_c3.txt) and print TFLOP/s per configuration next to the heuristic's choice."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from parallelwavegan_amd import ops  # noqa: E402


CFG = {0: (128, 128, 8), 1: (128, 128, 16), 2: (128, 128, 4), 3: (64, 256, 8), 4: (64, 256, 16), 5: (32, 256, 8),
       6: (32, 256, 16), 7: (32, 512, 8), 8: (32, 512, 16), 9: (128, 64, 8), 10: (32, 128, 8), 11: (64, 256, 4),
       12: (64, 128, 8), 13: (64, 64, 8), 14: (128, 32, 8), 15: (64, 64, 16), 16: (32, 128, 16), 17: (64, 64, 4)}


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


SHAPES = [  # (name, B, Cin, Cout, T, k, stride, dil, groups)
    ("pwg 64->64 k1", 6, 64, 64, 25600, 1, 1, 1, 1),
    ("pwg 80->128 k1", 6, 80, 128, 25600, 1, 1, 1, 1),
    ("pwg 128->80 k1", 6, 128, 80, 25600, 1, 1, 1, 1),
    ("pwg 64->128 k3 d1", 6, 64, 128, 25600, 3, 1, 1, 1),
    ("pwg 64->128 k3 d256", 6, 64, 128, 25600, 3, 1, 256, 1),
    ("mb 96->96 k1", 64, 96, 96, 2048, 1, 1, 1, 1),
    ("mb 192->192 k1", 64, 192, 192, 512, 1, 1, 1, 1),
    ("mb 48->48 k1", 64, 48, 48, 4096, 1, 1, 1, 1),
    ("mpd 1024 k5 d11 T110", 16, 1024, 1024, 110, 5, 1, 11, 1),
    ("g 128 k3 T2048", 16, 128, 128, 2048, 3, 1, 1, 1),
    ("g 64 k3 T4096", 16, 64, 64, 4096, 3, 1, 1, 1),
]
dev = torch.device("cuda:0")
for name, b, cin, cout, t, k, s, d, g in SHAPES:
    pad = (k - 1) // 2 * d
    t_out = ops.conv_out_length(t, k, s, d, pad, pad)
    desc = ops.make_conv_desc(b, cin, cout, t, t_out, k, stride=s, dilation=d, pad_left=pad, groups=g,
                              post_act="leaky_relu", post_slope=0.1)
    w = torch.randn(cout, cin // g, k, device=dev) * 0.05
    wp = ops.pack_weight(desc, w)
    x = torch.randn(b, cin, t, device=dev)
    bias = torch.randn(cout, device=dev)
    y = torch.empty(b, cout, t_out, device=dev)
    flops = 2.0 * (cin // g) * cout * k * t_out * b
    ms = timeit(lambda: ops.conv1d_forward(desc, x, wp, bias, out=y))
    res = []
    for cid, (bm, bn, ck) in CFG.items():
        if bm > 64 and cout // g <= 32:
            continue
        try:
            ops.conv1d_forward_cfg(desc, x, wp, bias, out=y, tile_config=cid, use_dma=True)
            torch.cuda.synchronize()
            tt = timeit(lambda: ops.conv1d_forward_cfg(desc, x, wp, bias, out=y, tile_config=cid, use_dma=True), reps=5)
            res.append((tt, cid))
        except RuntimeError:
            pass
    res.sort()
    print(f"{name:24s} default {ms * 1e3:7.1f} us {flops / ms / 1e9:6.1f} TF | " +
          "  ".join(f"c{c}({CFG[c][0]}x{CFG[c][1]}x{CFG[c][2]}) {flops / tt / 1e9:.0f}" for tt, c in res[:6]), flush=True)
