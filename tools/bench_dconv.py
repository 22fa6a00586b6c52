"""Tile-configuration sweep of the forward conv kernel on the discriminator / small-T training problems
(HiFi-GAN V1 MPD + MSD at B=16 x 8192, plus the deep generator layers).  GPU only.
usage: bench_dconv.py [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from parallelwavegan_amd import ops

# (label, c_in, c_out, t_in, t_out, width, kernel, stride, dilation, groups, transposed)
SHAPES = [
    ("MPD 1024 p2", 1024, 1024, 51, 51, 2, 5, 1, 1, 1, 0),
    ("MPD 1024 p11", 1024, 1024, 10, 10, 11, 5, 1, 1, 1, 0),
    ("MPD 512>1024 p2 s3", 512, 1024, 152, 51, 2, 5, 3, 1, 1, 0),
    ("MPD 512>1024 p11 s3", 512, 1024, 28, 10, 11, 5, 3, 1, 1, 0),
    ("MPD 128>512 p2 s3", 128, 512, 456, 152, 2, 5, 3, 1, 1, 0),
    ("MSD 1024 k5 T32", 1024, 1024, 32, 32, 1, 5, 1, 1, 1, 0),
    ("MSD 1024 k5 T9", 1024, 1024, 9, 9, 1, 5, 1, 1, 1, 0),
    ("MSD 1024 g16 k41 T32", 1024, 1024, 32, 32, 1, 41, 1, 1, 16, 0),
    ("MSD 512>1024 g16 k41 s4", 512, 1024, 128, 32, 1, 41, 4, 1, 16, 0),
    ("MSD 128 g4 k41 s4", 128, 128, 8192, 2048, 1, 41, 4, 1, 4, 0),
    ("MSD 1024>1 k3 T32", 1024, 1, 32, 32, 1, 3, 1, 1, 1, 0),
    ("G res 256 k11 T256", 256, 256, 256, 256, 1, 11, 1, 1, 1, 0),
    ("G res 256 k3 T256", 256, 256, 256, 256, 1, 3, 1, 1, 1, 0),
    ("G res 128 k11 T2048", 128, 128, 2048, 2048, 1, 11, 1, 1, 1, 0),
    ("G convT 512>256 k16 s8", 512, 256, 32, 256, 1, 16, 8, 1, 1, 1),
]
CFGS = [0, 2, 9, 10, 12, 13, 14, 15, 16]


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


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dev = torch.device("cuda:0")
    print(f"{'problem':26s} {'default us':>10s} {'TF':>6s} | " + " ".join(f"{'c%d' % c:>7s}" for c in CFGS))
    for name, ci, co, ti, to, w, k, s, d, g, tr in SHAPES:
        pad = (s // 2 + s % 2) if tr else (k - 1) // 2 * d
        desc = ops.make_conv_desc(B, ci, co, ti, to, k, stride=s, dilation=d, pad_left=pad, groups=g, transposed=bool(tr),
                                  width=w, pre_act="leaky_relu", pre_slope=0.1)
        wt = torch.randn((ci, co // g, k) if tr else (co, ci // g, k), device=dev) * 0.05
        wp = ops.pack_weight(desc, wt)
        x = torch.randn(B, ci, ti * w, device=dev)
        bias = torch.randn(co, device=dev)
        y = torch.empty(B, co, to * w, device=dev)
        flops = 2.0 * B * ci * (co // g) * k * (ti if tr else to) * w
        ms = timeit(lambda: ops.conv1d_forward(desc, x, wp, bias, None, out=y))
        cells = []
        for c in CFGS:
            try:
                t = timeit(lambda: ops.conv1d_forward_cfg(desc, x, wp, bias, None, tile_config=c, use_dma=True, out=y), reps=5)
                cells.append(f"{t*1e3:7.1f}")
            except Exception:
                cells.append(f"{'-':>7s}")
        print(f"{name:26s} {ms*1e3:10.1f} {flops/ms/1e9:6.1f} | " + " ".join(cells))


if __name__ == "__main__":
    main()
