"""Micro-benchmark of pwg_conv1d_backward_weight on the HiFi-GAN V1 training problem set (C3,
B=16 x 8192 samples): per-kernel HIP-event times (wgrad MFMA kernel / slab reduce / bias).  GPU only.
usage: bench_wgrad.py [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from parallelwavegan_amd import ops

# (label, c_in, c_out, t_in, t_out, width, kernel, stride, dilation, groups, transposed, pre_act)
SHAPES = [
    ("G res 128 k11", 128, 128, 2048, 2048, 1, 11, 1, 1, 1, 0, 1),
    ("G res 128 k7", 128, 128, 2048, 2048, 1, 7, 1, 1, 1, 0, 1),
    ("G res 128 k3 d5", 128, 128, 2048, 2048, 1, 3, 1, 5, 1, 0, 1),
    ("G res 64 k11", 64, 64, 4096, 4096, 1, 11, 1, 1, 1, 0, 1),
    ("G res 64 k3", 64, 64, 4096, 4096, 1, 3, 1, 1, 1, 0, 1),
    ("G res 32 k11", 32, 32, 8192, 8192, 1, 11, 1, 1, 1, 0, 1),
    ("G res 32 k3", 32, 32, 8192, 8192, 1, 3, 1, 1, 1, 0, 1),
    ("G res 256 k11", 256, 256, 256, 256, 1, 11, 1, 1, 1, 0, 1),
    ("G convT 512>256 k16 s8", 512, 256, 32, 256, 1, 16, 8, 1, 1, 1, 1),
    ("G convT 256>128 k16 s8", 256, 128, 256, 2048, 1, 16, 8, 1, 1, 1, 1),
    ("MPD 1024 p2", 1024, 1024, 51, 51, 2, 5, 1, 1, 1, 0, 0),
    ("MPD 1024 p11", 1024, 1024, 10, 10, 11, 5, 1, 1, 1, 0, 0),
    ("MPD 512>1024 p3 s3", 512, 1024, 102, 34, 3, 5, 3, 1, 1, 0, 0),
    ("MPD 128>512 p2 s3", 128, 512, 456, 152, 2, 5, 3, 1, 1, 0, 0),
    ("MSD 128 g4 k41 s4", 128, 128, 8192, 2048, 1, 41, 4, 1, 4, 0, 0),
    ("MSD 1024 g16 k41 s4", 1024, 1024, 128, 32, 1, 41, 4, 1, 16, 0, 0),
    ("MSD 1024 k5", 1024, 1024, 32, 32, 1, 5, 1, 1, 1, 0, 0),
    ("G out 32>1 k7", 32, 1, 8192, 8192, 1, 7, 1, 1, 1, 0, 1),
    ("G in 80>512 k7", 80, 512, 32, 32, 1, 7, 1, 1, 1, 0, 0),
]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dev = torch.device("cuda:0")
    print(f"{'problem':26s} {'wgrad us':>9s} {'TF':>6s} {'reduce us':>9s} {'bias us':>8s} {'total us':>9s} {'TF':>6s}")
    for name, ci, co, ti, to, w, k, s, d, g, tr, act in SHAPES:
        if tr:
            pad = s // 2 + s % 2
        else:
            pad = (k - 1) // 2 * d
        desc = ops.make_conv_desc(B, ci, co, ti, to, k, stride=s, dilation=d, pad_left=pad, groups=g, transposed=bool(tr),
                                  width=w, pre_act="leaky_relu" if act else None, pre_slope=0.1)
        x = torch.randn(B, ci, ti * w, device=dev)
        dy = torch.randn(B, co, to * w, device=dev)
        wshape = (ci, co // g, k) if tr else (co, ci // g, k)
        for _ in range(3):
            ops.conv1d_backward_weight(desc, x, dy, wshape)
        reps = 10
        with ops.profile() as p:
            for _ in range(reps):
                ops.conv1d_backward_weight(desc, x, dy, wshape)
        torch.cuda.synchronize()
        r = p.results
        flops = 2.0 * B * ci * (co // g) * k * (ti if tr else to) * w
        wg = r.get("conv1d_wgrad_kernel", dict(ms=0))["ms"] / reps
        rd = r.get("reduce_slabs_kernel", dict(ms=0))["ms"] / reps
        bg = r.get("bias_grad_kernel", dict(ms=0))["ms"] / reps
        tot = wg + rd + bg
        print(f"{name:26s} {wg*1e3:9.1f} {flops/wg/1e9:6.1f} {rd*1e3:9.1f} {bg*1e3:8.1f} {tot*1e3:9.1f} {flops/tot/1e9:6.1f}")


if __name__ == "__main__":
    main()
