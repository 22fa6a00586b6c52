"""One-launch MelGAN residual stack (csrc/resstack.hip) against the three launches it replaces, at the MB-MelGAN.v2
training shapes (B64: C192 x 512, C96 x 2048, C48 x 4096).  usage: bench_resstack.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parallelwavegan_amd import ops
from parallelwavegan_amd.layers.residual_stack import ResidualStack


def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


dev = torch.device("cuda:0")
B = 64
for C, T in ((192, 512), (96, 2048), (48, 4096)):
    for d in (1, 27):
        x = torch.randn(B, C, T, device=dev)
        w1 = torch.randn(C, C, 3, device=dev) * 0.05
        w2, ws = torch.randn(C, C, 1, device=dev) * 0.05, torch.randn(C, C, 1, device=dev) * 0.05
        b = torch.randn(C, device=dev)
        img = ops.resstack_pack_weight(w1, None, w2, None, ws, None)
        blk = ResidualStack(channels=C, dilation=d).to(dev)
        blk.fuse_unit = False
        fl = 2.0 * B * T * C * C * 5
        with torch.no_grad():
            t3 = timeit(lambda: blk(x))
            t1 = timeit(lambda: ops.resstack_forward(x, img, d, 0.2, b, b, b, save_h=False))
            t1h = timeit(lambda: ops.resstack_forward(x, img, d, 0.2, b, b, b, save_h=True))
        print(f"C{C} T{T} d{d:2d}: three launches {t3*1e3:7.1f} us ({fl/t3/1e9:5.1f} TF) | one launch {t1*1e3:7.1f} us ({fl/t1/1e9:5.1f} TF)"
              f" | + h out {t1h*1e3:7.1f} us", flush=True)
