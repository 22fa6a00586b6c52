"""Do the WaveNet layer's data-gradient chain (gate + dgrad, caller's stream) and its weight-gradient launches (side
stream) overlap?  Eager launches at the C2 batch (B6 x 25600): serial on one stream vs forked, 30 layers per pass.
usage: bench_wavenet_overlap.py"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parallelwavegan_amd import ops

dev = torch.device("cuda:0")
B, T = 6, 25600
z = torch.randn(B, 128, T, device=dev)
x, c, g = torch.randn(B, 64, T, device=dev), torch.randn(B, 80, T, device=dev), torch.randn(B, 64, T, device=dev)
gs, dxo = torch.randn(B, 64, T, device=dev), torch.randn(B, 64, T, device=dev)
w = [torch.randn(128, 64, 3, device=dev) * .07, torch.randn(128, 80, 1, device=dev) * .1, torch.randn(64, 64, 1, device=dev) * .1,
     torch.randn(64, 64, 1, device=dev) * .1]
descs = [ops.make_wavenet_desc(B, T, 2 ** (i % 10), out_mul=math.sqrt(.5)) for i in range(30)]
imgs = [ops.wavenet_pack_weights_bwd(d, w[0], None, w[1], None, w[2], None, w[3], None) for d in descs[:10]]
side = torch.cuda.Stream()


def one_pass(mode):
    cur = torch.cuda.current_stream()
    for i, d in enumerate(descs):
        img = imgs[i % 10]
        dz, go = ops.wavenet_gate_backward(d, z, dxo, gs, img)
        if mode == "serial":
            ops.wavenet_weight_backward(d, dz, x, c, gs, go, g)
            ops.wavenet_data_backward(d, dz, go, img)
        elif mode == "fork":
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                ops.wavenet_weight_backward(d, dz, x, c, gs, go, g)
            dz.record_stream(side); go.record_stream(side)
            ops.wavenet_data_backward(d, dz, go, img)
        elif mode == "chain_only":
            ops.wavenet_data_backward(d, dz, go, img)
        elif mode == "wgrad_only":
            ops.wavenet_weight_backward(d, dz, x, c, gs, go, g)
    cur.wait_stream(side)


for mode in ("serial", "fork", "chain_only", "wgrad_only", "serial", "fork"):
    for _ in range(2):
        one_pass(mode)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(5):
        one_pass(mode)
    e1.record(); torch.cuda.synchronize()
    print(f"{mode:10s}: {e0.elapsed_time(e1) / 5:7.2f} ms per 30-layer backward", flush=True)
