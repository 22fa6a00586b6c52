"""Run the one-launch WaveNet layer repeatedly (for rocprofv3 PMC collection). args: B T dilation save [reps]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parallelwavegan_amd import ops
B, T, D, SAVE = (int(v) for v in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 6
dev = torch.device("cuda:0")
x, c, s = torch.randn(B, 64, T, device=dev), torch.randn(B, 80, T, device=dev), torch.randn(B, 64, T, device=dev)
w = [torch.randn(128, 64, 3, device=dev) * .07, torch.randn(128, 80, 1, device=dev) * .1, torch.randn(64, 64, 1, device=dev) * .1,
     torch.randn(64, 64, 1, device=dev) * .1]
b = [torch.randn(128, device=dev), torch.randn(64, device=dev), torch.randn(64, device=dev)]
desc = ops.make_wavenet_desc(B, T, D, out_mul=math.sqrt(.5))
img = ops.wavenet_pack_weights(desc, w[0], None, w[1], None, w[2], None, w[3], None)
so = torch.empty_like(s)
for _ in range(reps):
    ops.wavenet_layer_forward(desc, x, c, s, img, b[0], b[1], b[2], save=bool(SAVE), skips_out=so)
torch.cuda.synchronize()
print(f"wavenet layer B{B} T{T} d{D} save{SAVE}: algorithmic bytes per launch = {4 * B * T * (64 * 4 + 80 + (192 if SAVE else 0))}")
