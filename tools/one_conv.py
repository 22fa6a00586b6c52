"""Run one conv shape repeatedly (for rocprofv3 PMC collection). args: C K D T B [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parallelwavegan_amd import ops
C, K, D, T, B = (int(v) for v in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 10
dev = torch.device("cuda:0")
desc = ops.make_conv_desc(B, C, C, T, T, K, dilation=D, pad_left=(K - 1) // 2 * D, pre_act="leaky_relu", pre_slope=0.1)
w = torch.randn(C, C, K, device=dev) * 0.05
wp = ops.pack_weight(desc, w)
x = torch.randn(B, C, T, device=dev); bias = torch.randn(C, device=dev); add1 = torch.randn(B, C, T, device=dev)
y = torch.empty(B, C, T, device=dev)
for _ in range(reps):
    ops.conv1d_forward(desc, x, wp, bias, add1, out=y)
torch.cuda.synchronize()
