"""Probe: does the weight-gradient kernel produce NaN when the LDS holds NaN patterns from an earlier
kernel?  (Edge chunks used to leave part of the X tile unwritten; columns past n_cols meet G = 0 but
0 * NaN = NaN.)  usage: wgrad_stale_lds.py  (GPU box)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from parallelwavegan_amd import ops
dev = torch.device("cuda:0")
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lds_poison.so"))
sink = torch.zeros(4, device=dev)
B, Cin, Cout, T, K, pad = 2, 1, 16, 400, 15, 7
g = torch.Generator().manual_seed(1)
x = torch.randn(B, Cin, T, generator=g).to(dev); dy = torch.randn(B, Cout, T, generator=g).to(dev)
desc = ops.make_conv_desc(B, Cin, Cout, T, T, K, 1, 1, pad, 1)
nbad = 0
for it in range(100):
    lib.lds_poison(ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    dw, db = ops.conv1d_backward_weight(desc, x, dy, (Cout, Cin, K))
    nbad += int(not (torch.isfinite(dw).all() and torch.isfinite(db).all()))
print("non-finite weight gradients:", nbad, "of 100")
