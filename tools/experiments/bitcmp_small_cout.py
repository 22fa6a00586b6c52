"""Outputs of the few-output-channel layers (HiFi-GAN 32 -> 1 k7 + tanh, PWG 64 -> 1 k1) and of one PWG upsampling stage
as raw bytes -> sha256, so that two runs with different PWG_SMALL_COUT_STREAM / PWG_STRETCH_FAST settings can be
compared bit for bit.  usage: python tools/experiments/bitcmp_small_cout.py"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from parallelwavegan_amd import layers  # noqa: E402
from parallelwavegan_amd.layers.conv import Conv1d  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(3)
with torch.no_grad():
    for cin, k, t, pre, post in ((32, 7, 8192, "leaky_relu", "tanh"), (64, 1, 25600, "relu", None), (24, 5, 4100, None, None)):
        conv = Conv1d(cin, 1, k, padding=(k - 1) // 2).to(dev)
        x = torch.randn(3, cin, t, device=dev)
        y = conv(x, pre_act=pre, pre_slope=0.01, post_act=post)
        print(f"conv {cin}->1 k{k} T{t}: {hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:16]}")
    up = layers.UpsampleNetwork([4, 4]).to(dev)
    c = torch.randn(2, 80, 50, device=dev)
    y = up(c)
    print(f"upsample x16: {hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:16]}")
