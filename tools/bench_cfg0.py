#!/usr/bin/env python3
"""FAST 128x128x8 (tile config 0) against the heuristic's 128x128x4 (config 2) on the k = 3 / 7 MRF layers of the
bench shape (B16 x 800 frames), with the epilogue the generator uses (bias + residual); prints us, TFLOP/s, max diff."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parallelwavegan_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for c, t, k, d in [(128, 51200, 3, 1), (128, 51200, 3, 5), (256, 6400, 3, 1), (128, 51200, 7, 1), (256, 6400, 7, 3)]:
    for pre in ("leaky_relu", None):
        desc = ops.make_conv_desc(16, c, c, t, t, k, dilation=d, pad_left=(k - 1) // 2 * d, pre_act=pre, pre_slope=0.1)
        x = torch.randn(16, c, t, device=dev)
        w = torch.randn(c, c, k, device=dev) / (c * k) ** 0.5
        b = torch.randn(c, device=dev)
        wp = ops.pack_weight(desc, w)
        outs = {}
        line = f"C={c} T={t} k={k} d={d} pre={pre}:"
        for cfg in (2, 0):
            y = torch.empty_like(x)
            for _ in range(2):
                ops.conv1d_forward_cfg(desc, x, wp, b, x, None, y, tile_config=cfg, use_dma=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.conv1d_forward_cfg(desc, x, wp, b, x, None, y, tile_config=cfg, use_dma=True)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 5 * 1e3
            outs[cfg] = y
            line += f"  cfg{cfg} {us:7.1f} us {2.0 * 16 * c * c * k * t / us * 1e-6:6.1f} TF"
        print(line, f" max diff {(outs[0] - outs[2]).abs().max().item():.2e}", flush=True)
