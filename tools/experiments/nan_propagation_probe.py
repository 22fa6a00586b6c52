"""Round 6: does a NaN first-layer output of a HiFi-GAN scale discriminator propagate to its later feature maps?
(The eager multi-stream anomaly shows map 0 of scale discriminator 1 all-NaN with maps 1.. finite.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from parallelwavegan_amd.utils import build_from_config  # noqa: E402

dev = torch.device("cuda:0")
conf = bench.load_conf("hifigan.v1")
torch.manual_seed(0)
model, *_ = build_from_config(conf, dev)
d = model["discriminator"]
x = 0.3 * torch.randn(16, 1, 8192, device=dev)
for mode in ("clean", "nan-bias", "nan-input-of-disc1"):
    sub = d.msd.discriminators[1]
    conv0 = sub.layers[0][0]
    saved = conv0.bias.detach().clone()
    if mode == "nan-bias":
        with torch.no_grad():
            conv0.bias.fill_(float("nan"))
    outs = d(x)
    with torch.no_grad():
        conv0.bias.copy_(saved)
    print(mode, "scale discriminator 1:", ["NaN" if not bool(torch.isfinite(m).all()) else "ok" for m in outs[1]],
          "| train mode", d.training, "| grad", torch.is_grad_enabled(), "| types", type(outs[1]).__name__)
    if mode == "nan-input-of-disc1":
        pass
