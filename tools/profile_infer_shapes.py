"""Per-(kernel, shape) times of ONE eager HiFi-GAN V1 generator forward at (batch, frames) (PWG_PROF_SHAPES=1): where a
batch-1 utterance's latency goes, launch by launch.  usage: python tools/profile_infer_shapes.py [batch] [frames] [top]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PWG_PROF_SHAPES", "1")
import torch  # noqa: E402

import bench  # noqa: E402
from parallelwavegan_amd import ops  # noqa: E402
from parallelwavegan_amd.models import HiFiGANGenerator  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
F = int(sys.argv[2]) if len(sys.argv) > 2 else 100
top = int(sys.argv[3]) if len(sys.argv) > 3 else 80
dev = torch.device("cuda:0")
torch.manual_seed(0)
g = HiFiGANGenerator(**bench.load_conf("hifigan.v1")["generator_params"])
g.remove_weight_norm()
g = g.to(dev).eval()
c = torch.randn(B, 80, F, device=dev)
with torch.no_grad():
    for _ in range(3):
        g(c)
    torch.cuda.synchronize()
    with ops.profile() as prof:
        for _ in range(5):
            g(c)
rows = sorted(prof.results.items(), key=lambda kv: -kv[1]["ms"])
tot = sum(v["ms"] for _, v in rows) / 5
print(f"B{B} x {F} frames: {len(rows)} distinct (kernel, shape) rows, {sum(v['launches'] for _, v in rows) // 5} launches, "
      f"{tot:.3f} ms of kernel time per forward (serial, eager)")
acc = 0.0
for name, v in rows[:top]:
    ms = v["ms"] / 5
    acc += ms
    tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["flops"] else 0.0
    print(f"{ms * 1e3:8.1f} us {100 * ms / tot:5.1f}% cum {100 * acc / tot:5.1f}%  n={v['launches'] // 5:3d}  {tf:6.1f} TF  {name}")
