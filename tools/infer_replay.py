"""N hipGraph replays of the HiFi-GAN V1 generator forward at (batch, frames) exactly as bench.py's latency legs run them
(the command tools/graph_gaps.py's kernel trace is taken from).  usage: infer_replay.py [batch] [frames] [replays]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from parallelwavegan_amd.graphs import GraphedInference  # noqa: E402
from parallelwavegan_amd.models import HiFiGANGenerator  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
F = int(sys.argv[2]) if len(sys.argv) > 2 else 100
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda:0")
torch.manual_seed(0)
g = HiFiGANGenerator(**bench.load_conf("hifigan.v1")["generator_params"])
g.remove_weight_norm()
g = g.to(dev).eval()
g.branch_streams = True
run = GraphedInference(g)
c = torch.randn(B, 80, F, device=dev)
with torch.no_grad():
    for _ in range(5):
        run(c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        run(c)
    torch.cuda.synchronize()
print(f"B{B} x {F} frames: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per replayed forward")
