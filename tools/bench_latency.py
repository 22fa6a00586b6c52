"""Small-batch HiFi-GAN V1 inference latency: eager vs hipGraph replay."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from parallelwavegan_amd.models import HiFiGANGenerator
from parallelwavegan_amd.graphs import GraphedInference
dev = torch.device("cuda:0")
torch.manual_seed(0)
g = HiFiGANGenerator(**bench.HIFIGAN_V1); g.remove_weight_norm(); g = g.to(dev).eval()
gg = GraphedInference(g)
import copy
gb = copy.deepcopy(g); gb.branch_streams = True
ggb = GraphedInference(gb)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for B, F in ((1, 100), (1, 800), (4, 800), (16, 800)):
    c = torch.randn(B, 80, F, device=dev)
    with torch.no_grad():
        te = t(lambda: g(c)); tg = t(lambda: gg(c)); tb = t(lambda: ggb(c))
        err = (g(c) - gg(c)).abs().max().item(); errb = (g(c) - ggb(c)).abs().max().item()
    n = B * F * 256
    print(f"B={B:2d} F={F:4d}: eager {te*1e3:7.3f} ms ({n/te/1e6:6.2f} Msamp/s)  graph {tg*1e3:7.3f} ms ({n/tg/1e6:6.2f})  graph+branches {tb*1e3:7.3f} ms ({n/tb/1e6:6.2f} Msamp/s)  max|diff| {err:.1e} {errb:.1e}")
