"""Probe for the eager multi-stream anomaly of round 5 (profiles/r05_eager_branch_streams_nan.txt) with
PLAIN TORCH OPS ONLY -- no kernel of this library: the fork / join / record_stream pattern of
parallelwavegan_amd.streams.run_branches, NaN-poisoned allocations, a producer and a consumer per side stream, a check
on the joined stream.  If this reproduces, the cause is below this engine (allocator / HIP runtime / queue scheduling).

    python tools/probes/multistream_nan_probe.py [iterations] [n_streams]
"""
import sys

import torch

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n_streams = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
cur = torch.cuda.current_stream(dev)
side = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
sizes = [16 * 128 * 8192, 16 * 128 * 4097, 16 * 128 * 2049] + [16 * 32 * 1366 * 2, 16 * 32 * 911 * 3, 16 * 32 * 547 * 5,
                                                                16 * 32 * 391 * 7, 16 * 32 * 249 * 11]
sizes = (sizes * 4)[:n_streams]
x = torch.randn(16 * 8192, device=dev)
bad = []  # (iteration, stream index, where) with device flags; read at the end (no host sync inside the loop)
flags = []


def poisoned(n):
    t = torch.empty(n, device=dev)
    t.fill_(float("nan"))
    return t


keep = []
for it in range(iters):
    fork = torch.cuda.Event()
    fork.record(cur)
    outs = []
    for k, s in enumerate(side):
        s.wait_event(fork)
        with torch.cuda.stream(s):
            n = sizes[k]
            y = poisoned(n)                                   # "feature map 0": allocation poison, then the producer
            y.copy_(x.repeat((n + x.numel() - 1) // x.numel())[:n] * 0.5)
            tmp = poisoned(n // 4)                            # scratch that is freed at once (allocator churn)
            tmp.copy_(y[: n // 4])
            z = poisoned(n // 4)                              # "feature map 1": computed from map 0 on the same stream
            torch.mul(y[: n // 4], 2.0, out=z)
            del tmp
            outs.append((y, z))
    for s in side:
        cur.wait_stream(s)
    for y, z in outs:
        y.record_stream(cur)
        z.record_stream(cur)
    for k, (y, z) in enumerate(outs):
        flags.append((it, k, "map0", torch.isfinite(y).all()))
        flags.append((it, k, "map1", torch.isfinite(z).all()))
    keep = outs  # freed one iteration later, as a training step's feature maps are
torch.cuda.synchronize()
bad = [(it, k, w) for it, k, w, f in flags if not bool(f.item())]
print(f"multistream probe: {iters} iterations x {n_streams} streams: {len(bad)} non-finite checks of {len(flags)}")
for b in bad[:20]:
    print("   iteration %d stream %d %s" % b)
