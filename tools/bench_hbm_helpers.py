"""HBM-bound helper kernels at the BASELINE training shapes: achieved GB/s (algorithmic bytes / HIP-event time of the
launch, events recorded inside the library on the launch stream) against the ~6.3 TB/s a streaming kernel reaches on
MI355X (MI355X_MICROARCH.md) -- VERDICT r04 item 5.  Also the time of a hipGraph replay of the same launch, i.e. what a
captured training step pays (event pairs around one eager launch include the gap the host leaves).

    python tools/bench_hbm_helpers.py > profiles/r05_hbm_helpers.txt
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parallelwavegan_amd import functional as Fn  # noqa: E402
from parallelwavegan_amd import layers, ops  # noqa: E402
from parallelwavegan_amd.layers.conv import Conv1d, ConvTranspose1d  # noqa: E402

ACHIEVABLE = 6300.0  # GB/s
dev = torch.device("cuda:0")
rows = []


def graph_us(fn, reps=20):
    """Per-launch time of `reps` back-to-back replays inside ONE captured graph (no host gaps)."""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)


def measure(label, fn, kernel=None, bytes_override=None, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    with ops.profile() as prof:
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
    res = prof.results
    if kernel is None:
        kernel = max(res, key=lambda k: res[k]["ms"])
    r = res[kernel]
    us = r["ms"] * 1e3 / r["launches"]
    by = bytes_override if bytes_override is not None else r["bytes"] / r["launches"]
    per_call = sum(v["launches"] for v in res.values()) / reps
    gus = graph_us(fn) if per_call == 1 else float("nan")  # (a call that launches several kernels: eager figure only)
    rows.append((label, kernel, by / 1e6, us, by / us / 1e3, gus, by / gus / 1e3))


with torch.no_grad():
    # ---- PQMF at the C4 batch (B = 64 x 16384): north_star names it as an HBM-roofline kernel
    pq = layers.PQMF(4).to(dev)
    y = torch.randn(64, 1, 16384, device=dev)
    sub = pq.analysis(y)
    measure("PQMF analysis  B64 x 16384 -> 4 bands", lambda: pq.analysis(y), "pqmf_down_kernel")
    measure("PQMF synthesis B64 x 4 x 4096 -> 16384", lambda: pq.synthesis(sub), "pqmf_up_kernel")
    y2 = torch.randn(16, 1, 204800, device=dev)
    sub2 = pq.analysis(y2)
    measure("PQMF analysis  B16 x 204800 (inference batch)", lambda: pq.analysis(y2), "pqmf_down_kernel")
    measure("PQMF synthesis B16 x 204800 (inference batch)", lambda: pq.synthesis(sub2), "pqmf_up_kernel")
    # ---- single-input-channel first layers of the scale discriminators
    for b, t, cout, tag in ((16, 8192, 128, "C3 MSD"), (16, 4097, 128, "C3 MSD pooled"), (64, 16384, 16, "C4 MelGAN D")):
        conv = Conv1d(1, cout, 15, padding=7).to(dev)
        x = torch.randn(b, 1, t, device=dev)
        measure(f"Conv1d 1 -> {cout} k15  B{b} x {t} ({tag})", lambda conv=conv, x=x: conv(x), "conv1d_small_cin_kernel")

    # ---- few-output-channel last layers of the generators (round 6: LDS-free stream; PWG_SMALL_COUT_STREAM=0 = the LDS kernel)
    for b, t, cin, k, tag in ((16, 204800, 32, 7, "HiFi-GAN output layer, inference batch"), (16, 8192, 32, 7, "HiFi-GAN output layer, C3 batch"),
                              (6, 25600, 64, 1, "PWG last layer, C2 batch")):
        conv = Conv1d(cin, 1, k, padding=(k - 1) // 2).to(dev)
        x = torch.randn(b, cin, t, device=dev)
        measure(f"Conv1d {cin} -> 1 k{k}  B{b} x {t} ({tag})", lambda conv=conv, x=x: conv(x, pre_act="leaky_relu", pre_slope=0.01, post_act="tanh"))
# ---- backward-side helpers (need autograd)
ct = ConvTranspose1d(64, 32, 4, 2, padding=1).to(dev)
x = torch.randn(16, 64, 4096, device=dev)
go = torch.randn(16, 32, 8192, device=dev)


def ct_bwd():
    for p in ct.parameters():
        p.grad = None
    ct(x).backward(go)


measure("bias gradient of ConvTranspose1d 64 -> 32, dy B16 x 32 x 8192", ct_bwd, "bias_grad_kernel")
a = torch.randn(16, 128, 2048, device=dev)
ya = torch.randn(16, 128, 2048, device=dev)
out = torch.empty_like(a)
from parallelwavegan_amd import _lib  # noqa: E402

measure("activation backward B16 x 128 x 2048", lambda: _lib.check(_lib.lib().pwg_act_backward(
    ops._ptr(a), ops._ptr(ya), ops._ptr(out), a.numel(), ops.ACT["leaky_relu"], 0.1, 1.0, ops._stream())), "act_backward_kernel",
    bytes_override=12.0 * a.numel())
up = layers.UpsampleNetwork([4, 4, 4, 4]).to(dev)
c = torch.randn(6, 80, 104, device=dev)
with torch.no_grad():
    measure("PWG upsampling stage (stretch x 4 + k9), C2 batch", lambda: up(c), "stretch_conv_fwd_kernel")

st = layers.UpsampleNetwork([4]).to(dev)  # ONE stage at the last stage's size (C2: 6 x 80 x 6400 -> 25600)
c1 = torch.randn(6, 80, 6400, device=dev)
with torch.no_grad():
    measure("PWG upsampling, last stage alone (6 x 80 x 6400 -> 25600)", lambda: st(c1), "stretch_conv_fwd_kernel")

print(f"{'kernel / shape':58s} {'MB':>8s} {'eager us':>9s} {'GB/s':>8s} {'frac':>6s} {'graph us':>9s} {'GB/s':>8s} {'frac':>6s}")
for label, kernel, mb, us, gbs, gus, ggbs in rows:
    print(f"{label:58s} {mb:8.2f} {us:9.1f} {gbs:8.0f} {gbs / ACHIEVABLE:6.2f} {gus:9.1f} {ggbs:8.0f} {ggbs / ACHIEVABLE:6.2f}   [{kernel}]")
print("frac = achieved GB/s / 6300 (achievable streaming rate); 'graph us' = per launch inside one captured graph of 20 launches")
