"""Round 6: which write puts the NaN into the feature map?  tools/experiments/debug_graphmode_eager_nan2.py
plus an in-memory log of every poison fill (pointer range, stream) and every output of ops.conv1d_forward (pointer
range, stream, kernel geometry); when a map turns out non-finite, every logged event that overlaps its memory is
printed in program order.  Run with PWG_EAGER_BRANCH_STREAMS=1."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from parallelwavegan_amd import ops  # noqa: E402
from tests.test_train_full_shape_gpu import _build, load_golden  # noqa: E402
from tests.util import poison_lds  # noqa: E402

from parallelwavegan_amd import streams  # noqa: E402

streams.EAGER_FORK = True  # (debugging aid: the product forks only under capture)
dev = torch.device("cuda:0")
LOG = []  # (seq, kind, ptr, nbytes, stream, note)


def cur():
    return torch.cuda.current_stream().cuda_stream


class poison_logged:
    def __enter__(self):
        self._orig = (torch.empty, torch.empty_like, torch.Tensor.new_empty)
        o_empty, o_like, o_new = self._orig

        def fill(t, how):
            if t.is_cuda and t.is_floating_point() and t.numel():
                LOG.append((len(LOG), "alloc+fill:" + how, t.data_ptr(), t.numel() * t.element_size(), cur(), tuple(t.shape)))
                t.fill_(float("nan"))
            return t

        torch.empty = lambda *a, **k: fill(o_empty(*a, **k), "empty")
        torch.empty_like = lambda *a, **k: fill(o_like(*a, **k), "empty_like")
        torch.Tensor.new_empty = lambda self_, *a, **k: fill(o_new(self_, *a, **k), "new_empty")
        return self

    def __exit__(self, *exc):
        torch.empty, torch.empty_like, torch.Tensor.new_empty = self._orig
        return False


orig_fwd = ops.conv1d_forward


def fwd_logged(desc, x, *a, **k):
    y = orig_fwd(desc, x, *a, **k)
    LOG.append((len(LOG), "conv1d_forward out", y.data_ptr(), y.numel() * 4, cur(),
                f"Cin{desc.c_in} Cout{desc.c_out} Tin{desc.t_in} k{desc.kernel} s{desc.stride} g{desc.groups} W{desc.width} in_ptr {x.data_ptr():#x}"))
    return y


ops.conv1d_forward = fwd_logged
orig_rec = torch.Tensor.record_stream


def rec_logged(self, s):
    LOG.append((len(LOG), "record_stream", self.data_ptr(), self.numel() * self.element_size(), cur(), f"-> stream {s.cuda_stream:#x}"))
    return orig_rec(self, s)


torch.Tensor.record_stream = rec_logged

gold = load_golden("c3_train_full")
flags, maps_seen = [], []
with poison_lds(), poison_logged():
    tr, batch, model, opt = _build("c3", gold, dev, use_hip_graph=True, graph_warmup_steps=100)
    disc = model["discriminator"]
    d_forward = disc.forward
    calls = [0]

    def fwd(x, *a, **k):
        LOG.append((len(LOG), f"--- D-call {calls[0] + 1} begins (step {tr.steps})", 0, 0, cur(), ""))
        out = d_forward(x, *a, **k)
        calls[0] += 1
        LOG.append((len(LOG), f"--- D-call {calls[0]} returned", 0, 0, cur(), ""))
        for i, maps in enumerate(out):
            for j, m in enumerate(maps or []):
                flags.append((f"step {tr.steps} D-call {calls[0]} disc {i} map {j} {tuple(m.shape)}", torch.isfinite(m).all(),
                              m.data_ptr(), m.numel() * 4, len(LOG)))
        return out

    disc.forward = fwd
    for _ in range(5):
        tr._train_step(batch)
    torch.cuda.synchronize()
bad = [(lab, p, n, at) for lab, f, p, n, at in flags if not bool(f.item())]
print(f"RESULT {len(bad)} non-finite of {len(flags)} checks; main stream {torch.cuda.default_stream().cuda_stream:#x}")
for lab, p, n, at in bad[:2]:
    print("NON-FINITE", lab, f"ptr {p:#x} bytes {n} (checked at log position {at})")
    lo = max(0, at - 4000)
    for seq, kind, q, m, s, note in LOG[lo:at + 400]:
        if kind.startswith("---") or (q < p + n and p < q + m):
            print(f"   [{seq}] {kind:28s} ptr {q:#x} +{m} on stream {s:#x} {note}")
