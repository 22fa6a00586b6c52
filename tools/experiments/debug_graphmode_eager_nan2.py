"""Debugging aid (round 5): where does the sporadic NaN of the eager branch-stream mode enter?  Finite-flags of every
discriminator feature map are taken WITHOUT host synchronisation (a) right after each discriminator forward and (b)
when the feature-matching loss reads the maps; printed after the run."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_train_full_shape_gpu import _build, load_golden  # noqa: E402
from tests.util import poison_empty, poison_lds  # noqa: E402

from parallelwavegan_amd import streams  # noqa: E402

streams.EAGER_FORK = True  # (debugging aid: the product forks only under capture)
NODETAIL = os.environ.get("NAN2_NODETAIL") == "1"
dev = torch.device("cuda:0")
gold = load_golden("c3_train_full")
flags = []  # (label, device bool tensor)
detail = []
with poison_lds(), poison_empty():
    tr, batch, model, opt = _build("c3", gold, dev, use_hip_graph=True, graph_warmup_steps=100)
    disc = model["discriminator"]
    d_forward = disc.forward
    calls = [0]

    def fwd(x, *a, **k):
        out = d_forward(x, *a, **k)
        calls[0] += 1
        flags.append((f"step {tr.steps} D-call {calls[0]} input", torch.isfinite(x).all()))
        for i, maps in enumerate(out):
            for j, m in enumerate(maps or []):
                flags.append((f"step {tr.steps} D-call {calls[0]} produced disc {i} map {j} {tuple(m.shape)}", torch.isfinite(m).all()))
                if j == 0 and i in (0, 1, 2) and not NODETAIL:  # where inside the map are the non-finite values? (no host sync)
                    bad = ~torch.isfinite(m.reshape(-1))
                    b8 = bad.to(torch.int8)
                    detail.append((f"step {tr.steps} D-call {calls[0]} disc {i} map 0 ptr {m.data_ptr():#x} numel {m.numel()}",
                                   bad.sum(), torch.argmax(b8), m.numel() - 1 - torch.argmax(b8.flip(0)),
                                   bad.reshape(m.shape[0], -1).sum(dim=1), bad.reshape(-1, m.shape[-1]).sum(dim=1)[:256]))
        return out

    disc.forward = fwd
    fm = tr.criterion["feat_match"]
    fm_forward = fm.forward

    def fmf(feats_hat, feats):
        for name, ff in (("hat", feats_hat), ("real", feats)):
            for i, maps in enumerate(ff):
                for j, m in enumerate(maps):
                    flags.append((f"step {tr.steps} FM-read {name} disc {i} map {j}", torch.isfinite(m).all()))
        return fm_forward(feats_hat, feats)

    fm.forward = fmf
    for _ in range(5):
        tr._train_step(batch)
    torch.cuda.synchronize()
bad = [lab for lab, f in flags if not bool(f.item())]
print(f"RESULT {len(bad)} non-finite of {len(flags)} checks")
for lab in bad[:12]:
    print("  ", lab)
for lab, n, first, last, per_item, per_row in detail:
    if int(n.item()):
        print("DETAIL", lab, "non-finite", int(n.item()), "first flat index", int(first.item()), "last", int(last.item()))
        print("   per batch item:", per_item.tolist())
        print("   per (item, channel) row, first 256 rows:", per_row.tolist())
