"""Debugging aid (round 5): six eager steps of C3 under the graph mode's launch plans (branch streams + concurrency
hint) with poisoned LDS / torch.empty, in a fresh process; variants by environment:
  CASE=a  use_hip_graph=True, graph_warmup_steps=100 (the failing configuration of the first strict graph==eager run)
  CASE=b  same, branch_streams=False            CASE=c  same, conv_concurrency_hint=1.0
  CASE=d  plain eager                           INSPECT=1  check every feature map of both passes for NaN per step
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_train_full_shape_gpu import _build, load_golden  # noqa: E402
from tests.util import poison_empty, poison_lds  # noqa: E402

tag = os.environ.get("TAG", "c3")
case = os.environ.get("CASE", "a")
cfg = {"a": dict(use_hip_graph=True, graph_warmup_steps=100),
       "b": dict(use_hip_graph=True, graph_warmup_steps=100, branch_streams=False),
       "c": dict(use_hip_graph=True, graph_warmup_steps=100, conv_concurrency_hint=1.0),
       "d": dict(use_hip_graph=False)}[case]
dev = torch.device("cuda:0")
gold = load_golden(f"{tag}_train_full")
ctx = [poison_lds(), poison_empty()] if os.environ.get("POISON", "1") == "1" else []
for c in ctx:
    c.__enter__()
tr, batch, model, opt = _build(tag, gold, dev, **cfg)
if os.environ.get("INSPECT") == "1":
    fm = tr.criterion["feat_match"]
    orig = fm.forward

    def forward(feats_hat, feats):
        torch.cuda.synchronize()
        for name, ff in (("hat", feats_hat), ("real", feats)):
            for i, maps in enumerate(ff):
                for j, m in enumerate(maps):
                    if not torch.isfinite(m).all():
                        bad = (~torch.isfinite(m)).sum().item()
                        print(f"  step {tr.steps}: {name} disc {i} map {j} shape {tuple(m.shape)}: {bad} non-finite of {m.numel()}",
                              flush=True)
        return orig(feats_hat, feats)

    fm.forward = forward
for _ in range(int(os.environ.get("STEPS", "4"))):
    tr._train_step(batch)
torch.cuda.synchronize()
for step, losses in tr.loss_history():
    bad = [k.split("/")[-1] for k, v in losses.items() if not np.isfinite(v)]
    print(f"case {case} step {step}: {'FINITE' if not bad else 'NON-FINITE ' + ','.join(bad)}", flush=True)
