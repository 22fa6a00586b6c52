#!/usr/bin/env python3
"""Debug aid: dump every gradient the first optimizer steps of the MB-MelGAN two-step test consume, so that two
runs (e.g. with / without PWG_POISON_LDS=1) can be compared tensor by tensor.
usage: debug_poison_grads.py dump <file> | compare <a> <b>"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def dump(path):
    from parallelwavegan_amd import optimizers
    import tests.test_pwg_mb_train_gpu as T

    out = []
    for cls in (optimizers.Adam, optimizers.RAdam):
        orig = cls.step

        def step(self, *a, _orig=orig, **k):
            out.append([(tuple(p.shape), None if p.grad is None else p.grad.detach().cpu().clone())
                        for g in self.param_groups for p in g["params"]])
            return _orig(self, *a, **k)

        cls.step = step
    try:
        T.test_mb_melgan_v2_two_train_steps(torch.device("cuda:0"))
    except AssertionError as e:
        print("test assertion:", str(e)[:200])
    torch.save(out, path)
    print("dumped", len(out), "optimizer steps")


def compare(a, b):
    A, B = torch.load(a), torch.load(b)
    for si, (sa, sb) in enumerate(zip(A, B)):
        for pi, ((sh, ga), (_, gb)) in enumerate(zip(sa, sb)):
            if ga is None or gb is None:
                continue
            d = (ga - gb).abs().max().item()
            if d > 0 or torch.isnan(ga).any() or torch.isnan(gb).any():
                print(f"step {si} param {pi} shape {sh}: max diff {d:.3e} (|a| max {ga.abs().max():.3e}) nan {int(torch.isnan(ga).sum())}/{int(torch.isnan(gb).sum())}")
        if si >= 1:
            break


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2])
    else:
        compare(sys.argv[2], sys.argv[3])
