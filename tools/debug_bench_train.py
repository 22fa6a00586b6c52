#!/usr/bin/env python3
"""Debug aid: run the training step of BASELINE configs exactly as bench.py's ``bench_train`` builds them, in ONE
process and in the given order, and report every loss of every step (from the trainer's device-side loss
history: no host sync between the steps unless --sync) plus the first non-finite parameter / moment.

usage: debug_bench_train.py <c2|c3|c4>[,<tag>...] [steps] [--no-graph] [--sync] [--quiet] [--loop N] [--gc]
env:   PWG_POISON_LDS=1 (NaN-fill the LDS before every MFMA launch), PWG_POISON_EMPTY=1 (NaN-fill torch.empty)"""
import contextlib
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def first_bad(model, opt):
    bad = []
    for key in ("generator", "discriminator"):
        for n, p in model[key].named_parameters():
            if not torch.isfinite(p).all():
                bad.append((key, n, "param", int((~torch.isfinite(p)).sum())))
            if p.grad is not None and not torch.isfinite(p.grad).all():
                bad.append((key, n, "grad", int((~torch.isfinite(p.grad)).sum())))
            st = opt[key].state.get(p, {})
            for sk, sv in st.items():
                if torch.is_tensor(sv) and sv.is_floating_point() and not torch.isfinite(sv).all():
                    bad.append((key, n, sk, int((~torch.isfinite(sv)).sum())))
    return bad


def install_stft_debug(tr):
    """PWG_DBG_STFT=1: log the two losses of every STFT resolution as extra "losses" (they ride in the graph's
    loss vector); PWG_DBG_STFT=unfused: use the op-by-op STFT loss chain instead of the fused kernel."""
    mode = os.environ.get("PWG_DBG_STFT")
    if not mode:
        return
    from parallelwavegan_amd.losses import stft_loss as SL

    if mode == "unfused":
        SL.STFTLoss.fused = False
        return

    def fwd(self, x, y):
        l2 = self.stft_magnitude.pair_losses(x, y)
        for j in range(2):
            tr._log(f"dbg/fft{self.fft_size}_l{j}", l2[j])
        return l2[0], l2[1]

    SL.STFTLoss.forward = fwd


def run_one(tag, steps, graph, sync, quiet):
    from parallelwavegan_amd.bin.train import Trainer
    from parallelwavegan_amd.utils import build_from_config

    dev = torch.device("cuda:0")
    conf = bench.load_conf(bench.TRAIN_CONFIGS[tag])
    torch.manual_seed(4321)
    model, criterion, opt, sched = build_from_config(conf, dev)
    conf.update(generator_train_start_steps=0, discriminator_train_start_steps=0, train_max_steps=10 ** 9,
                save_interval_steps=10 ** 9, eval_interval_steps=10 ** 9, log_interval_steps=10 ** 9,
                distributed=False, rank=0, outdir=tempfile.mkdtemp(), progress=False, use_hip_graph=graph,
                graph_warmup_steps=2, record_loss_history=True)
    batch = bench.synthetic_batch(conf, conf["batch_size"], dev, 0)
    tr = Trainer(steps=1, epochs=0, data_loader={"train": [batch], "dev": [batch]}, sampler={"train": None, "dev": None},
                 model=model, criterion=criterion, optimizer=opt, scheduler=sched, config=conf, device=dev)
    tr.tqdm = None
    install_stft_debug(tr)
    for _ in range(steps):
        tr._train_step(batch)
        if sync:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    ok = True
    for step, d in tr.loss_history():
        fin = all(v == v and abs(v) != float("inf") for v in d.values())
        if not quiet or not fin:
            print(f"step {step} " + " ".join(f"{k.split('/')[-1]}={v:.5g}" for k, v in d.items()), flush=True)
        if not fin:
            ok = False
            break
    if not ok:
        for b in first_bad(model, opt)[:12]:
            print("   BAD", b)
    print("RESULT", tag, "graph" if graph else "eager", "sync" if sync else "async",
          "lds-poison" if os.environ.get("PWG_POISON_LDS") else "", "empty-poison" if os.environ.get("PWG_POISON_EMPTY") else "",
          "finite" if ok else "NONFINITE", flush=True)
    del tr, model, criterion, opt, sched
    torch.cuda.empty_cache()


def main():
    tags = sys.argv[1].split(",")
    steps = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 15
    from tests.util import poison_empty

    ctx = poison_empty() if os.environ.get("PWG_POISON_EMPTY") else contextlib.nullcontext()
    loops = int(sys.argv[sys.argv.index("--loop") + 1]) if "--loop" in sys.argv else 1
    with ctx:
        for it in range(loops):
            for tag in tags:
                run_one(tag, steps, "--no-graph" not in sys.argv, "--sync" in sys.argv, "--quiet" in sys.argv)
                if "--gc" in sys.argv:
                    import gc

                    gc.collect()
                    torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
