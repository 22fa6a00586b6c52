"""One small training step (generator AND discriminator phase, eager) of EVERY training recipe of the reference
(egs/*/*/conf/*.yaml, staged unedited by oracle/make_ref.py as oracle/_ref/egs/...): what a user of the reference who
switches engines would run first.  Per recipe: build_from_config on the GPU, a synthetic batch of 2 segments at the
recipe's own segment length, two Trainer steps, finite losses.  Prints one line per recipe and a summary.
usage: run_all_recipes.py [substring[,substring...]]"""
import glob
import os
import sys
import tempfile
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parallelwavegan_amd.bin.train import Trainer  # noqa: E402
from parallelwavegan_amd.utils import build_from_config  # noqa: E402


def synthetic_batch(conf, b, dev):
    """What the reference's Collater hands the Trainer (bin/train.py:640-812): the segment is batch_max_steps rounded
    down to whole frames; (z, c) for Parallel WaveGAN, (c,) otherwise, (c, f0, excitation) for UHiFiGAN."""
    g = torch.Generator().manual_seed(5)
    hop = conf["hop_size"]
    frames = conf["batch_max_steps"] // hop
    t = frames * hop
    gp = conf["generator_params"]
    acw = gp.get("aux_context_window", 0)
    gtype = conf.get("generator_type", "ParallelWaveGANGenerator")
    c = torch.randn(b, gp.get("aux_channels", gp.get("in_channels", conf["num_mels"])), frames + 2 * acw, generator=g).to(dev)
    y = (0.3 * torch.randn(b, 1, t, generator=g)).to(dev)
    if gtype == "ParallelWaveGANGenerator":
        return ((torch.randn(b, 1, t, generator=g).to(dev), c), y)
    if gtype == "UHiFiGANGenerator":
        f0 = torch.rand(b, 1, frames, generator=g).to(dev)
        return ((c, f0, torch.randn(b, 1, t, generator=g).to(dev)), y)
    return ((c,), y)


def run_one(path, dev):
    with open(path) as f:
        conf = yaml.load(f, Loader=yaml.Loader)
    try:
        model, criterion, opt, sched = build_from_config(conf, dev)
    except NotImplementedError as e:
        return "out-of-scope", str(e)[:110]
    conf.update(generator_train_start_steps=0, discriminator_train_start_steps=0, train_max_steps=10 ** 9,
                save_interval_steps=10 ** 9, eval_interval_steps=10 ** 9, log_interval_steps=10 ** 9, distributed=False,
                rank=0, outdir=tempfile.mkdtemp(), progress=False, record_loss_history=True)
    batch = synthetic_batch(conf, 2, dev)
    tr = Trainer(steps=1, epochs=0, data_loader={"train": [batch], "dev": [batch]}, sampler={"train": None, "dev": None},
                 model=model, criterion=criterion, optimizer=opt, scheduler=sched, config=conf, device=dev)
    tr.tqdm = None
    for _ in range(2):
        tr._train_step(batch)
    torch.cuda.synchronize()
    hist = tr.loss_history()
    bad = [k for _, losses in hist for k, v in losses.items() if not (v == v and abs(v) != float("inf"))]
    del tr, model, criterion, opt, sched
    torch.cuda.empty_cache()
    return ("non-finite", ",".join(sorted(set(bad)))) if bad else ("ok", f"{len(hist[-1][1])} losses")


def main():
    dev = torch.device("cuda:0")
    files = sorted(glob.glob(os.path.join(ROOT, "oracle", "_ref", "egs", "*", "*", "conf", "*.yaml")))
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    res = {}
    for p in files:
        name = os.path.relpath(p, os.path.join(ROOT, "oracle", "_ref", "egs"))
        if not any(f in name for f in flt.split(",")):
            continue
        t0 = time.time()
        try:
            res[name] = run_one(p, dev)
        except Exception as e:  # noqa: BLE001
            res[name] = ("FAILED", f"{type(e).__name__}: {str(e)[:160]}")
            torch.cuda.empty_cache()
        print(f"{name:58s} {res[name][0]:13s} {time.time() - t0:5.1f}s  {res[name][1]}", flush=True)
    kinds = {}
    for k, (st, _) in res.items():
        kinds.setdefault(st, []).append(k)
    print("SUMMARY", {k: len(v) for k, v in kinds.items()}, "of", len(res))
    return res


if __name__ == "__main__":
    main()
