#!/usr/bin/env python3
"""Which ATen kernels are still launched by one eager C3 training step (next to the library's own kernels)?
torch.profiler over one step after warm-up; prints op name, input shapes, count, device time.  GPU only."""
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from parallelwavegan_amd.bin.train import Trainer  # noqa: E402
from parallelwavegan_amd.utils import build_from_config  # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "c3"
    dev = torch.device("cuda:0")
    conf = bench.load_conf(bench.TRAIN_CONFIGS[tag])
    torch.manual_seed(4321)
    model, criterion, opt, sched = build_from_config(conf, dev)
    conf.update(generator_train_start_steps=0, discriminator_train_start_steps=0, train_max_steps=10 ** 9,
                save_interval_steps=10 ** 9, eval_interval_steps=10 ** 9, log_interval_steps=10 ** 9,
                distributed=False, rank=0, outdir=tempfile.mkdtemp(), progress=False, use_hip_graph=False)
    batch = bench.synthetic_batch(conf, conf["batch_size"], dev, 0)
    tr = Trainer(steps=1, epochs=0, data_loader={"train": [batch], "dev": [batch]}, sampler={"train": None, "dev": None},
                 model=model, criterion=criterion, optimizer=opt, scheduler=sched, config=conf, device=dev)
    tr.tqdm = None
    for _ in range(3):
        tr._train_step(batch)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile

    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        tr._train_step(batch)
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_input_shape=True):
        dt = getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0))
        if dt > 0 and e.key.startswith("aten::"):
            rows.append((dt, e.count, e.key, str(e.input_shapes)[:110]))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print(f"{tag}: {sum(r[1] for r in rows)} ATen ops with device time, {tot / 1e3:.3f} ms device time in one step")
    for dt, n, k, sh in rows[:45]:
        print(f"{dt:9.1f} us  n={n:4d}  {k:28s} {sh}")


if __name__ == "__main__":
    main()
