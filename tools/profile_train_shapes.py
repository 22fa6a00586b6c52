"""Per-problem-shape kernel times of one eager training step (PWG_PROF_SHAPES=1): which launches of the
C3 / C2 / C4 step cost what.  usage: PWG_PROF_SHAPES=1 python tools/profile_train_shapes.py [c3|c2|c4] [top]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PWG_PROF_SHAPES", "1")
os.environ.setdefault("PWG_WAVENET_WGRAD_STREAM", "0")  # serial launches: every kernel is timed alone
import torch  # noqa: E402

import bench  # noqa: E402
from parallelwavegan_amd import ops  # noqa: E402
from parallelwavegan_amd.bin.train import Trainer  # noqa: E402
from parallelwavegan_amd.utils import build_from_config  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "c3"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda:0")
conf = bench.load_conf(bench.TRAIN_CONFIGS[tag])
torch.manual_seed(1)
model, criterion, opt, sched = build_from_config(conf, dev)
conf.update(generator_train_start_steps=0, discriminator_train_start_steps=0, train_max_steps=10 ** 9,
            save_interval_steps=10 ** 9, eval_interval_steps=10 ** 9, log_interval_steps=10 ** 9, distributed=False,
            rank=0, outdir=tempfile.mkdtemp(), progress=False, use_hip_graph=False)
batch = bench.synthetic_batch(conf, conf["batch_size"], dev, 0)
tr = Trainer(steps=1, epochs=0, data_loader={"train": [batch], "dev": [batch]}, sampler={"train": None, "dev": None},
             model=model, criterion=criterion, optimizer=opt, scheduler=sched, config=conf, device=dev)
tr.tqdm = None
for _ in range(3):
    tr._train_step(batch)
torch.cuda.synchronize()
with ops.profile() as prof:
    tr._train_step(batch)
rows = sorted(prof.results.items(), key=lambda kv: -kv[1]["ms"])
tot = sum(v["ms"] for _, v in rows)
print(f"{tag}: {len(rows)} distinct (kernel, shape) rows, {sum(v['launches'] for _, v in rows)} launches, {tot:.2f} ms of kernel time")
acc = 0.0
for name, v in rows[:top]:
    acc += v["ms"]
    tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["flops"] else 0.0
    print(f"{v['ms']:7.3f} ms {100 * v['ms'] / tot:5.1f}% cum {100 * acc / tot:5.1f}%  n={v['launches']:3d}  {tf:6.1f} TF  "
          f"{v['bytes'] / (v['ms'] * 1e-3) / 1e9:7.0f} GB/s  {name}")
