"""Where is the C3 step's critical path?  The captured step with sub-discriminators removed (PWG_ABL = comma list of:
msd0 msd1 msd2 mpd0..mpd4 | msd | mpd | nofm).  usage: PWG_ABL=msd0 python tools/ablate_c3.py [c3|c5] [steps]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from parallelwavegan_amd.bin.train import Trainer  # noqa: E402
from parallelwavegan_amd.utils import build_from_config  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "c3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
abl = [a for a in os.environ.get("PWG_ABL", "").split(",") if a]
dev = torch.device("cuda:0")
conf = bench.load_conf(bench.TRAIN_CONFIGS[tag])
torch.manual_seed(4321)
model, criterion, opt, sched = build_from_config(conf, dev)
d = model["discriminator"]
keep_msd = [i for i in range(len(d.msd.discriminators)) if f"msd{i}" not in abl and "msd" not in abl]
keep_mpd = [i for i in range(len(d.mpd.discriminators)) if f"mpd{i}" not in abl and "mpd" not in abl]
d.msd.discriminators = torch.nn.ModuleList([d.msd.discriminators[i] for i in keep_msd])
d.mpd.discriminators = torch.nn.ModuleList([d.mpd.discriminators[i] for i in keep_mpd])
if "nofm" in abl:
    conf["use_feat_match_loss"] = False
conf.update(generator_train_start_steps=0, discriminator_train_start_steps=0, train_max_steps=10 ** 9,
            save_interval_steps=10 ** 9, eval_interval_steps=10 ** 9, log_interval_steps=10 ** 9, distributed=False,
            rank=0, outdir=tempfile.mkdtemp(), progress=False, use_hip_graph=True, graph_warmup_steps=2)
batch = bench.synthetic_batch(conf, conf["batch_size"], dev, 0)
tr = Trainer(steps=1, epochs=0, data_loader={"train": [batch], "dev": [batch]}, sampler={"train": None, "dev": None},
             model=model, criterion=criterion, optimizer=opt, scheduler=sched, config=conf, device=dev)
tr.tqdm = None
for i in range(n):
    if i == n - 6:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    tr._train_step(batch)
torch.cuda.synchronize()
print(f"{tag} without [{','.join(abl) or '-'}] (msd {keep_msd}, mpd {keep_mpd}): {(time.perf_counter() - t0) / 6 * 1e3:.2f} ms per step")
