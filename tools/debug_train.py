import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_hifigan_train_gpu import build_trainer
from tests.util import load_golden
gold = load_golden("hifigan_v1_train")
batch, n_steps, seed = (int(v) for v in gold["meta"])
dev = torch.device("cuda:0")
tr, batches, model, opt = build_trainer(dev, seed, float(gold["g_scale"]), batch, n_steps)
tr.tqdm = None
prev = {}
for i in range(n_steps):
    tr._train_step(batches[i]); tr._flush_pending()
    cur = dict(tr.total_train_loss)
    for k, v in cur.items():
        want = float(gold[f"step{i}/{k}"]); got = v - prev.get(k, 0.0)
        print(f"step{i} {k:36s} got {got:.6f} want {want:.6f} rel {abs(got-want)/abs(want):.2e}")
    prev = cur
    if i == 0:
        for key in ("generator", "discriminator"):
            names = {p: n for n, p in model[key].named_parameters()}
            norms = {names[p]: float(s["exp_avg"].double().norm()) for p, s in opt[key].state.items()}
            gn = [str(n) for n in gold[f"gradnorm_names/{key}"]]
            got = np.array([norms[n] for n in gn]); want = gold[f"gradnorm/{key}"]
            rel = np.abs(got - want) / (np.abs(want) + 1e-12)
            idx = np.argsort(-rel)[:6]
            print(key, "worst gradnorm rel:", [(gn[j], f"{rel[j]:.1e}", f"{want[j]:.2e}") for j in idx])
for key, tag in (("generator", "g"), ("discriminator", "d")):
    sd = model[key].state_dict(); names = [str(n) for n in gold[f"final_names/{tag}"]]
    got = np.array([float(sd[n].double().sum()) for n in names]); want = gold[f"final_sum/{tag}"]
    err = np.abs(got - want); idx = np.argsort(-err)[:8]
    print(key, "worst final-sum abs diff:", [(names[j], f"{err[j]:.2e}", f"{want[j]:.3e}", sd[names[j]].numel()) for j in idx])
