import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from tests.test_train_full_shape_gpu import _build, load_golden
from tests.util import poison_empty, poison_lds
dev = torch.device("cuda:0")
tag = sys.argv[1]
gold = load_golden(f"{tag}_train_full")
for rep in range(int(sys.argv[2])):
    hist = {}
    for use_graph in (False, True):
        with poison_lds(), poison_empty():
            tr, batch, model, opt = _build(tag, gold, dev, use_hip_graph=use_graph, graph_warmup_steps=2)
            for _ in range(6):
                tr._train_step(batch)
            torch.cuda.synchronize()
            hist[use_graph] = tr.loss_history()
        del tr, model, opt
        torch.cuda.empty_cache()
    worst = []
    for i, ((sa, a), (sb, b)) in enumerate(zip(hist[False], hist[True])):
        k = max(a, key=lambda k: abs(a[k] - b[k]) / max(abs(a[k]), 1e-3))
        worst.append(f"{i}:{k.split('/')[-1]}={abs(a[k] - b[k]) / max(abs(a[k]), 1e-3):.1e}")
    print(rep, " ".join(worst), flush=True)
