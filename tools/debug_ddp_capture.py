"""Debug aid: run the data-parallel trainer (2 ranks sharing cuda:0, gloo) in eager / hipGraph mode with
1 or 4 exchange groups and print the per-step losses of rank 0, to locate a divergence."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def worker(rank, world, port, groups, graph, steps):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from tests.golden import synth
    from tests.test_hifigan_train_gpu import build_trainer

    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    tr, _, model, opt = build_trainer(dev, 41, 1.25, 2, steps, distributed=True, use_hip_graph=graph != 2,
                                      graph_warmup_steps=2 if graph == 1 else 10 ** 6, rank=rank,
                                      ddp_grad_groups=groups)
    tr.tqdm = None
    c = synth.synth_input("c", (2, 80, 32), seed=100 + rank)
    y = 0.5 * synth.synth_input("y", (2, 1, 8192), seed=100 + rank)
    prev = {}
    for i in range(steps):
        tr._train_step(((c,), y))
        tr._flush_pending()
        cur = dict(tr.total_train_loss)
        if rank == 0:
            print(f"groups={groups} graph={graph} step {i}: " +
                  " ".join(f"{k.split('/')[-1][:8]}={cur[k] - prev.get(k, 0.0):.6f}" for k in sorted(cur)), flush=True)
        prev = cur
    s = sum(p.double().sum().item() for p in model["discriminator"].parameters())
    print(f"groups={groups} graph={graph} rank {rank}: D param sum {s:.9f}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    port = 29700
    for w in (sys.argv[1:] or ["1:0", "4:0", "1:1", "4:1"]):
        g, b = w.split(":")
        port += 1
        mp.spawn(worker, args=(2, port, int(g), int(b), 6), nprocs=2, join=True)
