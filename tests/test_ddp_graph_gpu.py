"""GPU: data-parallel training on two ranks (gloo collectives, both ranks on cuda:0 -- the single-GPU
test box; RCCL itself is the same torch.distributed call).  The segmented hipGraph mode (graphs cut at
the gradient-exchange points, collectives eager in between) must follow the eager bucketed-hook mode
step for step, and both ranks must hold identical parameters afterwards."""
import os
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N_STEPS = 6


def _worker(rank, world, port, use_graph, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from tests.golden import synth
    from tests.test_hifigan_train_gpu import build_trainer

    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    tr, _, model, opt = build_trainer(dev, 41, 1.25, 2, N_STEPS, distributed=True, use_hip_graph=use_graph,
                                      graph_warmup_steps=2, rank=rank)
    tr.tqdm = None
    # every rank trains on its own shard
    c = synth.synth_input("c", (2, 80, 32), seed=100 + rank)
    y = 0.5 * synth.synth_input("y", (2, 1, 8192), seed=100 + rank)
    log = []
    for _ in range(N_STEPS):
        tr._train_step(((c,), y))
        tr._flush_pending()
        log.append(dict(tr.total_train_loss))
    if use_graph:
        assert len(tr._graphs) == 1
        (entry,) = tr._graphs.values()
        # G backward | exchange | G update + D forward + D backward group 0 | exchange 0 | group 1 | ... | D update
        n_groups = len(tr.reducers["discriminator"].groups)
        assert n_groups == 3
        assert [k for _, k in entry["segments"]] == (
            [("generator", None)] + [("discriminator", gi) for gi in range(n_groups)] + [None])
    # weight-gradient kernels write straight into the bucket slots (ops.claim_grad_slot): hook copies per step
    copies = {k: r.copies / N_STEPS for k, r in tr.reducers.items()} if not use_graph else None
    n_params = {k: len(r.params) for k, r in tr.reducers.items()}
    sums = {k: float(sum(p.double().sum().item() for p in model[k].parameters())) for k in model}
    absd = {k: float(sum(p.double().abs().sum().item() for p in model[k].parameters())) for k in model}
    torch.save(dict(log=log, sums=sums, absd=absd, copies=copies, n_params=n_params),
               os.path.join(out_dir, f"r{rank}_g{int(use_graph)}.pt"))
    dist.destroy_process_group()


def test_segmented_graph_ddp_matches_eager_ddp():
    out = tempfile.mkdtemp()
    for i, use_graph in enumerate((False, True)):
        mp.spawn(_worker, args=(2, 29620 + i, use_graph, out), nprocs=2, join=True)
    res = {(r, g): torch.load(os.path.join(out, f"r{r}_g{g}.pt")) for r in (0, 1) for g in (0, 1)}
    # every generator gradient arrives in its bucket slot without a copy.  The discriminator phase differentiates D(y) and
    # D(G(c)) in one pass: the first contribution of a parameter is written into the slot, the second added into it by
    # its own node (ops.claim_grad_slots), autograd sees one gradient per parameter.  Only the spectral-norm weights of
    # the first scale discriminator (their gradient is produced by SpectralNormFn, not by a convolution) are still copied.
    cp, npar = res[(0, 0)]["copies"], res[(0, 0)]["n_params"]
    print(f"[ddp] hook copies per step: {cp} of {npar} parameters")
    assert cp["generator"] == 0 and cp["discriminator"] <= 8
    for g in (0, 1):  # replicas stay in lock-step: identical parameters on both ranks
        for k in ("generator", "discriminator"):
            assert res[(0, g)]["sums"][k] == res[(1, g)]["sums"][k], (g, k)
    for r in (0, 1):  # graph mode follows eager mode
        for i, (a, b) in enumerate(zip(res[(r, 0)]["log"], res[(r, 1)]["log"])):
            for k in a:
                assert abs(a[k] - b[k]) <= 2e-4 * max(abs(a[k]), 1e-3), (r, i, k, a[k], b[k])
        for k in ("generator", "discriminator"):
            d = abs(res[(r, 0)]["sums"][k] - res[(r, 1)]["sums"][k])
            assert d <= 1e-6 * res[(r, 0)]["absd"][k], (r, k, d)


def _rccl_worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                      PWG_FORCE_COLLECTIVES="1")
    import torch.distributed as dist

    from tests.golden import synth
    from tests.test_hifigan_train_gpu import build_trainer

    dist.init_process_group("nccl", rank=0, world_size=1)  # "nccl" is RCCL on ROCm
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    c = synth.synth_input("c", (2, 80, 32), seed=100)
    y = 0.5 * synth.synth_input("y", (2, 1, 8192), seed=100)
    logs = {}
    for distributed in (False, True):
        tr, _, model, opt = build_trainer(dev, 41, 1.25, 2, N_STEPS, distributed=distributed, use_hip_graph=True,
                                          graph_warmup_steps=2, rank=0)
        tr.tqdm = None
        for _ in range(N_STEPS):
            tr._train_step(((c,), y))
        tr._flush_pending()
        torch.cuda.synchronize()
        if distributed:
            (entry,) = tr._graphs.values()
            assert len(entry["segments"]) == 5  # G | D group 0..2 | D update: RCCL calls between the replays
            assert all(r.force for r in tr.reducers.values())
        logs[distributed] = dict(tr.total_train_loss)
    torch.save(logs, os.path.join(out_dir, "rccl.pt"))
    dist.destroy_process_group()


def test_rccl_collectives_between_graph_segments_world_of_one():
    """The data-parallel hipGraph step on RCCL itself (process group of ONE rank, every bucket all-reduce
    really issued: PWG_FORCE_COLLECTIVES): communicator init, the watchdog thread next to a thread-local
    stream capture, async all-reduces on RCCL's stream between graph replays.  A sum over one rank is the
    identity, so the losses must follow the non-distributed trainer."""
    out = tempfile.mkdtemp()
    mp.spawn(_rccl_worker, args=(29640, out), nprocs=1, join=True)
    logs = torch.load(os.path.join(out, "rccl.pt"))
    for k, v in logs[False].items():
        assert abs(v - logs[True][k]) <= 2e-4 * max(abs(v), 1e-3), (k, v, logs[True][k])
