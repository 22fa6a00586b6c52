"""CPU, world_size 2 (gloo): the data-parallel gradient reducer averages exactly like single-process
training on the concatenated batch (the semantics apex DDP gives the reference)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from parallelwavegan_amd.distributed import GradReducer

    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))
    if rank == 1:  # ranks start different: broadcast must make them equal
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    params = list(model.parameters())
    red = GradReducer(params, bucket_bytes=100)  # two small buckets
    red.broadcast_parameters(params)
    g = torch.Generator().manual_seed(123)
    x_all = torch.randn(2, 5, 8, generator=g)  # per-rank shards of the global batch
    y_all = torch.randn(2, 5, 3, generator=g)
    red.prepare()
    loss = torch.nn.functional.mse_loss(model(x_all[rank]), y_all[rank])
    loss.backward()
    scale = red.finish()
    grads = [red.flat_grads[p] * scale for p in params]
    # single-process reference on the concatenated batch (mean over both shards)
    ref = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))
    ref.load_state_dict(model.state_dict())
    ref_loss = 0.5 * (torch.nn.functional.mse_loss(ref(x_all[0]), y_all[0]) +
                      torch.nn.functional.mse_loss(ref(x_all[1]), y_all[1]))
    ref_loss.backward()
    err = max((a - b.grad).abs().max().item() for a, b in zip(grads, ref.parameters()))
    none_left = all(p.grad is None for p in params)
    # a second step where one parameter gets no gradient: its slot must come back as exactly zero
    red.prepare()
    params[-1].requires_grad_(False)
    (model(x_all[rank]).sum() * 0 + params[0].sum()).backward()
    red.finish()
    zero_ok = float(red.flat_grads[params[-1]].abs().max()) == 0.0
    params[-1].requires_grad_(True)
    # deferred mode (what a hipGraph capture / replay uses): hooks only fill the buckets, finish() launches
    # nothing, exchange_all() performs the collectives -> same averaged gradients as the hook mode
    red.defer = True
    red.prepare()
    torch.nn.functional.mse_loss(model(x_all[rank]), y_all[rank]).backward()
    red.finish()
    local_only = max((red.flat_grads[p] * scale - g0).abs().max().item() for p, g0 in zip(params, grads))
    red.defer = False
    red.exchange_all()
    err_deferred = max((red.flat_grads[p] * scale - g0).abs().max().item() for p, g0 in zip(params, grads))
    red.remove()
    # exchange groups (hipGraph mode: one backward segment per group, group k's all-reduce overlaps group
    # k+1's backward).  The two ranks deliberately run the step DIFFERENTLY -- rank 0 eagerly with hooks,
    # rank 1 group by group in deferred mode with restricted backward calls -- and must still issue the
    # same collective sequence (bucket order) and end with the same averaged gradients.
    from parallelwavegan_amd.distributed import partition_modules

    groups = partition_modules([model[2], model[0]], 2)
    assert [len(g) for g in groups] == [2, 2] and groups[0][0] is model[2].bias
    red2 = GradReducer(params, bucket_bytes=100, groups=groups)
    assert [b.group for b in red2.buckets] == sorted(b.group for b in red2.buckets)
    loss = torch.nn.functional.mse_loss(model(x_all[rank]), y_all[rank])
    if rank == 0:
        red2.prepare()
        loss.backward()
        red2.finish()
    else:
        red2.defer = True
        red2.prepare()
        for gi, gp in enumerate(red2.groups):
            torch.autograd.backward(loss, inputs=gp, retain_graph=gi < len(red2.groups) - 1)
        red2.finish()  # capture-pass semantics: nothing exchanged yet
        red2.begin_replay()
        for gi in range(len(red2.groups)):
            red2.exchange_group(gi)
        red2.wait_all()
    err_groups = max((red2.flat_grads[p] * scale - g0).abs().max().item() for p, g0 in zip(params, grads))
    out[rank] = (err, none_left, zero_ok, len(red.buckets), local_only, err_deferred, err_groups)
    dist.destroy_process_group()


def test_grad_reducer_world_size_2():
    mp.set_start_method("spawn", force=True)
    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    procs = [mp.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in range(2):
        err, none_left, zero_ok, nb, local_only, err_deferred, err_groups = out[r]
        assert err_groups <= 1e-6, err_groups
        assert err <= 1e-6, err
        assert none_left and zero_ok and nb >= 2
        assert local_only > 1e-4        # before exchange_all the buckets hold this rank's gradients only
        assert err_deferred <= 1e-6, err_deferred
