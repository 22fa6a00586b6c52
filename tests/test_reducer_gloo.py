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


def _worker8(rank, world, port, out):
    """World of 8 (the node size the multi-GPU bench runs at): ranks execute the step in DIFFERENT ways -- even ranks
    eagerly with overlapped hook launches, odd ranks group by group in deferred (hipGraph) mode; rank 3's last layer
    gets no gradient at all, rank 5 produces its gradients in reversed group order -- and must still issue the same
    collective sequence (no hang) and end with the average of the per-rank gradients."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from parallelwavegan_amd.distributed import GradReducer, partition_modules

    torch.manual_seed(0)
    subs = [torch.nn.Sequential(torch.nn.Linear(6, 12), torch.nn.Tanh(), torch.nn.Linear(12, 2)) for _ in range(3)]
    model = torch.nn.ModuleList(subs)
    params = list(model.parameters())
    groups = partition_modules(list(model), 3)
    red = GradReducer(params, bucket_bytes=200, groups=groups)
    red.broadcast_parameters(params)
    n_buckets = len(red.buckets)
    assert [b.group for b in red.buckets] == sorted(b.group for b in red.buckets) and n_buckets >= 3
    g = torch.Generator().manual_seed(7)
    x_all = torch.randn(world, 4, 6, generator=g)

    def loss_of(r, skip_last):
        terms = [m(x_all[r]).pow(2).mean() for i, m in enumerate(model) if not (skip_last and i == 2)]
        return sum(terms)

    # reference: mean over ranks of each rank's own gradient (rank 3 contributes zero for sub-network 2)
    want = [torch.zeros_like(p) for p in params]
    for r in range(world):
        gr = torch.autograd.grad(loss_of(r, r == 3), params, allow_unused=True)
        for w, t in zip(want, gr):
            if t is not None:
                w += t / world
    errs = []
    for step in range(2):  # twice: the second step must not see the first step's gradients in un-refilled slots
        loss = loss_of(rank, rank == 3)
        if rank % 2 == 0:
            red.prepare()
            loss.backward()
            red.zero_missing()
            red.finish()
        else:
            red.defer = True
            red.prepare()
            order = list(range(len(red.groups)))
            if rank == 5:
                order.reverse()  # gradients become available in another order; the collectives must not
            for n, gi in enumerate(order):
                inputs = [p for p in red.groups[gi] if p.requires_grad]
                torch.autograd.backward(loss, inputs=inputs, retain_graph=n < len(order) - 1)
                red.zero_missing(gi)
            red.finish()
            red.defer = False
            red.begin_replay()
            for gi in range(len(red.groups)):
                red.exchange_group(gi)
            red.wait_all()
        scale = 1.0 / world
        errs.append(max((red.flat_grads[p] * scale - w).abs().max().item() for p, w in zip(params, want)))
    out[rank] = (max(errs), n_buckets, red.zero_fills)
    red.remove()
    dist.destroy_process_group()


def test_grad_reducer_world_size_8_mixed_execution_orders():
    mp.set_start_method("spawn", force=True)
    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    world = 8
    procs = [mp.Process(target=_worker8, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0, "a rank hung or died: collective sequences differ between ranks"
    for r in range(world):
        err, nb, zero_fills = out[r]
        assert err <= 1e-6, (r, err)
        # only rank 3 has empty slots (the 4 parameters of sub-network 2, two steps); nobody zero-fills whole buckets
        assert zero_fills == (8 if r == 3 else 0), (r, zero_fills)
