"""Fork/join helper: run independent branches (the 8 sub-discriminators of HiFi-GAN's MSD+MPD, the
3 MRF residual blocks of a generator stage) on separate HIP streams so that, inside a captured
hipGraph, they become parallel branches of the DAG and small kernels of different branches share
the 256 CUs.  Autograd replays each backward node on the stream of its forward, so the backward
pass is forked the same way.
"""
import torch

_POOL = {}


def _streams(device, n):
    key = (device.index if device.index is not None else torch.cuda.current_device())
    pool = _POOL.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


def _record(obj, stream):
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            _record(o, stream)


def run_branches(branches, device, enabled=True):
    """branches: list of zero-argument callables -> list of their results."""
    if not enabled or len(branches) < 2 or device.type != "cuda":
        return [fn() for fn in branches]
    cur = torch.cuda.current_stream(device)
    side = _streams(device, len(branches))
    fork = torch.cuda.Event()
    fork.record(cur)
    outs = []
    for fn, s in zip(branches, side):
        s.wait_event(fork)
        with torch.cuda.stream(s):
            outs.append(fn())
    for o, s in zip(outs, side):
        cur.wait_stream(s)
        _record(o, cur)  # produced on a side stream, consumed on the caller's stream
    return outs
