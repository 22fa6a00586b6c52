"""Fork/join helper: run independent branches (the 8 sub-discriminators of HiFi-GAN's MSD+MPD, the
3 MRF residual blocks of a generator stage) on separate HIP streams so that, inside a captured
hipGraph, they become parallel branches of the DAG and small kernels of different branches share
the 256 CUs.  Autograd replays each backward node on the stream of its forward, so the backward
pass is forked the same way.
"""
import os

import torch

_POOL = {}


# At most this many side streams: more branches than that share streams round-robin (branches on one stream run in
# program order).  The device exposes 4 hardware queues to a process by default; measured on the HiFi-GAN V1 step
# (profiles/r04_queues_and_streams.txt): 8 or 16 queues make the captured step 35 % SLOWER (the
# concurrent MFMA kernels evict each other's tiles), one stream is 22 % slower than the default.
MAX_SIDE_STREAMS = int(os.environ.get("PWG_MAX_SIDE_STREAMS", "8"))


# Fork only while the stream is being captured into a hipGraph (default), or also in eager execution
# (PWG_EAGER_BRANCH_STREAMS=1).  Inside a capture the side streams are bookkeeping: they become dependency edges of
# the graph and the replay involves no stream scheduling at all; the eager fork bought nothing a warm-up step needs,
# so eager steps (graph warm-up, the data-parallel fallback) run their branches one after the other on the caller's
# stream.
#
# Round 5 found the eager fork producing NaN about once in four fresh processes under NaN-poisoned allocations; round 6
# bisected it (profiles/r06_eager_nan_bisect.txt) to a missing ``record_stream`` on the branch INPUTS: a tensor
# allocated on the caller's stream and read on a side stream -- in the forward and again, as a saved tensor, by the
# branch's backward nodes, which autograd replays on the side stream -- went back to the caller's stream's pool the
# moment its last reference died, while the side stream's reader was still queued.  The pooled input of HiFi-GAN's
# second scale discriminator (16 x 4097 floats) was then handed to the equally-sized gradient of the pooling's
# backward on the caller's stream, whose (poison) fill overtook the first layer's weight-gradient kernel.  The outputs
# of a branch were always recorded; now the inputs are too (``inputs=`` below), which also pins them until the end of a
# capture (the caching allocator defers the reuse of a block with recorded streams to the end of the capture), so the
# captured graph cannot contain that write-after-read pair without an edge either.
# A second finding of the same hunt, in the data-parallel path: a LATER contribution to a gradient slot was added on the
# caller's stream when the stateful sub-discriminator was re-evaluated un-forked, unordered against the forked pass's first
# write (ops.slot_add; the eager two-rank run differed from its segmented-graph twin in 3 of 12 runs before, 0 of 12 after).
EAGER_FORK = os.environ.get("PWG_EAGER_BRANCH_STREAMS", "0") == "1"


def fork_now(device=None):
    """Should independent branches be forked onto side streams right now?"""
    return EAGER_FORK or (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing())


def reserve(device, n=None):
    """Create the side streams ahead of a capture (stream creation is not a capturable operation)."""
    return _streams(device, MAX_SIDE_STREAMS if n is None else n)


def _streams(device, n):
    key = (device.index if device.index is not None else torch.cuda.current_device())
    pool = _POOL.setdefault(key, [])
    m = max(1, min(n, MAX_SIDE_STREAMS))
    while len(pool) < m:
        pool.append(torch.cuda.Stream(device=device))
    return [pool[i % m] for i in range(n)]


def _record(obj, stream):
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            _record(o, stream)


def run_branches(branches, device, enabled=True, inputs=None):
    """branches: list of zero-argument callables -> list of their results.  ``inputs``: the tensors (nested lists /
    tuples allowed) that the branches read but did not allocate -- they are recorded on every side stream so that the
    caching allocator does not hand their memory out while a side stream still reads it (forward or backward)."""
    if not enabled or len(branches) < 2 or device.type != "cuda" or not fork_now():
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
    for s in dict.fromkeys(side):
        cur.wait_stream(s)
        _record(inputs, s)  # allocated on the caller's stream (or further up), read on the side stream
    for o in outs:
        _record(o, cur)  # produced on a side stream, consumed on the caller's stream
    return outs


def run_branches_chained(branches, device, inputs=None):
    """Like :func:`run_branches`, for branches whose results are summed in order (``cs += block(c)``): branch j is
    called as ``fn(join)`` where ``join()`` -- to be called right before the branch's LAST kernel -- makes the
    branch's stream wait for branch j-1 and returns that branch's result (None for j = 0), so the running sum
    rides in the last kernel's epilogue instead of a separate combine launch.  Everything before the last kernel
    of every branch still runs concurrently.  Returns the last branch's result on the caller's stream."""
    cur = torch.cuda.current_stream(device)
    side = _streams(device, len(branches))
    if len(set(side)) < len(branches):
        # fewer side streams than branches (PWG_MAX_SIDE_STREAMS < 3): two chained branches would share a stream and the
        # capture of that pattern crashes inside hipGraphInstantiate -- run the chain in order on the caller's stream
        res = None
        for fn in branches:
            res = fn(lambda r=res: r)
        return res
    fork = torch.cuda.Event()
    fork.record(cur)
    prev = None  # (completion event, result) of the previous branch
    for fn, s in zip(branches, side):
        s.wait_event(fork)
        with torch.cuda.stream(s):
            def join(prev=prev, s=s):
                if prev is None:
                    return None
                s.wait_event(prev[0])
                _record(prev[1], s)  # produced on the previous branch's stream, consumed on this one
                return prev[1]

            out = fn(join)
            done = torch.cuda.Event()
            done.record(s)
            prev = (done, out)
    for s in dict.fromkeys(side):
        cur.wait_stream(s)
        _record(inputs, s)  # (see run_branches)
    _record(prev[1], cur)
    return prev[1]
