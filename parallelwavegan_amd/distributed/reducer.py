"""Data-parallel gradient exchange over RCCL (xGMI), one process per GPU.

Replaces ``apex.parallel.DistributedDataParallel`` at
/root/reference/parallel_wavegan/bin/train.py:1494-1503.  Parameters are assigned, in reverse
registration order (the order backward produces their gradients), to a few large flat fp32
buckets.  A post-accumulate hook copies each gradient into its bucket slot as soon as autograd
has produced it; when a bucket is complete its all-reduce (sum) is launched asynchronously, so
the exchange of the discriminator's 283 MB overlaps the rest of its backward pass.  The fused
optimizers then read the reduced gradients straight from the buckets (``flat_grads``) with the
1/world_size averaging folded into the update kernel (``grad_scale``) -- no copy back.

Collectives are always issued in bucket-index order (a complete bucket waits for its
predecessors), whatever order the hooks fire in: every rank issues the same sequence even when
the ranks execute the step differently (eager vs hipGraph replay, other stream interleaving).

``groups``: optional partition of the parameters into *exchange groups*, in the order in which
their gradients become available.  Buckets never straddle a group, so a caller that produces the
gradients group by group -- the trainer's hipGraph mode replays the discriminator backward as one
graph per group -- can start group k's all-reduce (:meth:`exchange_group`) while group k+1 is
still being computed.

Buckets are large (default 64 MiB): xGMI is a point-to-point mesh (7 links per GPU), so a
collective is per-link bound and a few big messages beat many small ones.
"""
import os

import torch
import torch.distributed as dist


def partition_modules(modules, n_groups):
    """Exchange groups for :class:`GradReducer` from a list of independent sub-networks: contiguous runs
    of ``modules`` with roughly equal parameter bytes; inside a group the parameters are listed in
    reverse registration order (the order a backward pass produces their gradients)."""
    sizes = [sum(p.numel() for p in m.parameters()) for m in modules]
    n_groups = max(1, min(int(n_groups), len(modules)))
    total, groups, cur, acc = float(sum(sizes)), [], [], 0.0
    for i, (m, sz) in enumerate(zip(modules, sizes)):
        cur.append(m)
        acc += sz
        left_modules, left_groups = len(modules) - i - 1, n_groups - len(groups) - 1
        if left_groups > 0 and (acc >= total * (len(groups) + 1) / n_groups or left_modules <= left_groups):
            groups.append(cur)
            cur = []
    if cur:
        groups.append(cur)
    return [[p for m in reversed(g) for p in reversed(list(m.parameters()))] for g in groups]


def _drop_slots(table, keys, owner_ref):
    for k in keys:
        e = table.get(k)
        if e is not None and e[1] is owner_ref:
            del table[k]


class _Bucket:
    def __init__(self, params, device, group):
        self.params = params
        self.group = group
        self.offsets = {}
        n = 0
        for p in params:
            self.offsets[p] = n
            n += p.numel()
        self.flat = torch.zeros(n, device=device, dtype=torch.float32)
        self.zeroed = False
        self.views = {p: self.flat[o:o + p.numel()].view_as(p) for p, o in self.offsets.items()}
        self.pending = len(params)
        self.work = None
        self.launched = False
        self.events = []  # one per hook copy of the current step (the copies may run on several streams)


class GradReducer:
    def __init__(self, params, bucket_bytes=64 << 20, process_group=None, groups=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        params = [p for p in params]
        self.params = params
        if groups is None:
            groups = [list(reversed(params))]
        else:
            groups = [list(g) for g in groups if len(g)]
            seen = {id(p) for g in groups for p in g}
            assert len(seen) == sum(len(g) for g in groups), "a parameter appears in two exchange groups"
            rest = [p for p in reversed(params) if id(p) not in seen]
            if rest:
                groups.append(rest)
        self.groups = groups
        self.buckets = []
        for gi, gparams in enumerate(groups):
            cur, cur_bytes = [], 0
            for p in gparams:
                cur.append(p)
                cur_bytes += p.numel() * 4
                if cur_bytes >= bucket_bytes:
                    self.buckets.append(_Bucket(cur, p.device, gi))
                    cur, cur_bytes = [], 0
            if cur:
                self.buckets.append(_Bucket(cur, cur[0].device, gi))
        self.bucket_of = {p: b for b in self.buckets for p in b.params}
        self.flat_grads = {p: b.views[p] for b in self.buckets for p in b.params}
        self._handles = [p.register_post_accumulate_grad_hook(self._hook) for p in params]
        self.enabled = True
        # Weight-gradient kernels write into the bucket slots directly (ops.claim_grad_slot) -- apex flattens in place
        # too (reference bin/train.py:1494-1503): the hook then has nothing to copy.  One claim per parameter and
        # backward pass (``epoch``).  PWG_DDP_DIRECT=0: every gradient goes through a hook copy as in round 3.
        self.direct_slots = os.environ.get("PWG_DDP_DIRECT", "1") == "1"
        self.epoch = 0
        self._filled = set()
        self.zero_buckets = os.environ.get("PWG_DDP_ZERO_BUCKETS", "0") == "1"
        self.zero_fills = 0  # slots zeroed because no gradient arrived (tests / bench)
        self.copies = 0  # hook copies since construction (bench / tests: how many gradients did NOT arrive in place)
        from .. import ops

        import weakref

        me = weakref.ref(self)
        keys = []
        for p in params:
            if p.is_cuda:
                ops.GRAD_SLOTS[p.data_ptr()] = [self.flat_grads[p], me, -1]
                keys.append(p.data_ptr())
        # a reducer that is dropped without remove() (a Trainer going out of scope) takes its entries -- and with them
        # the strong references to its buckets -- out of the process-wide table
        weakref.finalize(self, _drop_slots, ops.GRAD_SLOTS, keys, me)
        self.defer = False  # True: hooks only fill the buckets (hipGraph capture / replay); the caller exchanges
        self.skip_comm = False  # measurement aid: run the step without its collectives (bench: exposed time)
        # test aid: issue the collectives even in a world of one (exercises RCCL init, its stream semantics and
        # its interplay with hipGraph capture / replay on a single-GPU box)
        self.force = bool(os.environ.get("PWG_FORCE_COLLECTIVES")) and dist.is_initialized()
        self._next = 0  # index of the next bucket to launch (collectives go out in bucket order)
        self.bytes = sum(b.flat.numel() * 4 for b in self.buckets)

    def broadcast_parameters(self, tensors, src=0):
        """Rank 0's parameters and buffers become everyone's (as DDP does at wrap time)."""
        if self.world > 1:
            for t in tensors:
                dist.broadcast(t.data, src, group=self.group)

    def prepare(self):
        """Call before each backward whose gradients are to be exchanged.  The buckets are NOT zero-filled (283 MB per
        discriminator step until round 4): every gradient that arrives overwrites its slot (the first weight-gradient
        launch of a pass writes it, the hook copy overwrites it), and the slots of parameters that received none are
        zeroed individually before the bucket goes out (:meth:`zero_missing`)."""
        for b in self.buckets:
            b.pending = sum(1 for p in b.params if p.requires_grad)
            b.work = None
            b.launched = False
            b.events = []
            b.zeroed = False
            if self.zero_buckets:  # (PWG_DDP_ZERO_BUCKETS=1: the round-4 behaviour, for A/B measurements)
                b.flat.zero_()
                b.zeroed = True
        self._filled = set()
        self._next = 0
        self.epoch += 1  # every slot may be claimed once in the coming backward pass

    def zero_missing(self, gi=None):
        """Zero the slots of the parameters (of exchange group ``gi``, default all) whose gradient has not arrived:
        they contribute exactly 0 to the sum, as a parameter without a gradient does under DDP.  The trainer calls it
        after the backward pass of a group -- inside the captured graph segment in hipGraph mode, so that a replay
        re-zeroes the same slots --; buckets that complete earlier (overlapped launch from a hook) are handled by
        :meth:`_all_reduce`.  Returns the number of slots zeroed."""
        n = 0
        for b in self.buckets:
            if (gi is None or b.group == gi) and not b.zeroed:
                for p in b.params:
                    if id(p) not in self._filled:
                        b.views[p].zero_()
                        n += 1
                b.zeroed = True
        self.zero_fills += n
        return n

    def begin_replay(self):
        """Host bookkeeping of :meth:`prepare` for a step whose kernels (incl. the bucket zero-fill and
        the hook copies) are replayed from a captured hipGraph."""
        for b in self.buckets:
            b.pending = 0
            b.work = None
            b.launched = False
            b.zeroed = True  # the captured segment re-zeroes the slots that were empty at capture time
        self._next = 0

    def _hook(self, p):
        if not self.enabled:
            return
        b = self.bucket_of[p]
        if p.grad is None:
            # (torch fires the hook also when the incoming gradient is undefined, e.g. the unused residual 1x1 of the
            # last WaveNet layer: the zero-filled slot is that parameter's contribution)
            pass
        elif p.grad.data_ptr() != b.views[p].data_ptr():  # (else: the kernel wrote the slot itself, see direct_slots)
            b.views[p].copy_(p.grad)
            self.copies += 1
            self._filled.add(id(p))
        else:
            self._filled.add(id(p))
        p.grad = None  # the bucket slot now owns this gradient
        if not self.defer and b.flat.is_cuda:
            # autograd runs this hook on the stream of the node that produced the gradient; with the
            # sub-discriminators forked onto side streams one bucket is filled from several streams, and
            # the collective (launched from whichever hook completes the bucket) must wait for all of them
            ev = torch.cuda.Event()
            ev.record()
            b.events.append(ev)
        b.pending -= 1
        if b.pending == 0 and not self.defer:
            self._launch_ready()

    def _all_reduce(self, b):
        b.launched = True
        if b.events:  # first: this stream joins every stream a slot of the bucket was produced on
            cur = torch.cuda.current_stream(b.flat.device)
            for ev in b.events:
                cur.wait_event(ev)
            b.events = []
        if not b.zeroed and not self.defer:
            # slots no gradient arrived for contribute 0 (disjoint from the produced slots; issued after the event
            # waits so that fill and collective are ordered behind every producer of the bucket all the same)
            for p in b.params:
                if id(p) not in self._filled:
                    b.views[p].zero_()
                    self.zero_fills += 1
            b.zeroed = True
        if (self.world > 1 or self.force) and not self.skip_comm:
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _launch_ready(self):
        """Launch, in bucket order, every complete bucket whose predecessors have been launched."""
        while self._next < len(self.buckets) and self.buckets[self._next].pending <= 0:
            self._all_reduce(self.buckets[self._next])
            self._next += 1

    def exchange_group(self, gi):
        """Start (asynchronously) the all-reduce of every bucket of exchange group ``gi``; groups must
        be exchanged in index order.  hipGraph mode: the buckets were filled by the replayed graph
        segment of this group, whose capture recorded the hook copies but no collective."""
        for i, b in enumerate(self.buckets):
            if b.group == gi and not b.launched:
                assert i == self._next, "exchange groups out of order"
                self._all_reduce(b)
                self._next += 1

    def exchange_all(self):
        """All-reduce every bucket not yet launched and wait."""
        for gi in range(len(self.groups)):
            self.exchange_group(gi)
        self.wait_all()

    def wait_all(self):
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
                b.work = None

    def finish(self):
        """Wait for every exchange (parameters that received no gradient count as zero);
        returns the averaging factor to fold into the optimizer step."""
        if self.defer:  # capture pass of a hipGraph: nothing ran, nothing to exchange
            for b in self.buckets:
                b.pending = 0
            return 1.0 / self.world
        for b in self.buckets:
            b.pending = 0  # parameters without a gradient this step: their (zero-filled) slots go out as they are
        self._launch_ready()
        self.wait_all()
        return 1.0 / self.world

    def remove(self):
        from .. import ops

        for h in self._handles:
            h.remove()
        for p in self.params:
            e = ops.GRAD_SLOTS.get(p.data_ptr()) if p.is_cuda else None
            if e is not None and e[1]() is self:
                del ops.GRAD_SLOTS[p.data_ptr()]
