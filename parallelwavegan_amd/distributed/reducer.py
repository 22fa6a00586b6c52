"""Data-parallel gradient exchange over RCCL (xGMI), one process per GPU.

Replaces ``apex.parallel.DistributedDataParallel`` at
/root/reference/parallel_wavegan/bin/train.py:1494-1503.  Parameters are assigned, in reverse
registration order (the order backward produces their gradients), to a few large flat fp32
buckets.  A post-accumulate hook copies each gradient into its bucket slot as soon as autograd
has produced it; when a bucket is complete its all-reduce (sum) is launched asynchronously, so
the exchange of the discriminator's 283 MB overlaps the rest of its backward pass.  The fused
optimizers then read the reduced gradients straight from the buckets (``flat_grads``) with the
1/world_size averaging folded into the update kernel (``grad_scale``) -- no copy back.

Buckets are large (default 64 MiB): xGMI is a point-to-point mesh (7 links per GPU), so a
collective is per-link bound and a few big messages beat many small ones.
"""
import torch
import torch.distributed as dist


class _Bucket:
    def __init__(self, params, device):
        self.params = params
        self.offsets = {}
        n = 0
        for p in params:
            self.offsets[p] = n
            n += p.numel()
        self.flat = torch.zeros(n, device=device, dtype=torch.float32)
        self.views = {p: self.flat[o:o + p.numel()].view_as(p) for p, o in self.offsets.items()}
        self.pending = len(params)
        self.work = None


class GradReducer:
    def __init__(self, params, bucket_bytes=64 << 20, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        params = [p for p in params]
        self.params = params
        self.buckets = []
        cur, cur_bytes = [], 0
        for p in reversed(params):
            cur.append(p)
            cur_bytes += p.numel() * 4
            if cur_bytes >= bucket_bytes:
                self.buckets.append(_Bucket(cur, p.device))
                cur, cur_bytes = [], 0
        if cur:
            self.buckets.append(_Bucket(cur, cur[0].device))
        self.bucket_of = {p: b for b in self.buckets for p in b.params}
        self.flat_grads = {p: b.views[p] for b in self.buckets for p in b.params}
        self._handles = [p.register_post_accumulate_grad_hook(self._hook) for p in params]
        self.enabled = True
        self.defer = False  # True while a hipGraph is being captured: hooks fill buckets, nothing is launched

    def broadcast_parameters(self, tensors, src=0):
        """Rank 0's parameters and buffers become everyone's (as DDP does at wrap time)."""
        if self.world > 1:
            for t in tensors:
                dist.broadcast(t.data, src, group=self.group)

    def prepare(self):
        """Call before each backward whose gradients are to be exchanged."""
        for b in self.buckets:
            b.pending = sum(1 for p in b.params if p.requires_grad)
            b.work = None
            b.flat.zero_()  # parameters that receive no gradient this step contribute exactly 0

    def _hook(self, p):
        if not self.enabled:
            return
        b = self.bucket_of[p]
        b.views[p].copy_(p.grad)
        p.grad = None  # the bucket slot now owns this gradient
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _launch(self, b):
        if self.world > 1 and not self.defer:
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def exchange_all(self):
        """All-reduce every bucket now and wait (hipGraph mode: the buckets were filled by a replayed
        graph, whose capture recorded the hook copies but no collective)."""
        if self.world > 1:
            works = [dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                     for b in self.buckets]
            for w in works:
                w.wait()

    def finish(self):
        """Wait for every exchange (parameters that received no gradient count as zero);
        returns the averaging factor to fold into the optimizer step."""
        if self.defer:  # capture pass of a hipGraph: nothing ran, nothing to exchange
            for b in self.buckets:
                b.pending = 0
            return 1.0 / self.world
        for b in self.buckets:
            if b.pending > 0:  # some parameters got no gradient: exchange the (zero-filled) rest
                self._launch(b)
                b.pending = 0
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
                b.work = None
        return 1.0 / self.world

    def remove(self):
        for h in self._handles:
            h.remove()
