"""Fused multi-tensor optimizers on the HIP kernels (pwg_adam_step_dev / pwg_radam_step_dev).

They subclass ``torch.optim.Optimizer`` only for parameter-group bookkeeping, LR schedulers and a
``state_dict`` layout interchangeable with ``torch.optim.Adam`` / the reference's
``parallel_wavegan/optimizers/radam.py``: per-parameter ``step``, ``exp_avg``, ``exp_avg_sq``
(+ ``max_exp_avg_sq`` with amsgrad).  One kernel launch updates every parameter of a group:
the launch reads a device table of chunks (pointer, length; 4 - 64 Ki elements, see ``_chunk_for``) built on the host.

hipGraph friendliness: the scalars of an update (lr, bias corrections, rectification, gradient
scale) live in 8 floats of device memory per group.  ``step()`` = ``prepare()`` (host: advance the
step count, compute the scalars, stage them through pinned memory) + the kernel launch.  While a
stream is being captured only the launch is recorded; the trainer calls ``prepare()`` before
every replay.
"""
import ctypes
import math

import numpy as np
import torch

from .. import _lib
from ..ops import _stream, bump_params

CHUNK = 65536


def _chunk_for(total):
    """Elements per chunk (= per workgroup) of an optimizer launch over ``total`` elements: a workgroup walks its chunk
    with 256 lanes, one dependent load / store round per 256 elements, so a launch lasts as long as ONE chunk takes
    however few chunks there are (round 6: 0.36 ms for MB-MelGAN's 8.6 M discriminator parameters in 64 Ki chunks,
    the same as for HiFi-GAN's 52 M).  Aim at >= 2048 chunks; element-wise updates do not depend on the cut."""
    c = 4096
    while c < CHUNK and c * 2048 < total:
        c *= 2
    return c


def _capturing():
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def _reserve(pool, rows):
    """Eager steps keep one pinned (rows, 6) int64 buffer in ``pool`` for the next hipGraph capture."""
    if not pool or pool[-1].shape[0] < rows:
        pool.append(torch.zeros((rows, 6), dtype=torch.int64, device="cpu").pin_memory())


def _take_reserved(pool, rows):
    for i in range(len(pool) - 1, -1, -1):
        if pool[i].shape[0] >= rows:
            return pool.pop(i)
    raise RuntimeError("no pinned staging buffer is reserved for this capture: run one eager step of the same "
                       "shapes before capturing the step in a hipGraph")


class _FusedBase(torch.optim.Optimizer):
    _entry = None  # name of the C entry point

    def __init__(self, params, defaults):
        super().__init__(params, defaults)
        self.grad_scale = 1.0
        self.flat_grads = None
        self._dev = {}  # group index -> dict(hyper_dev, table_dev, table_key, n_chunks, graph_tables)
        self._active = {}  # group index -> set of parameters that had a gradient at the last step()

    # ---- per-parameter state (torch-compatible layout)
    def _state_for(self, p, amsgrad):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if amsgrad:
                st["max_exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @staticmethod
    def _as_int(step):
        return int(step.item()) if isinstance(step, torch.Tensor) else int(step)

    def _pairs(self, group):
        """(param, grad) pairs; ``flat_grads`` (set by the DDP reducer) maps param -> reduced grad view."""
        out = []
        override = self.flat_grads
        for p in group["params"]:
            g = override.get(p) if override is not None else None
            if g is None:
                g = p.grad
            if g is None:
                continue
            if g.is_sparse:
                raise RuntimeError("sparse gradients are not supported")
            if not p.is_cuda:
                raise RuntimeError("fused optimizers run only on MI355X device tensors (no CPU fallback)")
            if not p.is_contiguous() or not g.is_contiguous() or p.dtype != torch.float32:
                raise RuntimeError("fused optimizers need contiguous fp32 parameters and gradients")
            out.append((p, g))
        return out

    def _slot(self, gi, device):
        d = self._dev.get(gi)
        if d is None:
            d = dict(hyper_dev=torch.zeros(8, device=device, dtype=torch.float32), table_dev=None, table_key=None,
                     n_chunks=0, graph_tables=[], reserve=[])
            self._dev[gi] = d
        return d

    def _table(self, d, entries, device):
        """Device table of chunks for this launch.

        Host staging never outlives its copy: every upload goes through a FRESH pinned tensor (torch's
        caching host allocator does not recycle a pinned block while a copy that reads it is pending),
        so a host that runs several steps ahead of the device cannot overwrite a table whose upload has
        not happened yet.  While a hipGraph is being captured the upload becomes a memcpy node that
        re-reads its pinned source at every replay; that source is owned by this capture
        (``graph_tables``) and never written again, so later eager steps or further captures (other
        batch shapes) cannot redirect an already captured optimizer launch to stale gradient memory."""
        rows = []
        chunk = _chunk_for(sum(e[0].numel() for e in entries))
        for p, g, m, v, vmax in entries:
            n = p.numel()
            pp, gp, mp, vp = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
            xp = vmax.data_ptr() if vmax is not None else 0
            for off in range(0, n, chunk):
                b = off * 4
                rows.append((pp + b, gp + b, mp + b, vp + b, (xp + b) if xp else 0, min(chunk, n - off)))
        arr = np.asarray(rows, dtype=np.int64)
        capturing = _capturing()
        if capturing:
            # pinned memory cannot be allocated while a stream is capturing: take a buffer that an earlier
            # eager step set aside for exactly this purpose (and that nothing else ever writes)
            pin = _take_reserved(d["reserve"], arr.shape[0])
            pin[: arr.shape[0]].copy_(torch.from_numpy(arr))
            dev = pin[: arr.shape[0]].to(device, non_blocking=True)
            d["graph_tables"].append((pin, dev))  # keep the memcpy node's source and target alive, untouched
            return dev, arr.shape[0]
        _reserve(d["reserve"], arr.shape[0])
        d["rows"] = arr.shape[0]
        key = hash(arr.tobytes())
        if key == d["table_key"] and d["table_dev"] is not None:
            return d["table_dev"], d["n_chunks"]  # same pointers as the previous eager step: table still valid
        pin = torch.from_numpy(arr).pin_memory()
        dev = pin.to(device, non_blocking=True)
        d["table_dev"], d["table_key"], d["n_chunks"] = dev, key, arr.shape[0]
        return dev, arr.shape[0]

    def reserve_for_capture(self):
        """Called by the trainer right BEFORE it starts capturing a step: make sure every group has a pinned
        chunk-table buffer for that capture (eager steps reserve one, but a capture consumes it, and a second
        capture -- another batch shape -- may follow without an eager step of this optimizer in between)."""
        for d in self._dev.values():
            if d.get("rows"):
                _reserve(d["reserve"], d["rows"])

    # ---- to be provided by subclasses
    def _hyper(self, group, t):
        raise NotImplementedError

    def _amsgrad(self, group):
        return False

    # ---- public
    def prepare(self):
        """Host half of a step: advance the step counts, refresh the device scalars.  Called by
        ``step()`` in eager mode and by the trainer before each hipGraph replay.  Only parameters
        that received a gradient at the last ``step()`` are advanced (torch.optim semantics)."""
        for gi, group in enumerate(self.param_groups):
            steps = set()
            dev = None
            active = self._active.get(gi)
            for p in group["params"]:
                if (active is None and p.requires_grad) or (active is not None and p in active):
                    st = self._state_for(p, self._amsgrad(group))
                    st["step"] = self._as_int(st["step"]) + 1
                    steps.add(st["step"])
                    dev = p.device
            if not steps:
                continue
            if len(steps) != 1:
                raise RuntimeError("fused optimizers need one common step count per parameter group")
            d = self._slot(gi, dev)
            h = self._hyper(group, steps.pop())
            # fresh pinned staging per step (see _table): a pending upload is never overwritten by a later step
            pin = torch.tensor(h + [float(self.grad_scale)], dtype=torch.float32, device="cpu").pin_memory()
            d["hyper_dev"].copy_(pin, non_blocking=True)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        capturing = _capturing()
        for gi, group in enumerate(self.param_groups):
            self._active[gi] = {p for p, _ in self._pairs(group)}
        if not capturing:
            self.prepare()
        lib = _lib.lib()
        for gi, group in enumerate(self.param_groups):
            pairs = self._pairs(group)
            if not pairs:
                continue
            d = self._slot(gi, pairs[0][0].device)
            entries = []
            for p, g in pairs:
                st = self._state_for(p, self._amsgrad(group))
                entries.append((p, g, st["exp_avg"], st["exp_avg_sq"], st.get("max_exp_avg_sq")))
            table, n_chunks = self._table(d, entries, pairs[0][0].device)
            fn = getattr(lib, self._entry)
            _lib.check(fn(ctypes.c_void_p(table.data_ptr()), n_chunks,
                          ctypes.c_void_p(d["hyper_dev"].data_ptr()), _stream()), self._entry)
        # parameters changed behind torch's back: invalidate the packed-weight caches of THESE parameters
        # (the other model's images -- e.g. the discriminator's during the generator step -- stay valid)
        for group in self.param_groups:
            bump_params(group["params"])
        return loss


# torch.optim keyword arguments that select an implementation, not an update rule: accepted and ignored
_NOOP_KWARGS = {"foreach": (None, True, False), "fused": (None, True, False), "capturable": (True, False),
                "differentiable": (False,), "maximize": (False,)}


def _reject_unknown(cls, kwargs):
    """Keyword arguments beyond ``_NOOP_KWARGS`` (e.g. ``maximize=True``) would change the update rule: refuse them
    instead of running the recipe with a different optimiser than it asked for."""
    for k, v in kwargs.items():
        if k not in _NOOP_KWARGS or v not in _NOOP_KWARGS[k]:
            raise TypeError(f"{cls.__name__}: unsupported option {k}={v!r} (the fused optimizer implements "
                            f"lr / betas / eps / weight_decay" + (" / amsgrad" if cls is not RAdam else "") + " only)")


class Adam(_FusedBase):
    """``torch.optim.Adam`` semantics (L2 weight decay, optional amsgrad) in one launch per group."""

    _entry = "pwg_adam_step_dev"

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **unused):
        _reject_unknown(type(self), unused)
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=amsgrad))

    def _amsgrad(self, group):
        return bool(group["amsgrad"])

    def _hyper(self, group, t):
        b1, b2 = group["betas"]
        lr = float(group["lr"])
        return [lr, float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                lr / (1.0 - b1 ** t), math.sqrt(1.0 - b2 ** t)]


class AdamW(Adam):
    """``torch.optim.AdamW`` semantics (decoupled weight decay ``p *= 1 - lr * wd`` before the Adam update; default
    ``weight_decay`` 1e-2 as torch's) on the same launch: the kernel takes the decay with a negative sign
    (include/pwg_kernels.h).  Two of the reference's recipes (egs/yesno/voc1/conf/*.v1.debug.yaml) name it."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, **unused):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, **unused)

    def _hyper(self, group, t):
        h = super()._hyper(group, t)
        h[4] = -abs(h[4])
        return h


class RAdam(_FusedBase):
    """Rectified Adam with the update rule of the reference's ``optimizers/radam.py:27-99``."""

    _entry = "pwg_radam_step_dev"

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, **unused):
        _reject_unknown(type(self), unused)
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))

    def _hyper(self, group, t):
        b1, b2 = group["betas"]
        lr = float(group["lr"])
        beta2_t = b2 ** t
        n_sma_max = 2.0 / (1.0 - b2) - 1.0
        n_sma = n_sma_max - 2.0 * t * beta2_t / (1.0 - beta2_t)
        if n_sma >= 5:
            step_size = lr * math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_sma_max - 4) * (n_sma - 2) / n_sma
                                       * n_sma_max / (n_sma_max - 2)) / (1 - b1 ** t)
            rect = 1.0
        else:
            step_size = lr / (1 - b1 ** t)
            rect = 0.0
        return [lr, float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), step_size, rect]


def clip_grad_norm_(params_and_grads, max_norm):
    """``torch.nn.utils.clip_grad_norm_`` (L2) over (param, grad) pairs with three HIP launches;
    returns a device tensor [total_norm, applied_coefficient] (no host sync)."""
    grads = [g for _, g in params_and_grads]
    if not grads:
        return None
    dev = grads[0].device
    rows = []
    for g in grads:
        n, gp = g.numel(), g.data_ptr()
        for off in range(0, n, CHUNK):
            rows.append((gp + off * 4, gp + off * 4, 0, 0, 0, min(CHUNK, n - off)))
    arr = np.asarray(rows, dtype=np.int64)
    capturing = _capturing()
    pool = _CLIP_RESERVE.setdefault(tuple(g.numel() for g in grads), [])  # one pool per parameter set (G, D)
    if capturing:
        pin = _take_reserved(pool, arr.shape[0])  # see _FusedBase._table
        pin[: arr.shape[0]].copy_(torch.from_numpy(arr))
        table = pin[: arr.shape[0]].to(dev, non_blocking=True)
        _GRAPH_KEEP.append(pin)  # the captured memcpy node re-reads it at every replay
    else:
        _reserve(pool, arr.shape[0])
        _CLIP_ROWS[id(pool)] = (pool, arr.shape[0])
        table = torch.from_numpy(arr).pin_memory().to(dev, non_blocking=True)  # fresh staging per call
    out = torch.empty(2, device=dev, dtype=torch.float32)
    ws = torch.empty(len(rows), device=dev, dtype=torch.float32)
    _lib.check(_lib.lib().pwg_clip_grad_norm(ctypes.c_void_p(table.data_ptr()), len(rows), float(max_norm),
                                             ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
                                             _stream()), "clip_grad_norm")
    out._keep = table
    return out


def reserve_clip_tables_for_capture():
    """Counterpart of :meth:`_FusedBase.reserve_for_capture` for the gradient-clipping chunk tables."""
    for pool, rows in _CLIP_ROWS.values():
        _reserve(pool, rows)


_GRAPH_KEEP = []
_CLIP_RESERVE = {}
_CLIP_ROWS = {}
