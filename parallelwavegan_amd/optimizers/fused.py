"""Fused multi-tensor optimizers on the HIP kernels (pwg_adam_step / pwg_radam_step).

They subclass ``torch.optim.Optimizer`` only for parameter-group bookkeeping, LR schedulers and a
``state_dict`` layout interchangeable with ``torch.optim.Adam`` / the reference's
``parallel_wavegan/optimizers/radam.py``: per-parameter ``step``, ``exp_avg``, ``exp_avg_sq``
(+ ``max_exp_avg_sq`` with amsgrad).  One kernel launch updates every parameter of a group:
the launch reads a device table of 64 Ki-element chunks (pointer, length) built on the host.
"""
import ctypes

import numpy as np
import torch

from .. import _lib
from ..ops import _stream, bump_param_epoch

CHUNK = 65536


class _FusedBase(torch.optim.Optimizer):
    _kernel = None
    _has_vmax = False

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

    def _build_table(self, entries, device):
        """entries: list of (p, g, m, v, vmax_or_None) tensors -> device int64 table (n_chunks, 6)."""
        rows = []
        for p, g, m, v, vmax in entries:
            n = p.numel()
            pp, gp, mp, vp = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
            xp = vmax.data_ptr() if vmax is not None else 0
            for off in range(0, n, CHUNK):
                cnt = min(CHUNK, n - off)
                b = off * 4
                rows.append((pp + b, gp + b, mp + b, vp + b, (xp + b) if xp else 0, cnt))
        table = np.asarray(rows, dtype=np.int64)
        return torch.from_numpy(table).to(device, non_blocking=True), len(rows)

    def _grads(self, group):
        """(param, grad) pairs; ``flat_grads`` (set by the DDP reducer) maps param -> reduced grad view."""
        out = []
        override = getattr(self, "flat_grads", None)
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


class Adam(_FusedBase):
    """``torch.optim.Adam`` semantics (L2 weight decay, optional amsgrad) in one launch per group."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **unused):
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=amsgrad)
        super().__init__(params, defaults)
        self.grad_scale = 1.0

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.lib()
        for group in self.param_groups:
            pairs = self._grads(group)
            if not pairs:
                continue
            by_step = {}
            for p, g in pairs:
                st = self._state_for(p, group["amsgrad"])
                step = self._as_int(st["step"]) + 1
                st["step"] = step
                by_step.setdefault(step, []).append((p, g, st["exp_avg"], st["exp_avg_sq"], st.get("max_exp_avg_sq")))
            for step, entries in by_step.items():
                table, n = self._build_table(entries, entries[0][0].device)
                b1, b2 = group["betas"]
                _lib.check(lib.pwg_adam_step(ctypes.c_void_p(table.data_ptr()), n, float(group["lr"]), float(b1),
                                             float(b2), float(group["eps"]), float(group["weight_decay"]), step,
                                             float(self.grad_scale), _stream()), "adam_step")
                self._keep = table  # the launch is asynchronous: keep the table alive until the next step
        bump_param_epoch()  # parameters changed behind torch's back: invalidate packed-weight caches
        return loss


class RAdam(_FusedBase):
    """Rectified Adam with the update rule of the reference's ``optimizers/radam.py:27-99``."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, **unused):
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.grad_scale = 1.0

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.lib()
        for group in self.param_groups:
            pairs = self._grads(group)
            if not pairs:
                continue
            by_step = {}
            for p, g in pairs:
                st = self._state_for(p, False)
                step = self._as_int(st["step"]) + 1
                st["step"] = step
                by_step.setdefault(step, []).append((p, g, st["exp_avg"], st["exp_avg_sq"], None))
            for step, entries in by_step.items():
                table, n = self._build_table(entries, entries[0][0].device)
                b1, b2 = group["betas"]
                _lib.check(lib.pwg_radam_step(ctypes.c_void_p(table.data_ptr()), n, float(group["lr"]), float(b1),
                                              float(b2), float(group["eps"]), float(group["weight_decay"]), step,
                                              float(self.grad_scale), _stream()), "radam_step")
                self._keep = table
        bump_param_epoch()
        return loss


def clip_grad_norm_(params_and_grads, max_norm):
    """``torch.nn.utils.clip_grad_norm_`` (L2) over (param, grad) pairs with three HIP launches;
    returns a device tensor [total_norm, applied_coefficient] (no host sync)."""
    entries = [(g, g, g, g, None) for _, g in params_and_grads]
    if not entries:
        return None
    dev = entries[0][0].device
    rows = []
    for g, *_ in entries:
        n, gp = g.numel(), g.data_ptr()
        for off in range(0, n, CHUNK):
            rows.append((gp + off * 4, gp + off * 4, 0, 0, 0, min(CHUNK, n - off)))
    table = torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(dev)
    out = torch.empty(2, device=dev, dtype=torch.float32)
    ws = torch.empty(len(rows), device=dev, dtype=torch.float32)
    _lib.check(_lib.lib().pwg_clip_grad_norm(ctypes.c_void_p(table.data_ptr()), len(rows), float(max_norm),
                                             ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
                                             _stream()), "clip_grad_norm")
    out._keep = table
    return out
