"""Activation "marker" modules.

In the reference, activations are separate ``torch.nn`` modules inside
``torch.nn.Sequential`` containers (e.g. layers/residual_block.py:186-196), which
fixes the state-dict key indices (``convs1.0.1.weight_g``: index 1 is the conv).
Here the activation is fused into the consuming convolution kernel, so these
modules only carry the hyper-parameters and keep the indices identical.
"""
import torch

_SUPPORTED = {"LeakyReLU": "leaky_relu", "ReLU": "relu"}


class FusedActivation(torch.nn.Module):
    def __init__(self, name="LeakyReLU", **params):
        super().__init__()
        if name not in _SUPPORTED:
            raise NotImplementedError(
                f"activation {name!r} has no fused gfx950 kernel (supported: {sorted(_SUPPORTED)})"
            )
        self.name = name
        self.kind = _SUPPORTED[name]
        # torch.nn.LeakyReLU default slope = 0.01
        self.slope = float(params.get("negative_slope", 0.01)) if name == "LeakyReLU" else 0.0

    def forward(self, x):  # pragma: no cover - never on the hot path
        raise RuntimeError("FusedActivation is fused into the next convolution kernel; it is never called")

    def extra_repr(self):
        return f"{self.name}, slope={self.slope}"


class PreActivated(list):
    """Feature maps of a discriminator in *deferred-activation* form: entry ``k < len - 1`` holds the convolution output
    BEFORE its LeakyReLU (``slope``); consumers apply the activation themselves -- the next convolution on load
    (``pre_act``), the feature-matching loss inside its reduction kernel.  The last entry (the logits, no activation in
    the reference) is an ordinary tensor.  Values seen by every consumer are those of the reference's post-activation
    maps; what disappears is the separate activation-gradient pass of the backward step (the mask is applied by the
    next convolution's data-gradient epilogue and by the loss's backward kernel)."""

    def __init__(self, items, slope):
        super().__init__(items)
        self.preact_slope = float(slope)


def deferrable(acts):
    """Can a chain with these activation markers run in deferred form?  (LeakyReLU with one slope in (0, 1): the
    operand activation of the weight-gradient kernels is max(v, slope * v).)"""
    slopes = {a.slope for a in acts}
    return len(acts) > 0 and all(a.kind == "leaky_relu" for a in acts) and len(slopes) == 1 and 0.0 < next(iter(slopes)) < 1.0


def set_deferred_activation(module, flag):
    """Switch every sub-network of ``module`` that supports the deferred-activation form; returns how many did."""
    n = 0
    for m in module.modules():
        if hasattr(m, "deferred_activation"):
            m.deferred_activation = bool(flag)
            n += 1
    return n
