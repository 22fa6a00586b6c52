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
