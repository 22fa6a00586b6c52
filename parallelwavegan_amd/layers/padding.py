"""Padding "marker" modules: in the reference ``ReflectionPad1d`` etc. are separate modules inside
``torch.nn.Sequential`` containers (models/melgan.py:71,136; layers/residual_stack.py:49); here the
padding is fused into the following convolution kernel and the marker only keeps the Sequential
indices (hence the state-dict keys) identical."""
import torch

_MODES = {"ReflectionPad1d": "reflect", "ReplicationPad1d": "replicate", "ConstantPad1d": "zero",
          "ZeroPad1d": "zero"}


class FusedPad(torch.nn.Module):
    def __init__(self, name, padding, **params):
        super().__init__()
        if name not in _MODES:
            raise NotImplementedError(f"padding {name!r} has no fused gfx950 path (supported: {sorted(_MODES)})")
        if name == "ConstantPad1d" and params.get("value", 0.0) != 0.0:
            raise NotImplementedError("ConstantPad1d with a non-zero value")
        self.name, self.padding, self.mode = name, int(padding), _MODES[name]

    def forward(self, x):  # pragma: no cover
        raise RuntimeError("FusedPad is fused into the next convolution kernel; it is never called")

    def extra_repr(self):
        return f"{self.name}({self.padding})"


def get_pad(name, padding, **params):
    return FusedPad(name, padding, **params)
