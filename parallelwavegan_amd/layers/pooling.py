"""Pooling modules on the HIP kernels (drop-in for the ``torch.nn`` poolings the reference
instantiates by name, e.g. /root/reference/parallel_wavegan/models/hifigan.py:757-759)."""
import torch

from .. import functional as Fn


class AvgPool1d(torch.nn.Module):
    def __init__(self, kernel_size, stride=None, padding=0, ceil_mode=False, count_include_pad=True):
        super().__init__()
        if ceil_mode:
            raise NotImplementedError("AvgPool1d: ceil_mode is not used on the hot path")
        self.kernel_size = int(kernel_size)
        self.stride = int(stride if stride is not None else kernel_size)
        self.padding = int(padding)
        self.count_include_pad = bool(count_include_pad)

    def forward(self, x):
        return Fn.avg_pool1d(x, self.kernel_size, self.stride, self.padding, self.count_include_pad)

    def extra_repr(self):
        return (f"kernel_size={self.kernel_size}, stride={self.stride}, padding={self.padding}, "
                f"count_include_pad={self.count_include_pad}")


POOLINGS = {"AvgPool1d": AvgPool1d}


def get_pooling(name, **params):
    if name not in POOLINGS:
        raise NotImplementedError(f"pooling {name!r} has no gfx950 kernel (supported: {sorted(POOLINGS)})")
    return POOLINGS[name](**params)
