"""Training-mode dropout on the HIP kernel (``pwg_dropout``): stand-in for ``torch.nn.Dropout`` /
``F.dropout`` at /root/reference/parallel_wavegan/layers/residual_block.py:115 and
models/uhifigan.py:86,130."""
import torch

from .. import functional as Fn


class Dropout(torch.nn.Module):
    """No parameters, no state-dict entries: identity in eval mode, counter-based mask in training.
    The mask seed is (host seed) + (a device counter that the forward advances), so eager steps and
    hipGraph replays both draw a new mask every call; the backward regenerates the same mask.
    ``last_seeds`` = (host seed, device counter value) of the latest call, for tests that rebuild the
    mask (``hash(seed + counter, index) >= p * 2^32`` keeps an element, see csrc/elementwise.hip)."""

    def __init__(self, p):
        super().__init__()
        self.p = float(p)
        self._counter = None  # int64 device scalar, created lazily on the input's device
        self.last_seeds = None
        self._uid = None

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        if self._counter is None or self._counter.device != x.device:
            self._counter = torch.zeros(1, dtype=torch.int64, device=x.device)
        used = self._counter.clone()  # the value this call (and its backward) uses
        self._counter += 7919           # plumbing: advance the device counter for the next call / replay
        if self._uid is None:
            # per-layer stream id drawn from torch's CPU generator at the first training call: masks are
            # reproducible after torch.manual_seed() (like torch.nn.Dropout's), independent of object addresses
            self._uid = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        seed = (int(torch.initial_seed()) * 1000003 + self._uid) & ((1 << 62) - 1)
        self.last_seeds = (seed, used)
        return Fn.DropoutFn.apply(x, self.p, seed, used)

    def extra_repr(self):
        return f"p={self.p}"
