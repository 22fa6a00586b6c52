"""Adversarial losses (drop-in for parallel_wavegan.losses.adversarial_loss).

Reference: /root/reference/parallel_wavegan/losses/adversarial_loss.py -- per discriminator output an
``F.mse_loss`` against ones / zeros (:54-58, :113-117) or a hinge term (:119-123), summed (and
optionally averaged) in Python.  Here every discriminator's logits are one item of a multi-tensor
reduction (``pwg_multi_reduce_*``), the 1/numel of the mean and the 1/#discriminators average folded
into the item weight: one launch pair for the generator loss, one for the (real, fake) pair.
"""
import torch

from .. import functional as Fn


def _logits(outputs):
    """Per-discriminator logits: a tensor, a list of tensors, or a list of feature-map lists whose
    last entry is the logits (adversarial_loss.py:33-37)."""
    if not isinstance(outputs, (tuple, list)):
        return [outputs], False
    return [o[-1] if isinstance(o, (tuple, list)) else o for o in outputs], True


class _AdvBase(torch.nn.Module):
    def __init__(self, average_by_discriminators=True, loss_type="mse"):
        super().__init__()
        self.average_by_discriminators = average_by_discriminators
        assert loss_type in ["mse", "hinge"], f"{loss_type} is not supported."
        self.loss_type = loss_type

    def _weight(self, n_disc, is_list):
        return 1.0 / n_disc if (is_list and self.average_by_discriminators) else 1.0


class GeneratorAdversarialLoss(_AdvBase):
    """mse: mean (D(G) - 1)^2 ;  hinge: -mean D(G)   (adversarial_loss.py:12-58)."""

    def forward(self, outputs):
        xs, is_list = _logits(outputs)
        w = self._weight(len(xs), is_list)
        if self.loss_type == "mse":
            spec = [("sq_diff_const", w / x.numel(), 1.0, 0) for x in xs]
        else:
            spec = [("sum", -w / x.numel(), 0.0, 0) for x in xs]
        tensors = [t for x in xs for t in (x, None)]
        return Fn.MultiReduceFn.apply(spec, 1, *tensors)[0]


class DiscriminatorAdversarialLoss(_AdvBase):
    """Returns (real_loss, fake_loss).  mse: mean (D(y) - 1)^2, mean D(G)^2 ;
    hinge: -mean min(D(y) - 1, 0), -mean min(-D(G) - 1, 0)   (adversarial_loss.py:61-123)."""

    def forward(self, outputs_hat, outputs):
        fake, is_list = _logits(outputs_hat)
        real, _ = _logits(outputs)
        assert len(fake) == len(real)
        w = self._weight(len(real), is_list)
        if self.loss_type == "mse":
            real_mode, real_c, fake_mode, fake_c = "sq_diff_const", 1.0, "sq_diff_const", 0.0
        else:
            real_mode, real_c, fake_mode, fake_c = "hinge_real", 0.0, "hinge_fake", 0.0
        spec, tensors = [], []
        for x in real:
            spec.append((real_mode, w / x.numel(), real_c, 0))
            tensors += [x, None]
        for x in fake:
            spec.append((fake_mode, w / x.numel(), fake_c, 1))
            tensors += [x, None]
        out = Fn.MultiReduceFn.apply(spec, 2, *tensors)
        return out[0], out[1]
