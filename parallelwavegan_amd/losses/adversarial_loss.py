"""Adversarial losses (drop-in for parallel_wavegan.losses.adversarial_loss).  The per-output
reductions are HIP kernels; combining the handful of scalars is 0-dim tensor arithmetic."""
import torch

from .. import functional as Fn


def _mean(x):
    return Fn.ReduceFn.apply(x, None, "sum", 1.0 / x.numel(), 0.0)


class GeneratorAdversarialLoss(torch.nn.Module):
    """Reference: losses/adversarial_loss.py:12-58."""

    def __init__(self, average_by_discriminators=True, loss_type="mse"):
        super().__init__()
        self.average_by_discriminators = average_by_discriminators
        assert loss_type in ["mse", "hinge"], f"{loss_type} is not supported."
        self.criterion = self._mse_loss if loss_type == "mse" else self._hinge_loss

    def forward(self, outputs):
        if isinstance(outputs, (tuple, list)):
            adv_loss = 0.0
            for i, outputs_ in enumerate(outputs):
                if isinstance(outputs_, (tuple, list)):
                    outputs_ = outputs_[-1]  # feature-map lists: the last entry is the logits
                adv_loss = adv_loss + self.criterion(outputs_)
            if self.average_by_discriminators:
                adv_loss = adv_loss / (i + 1)
        else:
            adv_loss = self.criterion(outputs)
        return adv_loss

    def _mse_loss(self, x):
        return Fn.mse_to_const_mean(x, 1.0)

    def _hinge_loss(self, x):
        return -_mean(x)


class DiscriminatorAdversarialLoss(torch.nn.Module):
    """Reference: losses/adversarial_loss.py:61-123.  Returns (real_loss, fake_loss)."""

    def __init__(self, average_by_discriminators=True, loss_type="mse"):
        super().__init__()
        self.average_by_discriminators = average_by_discriminators
        assert loss_type in ["mse", "hinge"], f"{loss_type} is not supported."
        if loss_type != "mse":
            raise NotImplementedError("hinge discriminator loss has no gfx950 kernel yet (configs C2-C5 use mse)")
        self.fake_criterion = lambda x: Fn.mse_to_const_mean(x, 0.0)
        self.real_criterion = lambda x: Fn.mse_to_const_mean(x, 1.0)

    def forward(self, outputs_hat, outputs):
        if isinstance(outputs, (tuple, list)):
            real_loss = 0.0
            fake_loss = 0.0
            for i, (outputs_hat_, outputs_) in enumerate(zip(outputs_hat, outputs)):
                if isinstance(outputs_hat_, (tuple, list)):
                    outputs_hat_ = outputs_hat_[-1]
                    outputs_ = outputs_[-1]
                real_loss = real_loss + self.real_criterion(outputs_)
                fake_loss = fake_loss + self.fake_criterion(outputs_hat_)
            if self.average_by_discriminators:
                fake_loss = fake_loss / (i + 1)
                real_loss = real_loss / (i + 1)
        else:
            real_loss = self.real_criterion(outputs)
            fake_loss = self.fake_criterion(outputs_hat)
        return real_loss, fake_loss
