"""Scalar restatement of the reference's RAdam update (optimizers/radam.py:27-99) for one tensor.
TEST INFRASTRUCTURE (see oracle/__init__.py)."""
import math

import torch


def radam_step(p, g, exp_avg, exp_avg_sq, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """In-place update of p / exp_avg / exp_avg_sq for the 1-based ``step``; returns nothing."""
    b1, b2 = betas
    exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
    exp_avg.mul_(b1).add_(g, alpha=1 - b1)
    beta2_t = b2 ** step
    n_sma_max = 2 / (1 - b2) - 1
    n_sma = n_sma_max - 2 * step * beta2_t / (1 - beta2_t)
    if n_sma >= 5:
        step_size = math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_sma_max - 4) * (n_sma - 2) / n_sma * n_sma_max
                              / (n_sma_max - 2)) / (1 - b1 ** step)
    else:
        step_size = 1.0 / (1 - b1 ** step)
    if weight_decay != 0:
        p.add_(p, alpha=-weight_decay * lr)
    if n_sma >= 5:
        p.addcdiv_(exp_avg, exp_avg_sq.sqrt().add_(eps), value=-step_size * lr)
    else:
        p.add_(exp_avg, alpha=-step_size * lr)
