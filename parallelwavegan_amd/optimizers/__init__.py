"""Optimizers selectable by name from the YAML (``generator_optimizer_type`` ...), as
``parallel_wavegan.optimizers`` does (it re-exports ``torch.optim`` plus RAdam)."""
from torch.optim import *  # noqa: F401,F403  (schedulers and other optimizers stay torch's)

from .fused import Adam, AdamW, RAdam, clip_grad_norm_  # noqa: F401  fused HIP versions shadow torch's Adam
