import os

import numpy as np
import torch

from tests.golden import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# north-star parity bar (BASELINE.json): generated waveforms <= 1e-4 max-abs fp32
WAVE_TOL = 1e-4


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def synth_for(module, seed, g_scale):
    return synth.synth_state_dict(module.state_dict(), seed=seed, g_scale=g_scale)


def max_abs(a, b):
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return (a - b).abs().max().item()
