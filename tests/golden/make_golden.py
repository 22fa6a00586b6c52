"""Generate golden vectors by running the UNMODIFIED reference (/root/reference)
on CPU through oracle/ref_shim.py.  Run in the build container only:

    python tests/golden/make_golden.py [names...]

Inputs and weights are regenerated from seeds by tests/golden/synth.py, so only
the reference's outputs are stored (float32 .npz, a few hundred KB in total).
"""
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402
from tests.golden import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
warnings.filterwarnings("ignore")
G_SCALE = 1.25


def _stats(t):
    t = t.double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


def hifigan_generator(name, cfg, batch, frames, seed):
    import parallel_wavegan.models as RM

    g = RM.HiFiGANGenerator(**cfg).eval()
    sd = synth.synth_state_dict(g.state_dict(), seed=seed, g_scale=G_SCALE)
    g.load_state_dict(sd)
    c = synth.synth_input("c", (batch, cfg["in_channels"], frames), seed=seed)
    feats = []
    hooks = [g.input_conv.register_forward_hook(lambda m, i, o: feats.append(o))]
    for up in g.upsamples:
        hooks.append(up.register_forward_hook(lambda m, i, o: feats.append(o)))
    pre = []
    hooks.append(g.output_conv[1].register_forward_hook(lambda m, i, o: pre.append(o)))
    with torch.no_grad():
        y = g(c)
        # same weights after remove_weight_norm + the (T, C) inference API
        g.remove_weight_norm()
        y_inf = g.inference(c[0].transpose(0, 1).numpy())
    for h in hooks:
        h.remove()
    np.savez_compressed(
        os.path.join(OUT, f"{name}.npz"),
        y=y.numpy(),
        y_inference=y_inf.numpy(),
        y_pre_tanh=pre[0].numpy(),
        input_conv_head=feats[0][:, :8, :].numpy(),
        upsample_stats=np.stack([_stats(f) for f in feats[1:]]),
        upsample0_head=feats[1][:, :4, :64].numpy(),
        meta=np.array([batch, frames, seed], dtype=np.int64),
        g_scale=np.float64(G_SCALE),
    )
    print(name, "y", tuple(y.shape), "std %.4f max %.4f pre-tanh std %.3f" % (y.std().item(), y.abs().max().item(), pre[0].std().item()))


JOBS = {
    "hifigan_v1_g": lambda: hifigan_generator("hifigan_v1_g", synth.HIFIGAN_V1, 2, 32, 11),
    "hifigan_v1_libritts_g": lambda: hifigan_generator("hifigan_v1_libritts_g", synth.HIFIGAN_V1_LIBRITTS, 1, 28, 12),
    "hifigan_tiny_g": lambda: hifigan_generator("hifigan_tiny_g", synth.HIFIGAN_TINY, 3, 21, 13),
}

if __name__ == "__main__":
    ref_shim.install()
    torch.set_num_threads(os.cpu_count())
    names = sys.argv[1:] or list(JOBS)
    for n in names:
        JOBS[n]()
