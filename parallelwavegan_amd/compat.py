"""Opt-in import alias: after ``parallelwavegan_amd.compat.install()`` the reference's import paths
resolve to this package, so user code written against kan-bayashi/ParallelWaveGAN --

    from parallel_wavegan.utils import load_model
    from parallel_wavegan.models import HiFiGANGenerator
    from parallel_wavegan.losses import MultiResolutionSTFTLoss
    from parallel_wavegan.bin.train import Trainer, Collater

-- runs on the MI355X kernels without edits.  It is opt-in (not a top-level ``parallel_wavegan``
directory) so that an installed reference package is never shadowed by accident; ``install()``
refuses to run when the real ``parallel_wavegan`` has already been imported.
"""
import importlib
import sys

_SUBMODULES = ("models", "layers", "losses", "optimizers", "utils", "utils.utils", "bin", "bin.train",
               "bin.preprocess", "distributed", "distributed.launch")


def install(force=False):
    import parallelwavegan_amd as pkg

    cur = sys.modules.get("parallel_wavegan")
    if cur is not None and cur is not pkg and not force:
        raise RuntimeError("the reference package `parallel_wavegan` is already imported "
                           f"({getattr(cur, '__file__', '?')}); call install() first or pass force=True")
    if force:
        for name in [n for n in sys.modules if n == "parallel_wavegan" or n.startswith("parallel_wavegan.")]:
            del sys.modules[name]
    sys.modules["parallel_wavegan"] = pkg
    for sub in _SUBMODULES:
        sys.modules[f"parallel_wavegan.{sub}"] = importlib.import_module(f"parallelwavegan_amd.{sub}")
    return pkg


def uninstall():
    import parallelwavegan_amd as pkg

    if sys.modules.get("parallel_wavegan") is pkg:
        for name in [n for n in sys.modules if n == "parallel_wavegan" or n.startswith("parallel_wavegan.")]:
            del sys.modules[name]
