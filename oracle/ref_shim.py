"""Import shim so the UNMODIFIED reference (/root/reference) runs in the build
container.  TEST INFRASTRUCTURE (see oracle/__init__.py).

/root/reference does not exist on the GPU box.  ``oracle/make_ref.py`` stages the reference's
``parallel_wavegan`` package byte for byte as ``oracle/_ref/`` (git-ignored, shipped with the snapshot like a
built ``.so``); when /root/reference is absent the shim resolves to that copy, so the GPU box can time the
reference's own CPU path.  Callers must check ``available()`` and skip.  What is shimmed and why:
  * ``scipy.signal.kaiser`` was removed in scipy>=1.13 (layers/pqmf.py:11)
  * h5py, librosa, soundfile, kaldiio, tensorboardX are not installed
    (utils/utils.py:16, losses/mel_loss.py:8, bin/train.py:17,20)
  * ``librosa.filters.mel`` is served by oracle/slaney_mel.py
"""
import os
import sys
import types

STAGED_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")
REF_ROOT = "/root/reference" if os.path.isdir("/root/reference/parallel_wavegan") else STAGED_ROOT


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "parallel_wavegan"))


def is_staged_copy():
    """True when ``REF_ROOT`` is the oracle/_ref copy (the GPU box), False for the reference checkout itself."""
    return REF_ROOT == STAGED_ROOT


class _NullWriter:
    def __init__(self, *a, **k):
        pass

    def add_scalar(self, *a, **k):
        pass

    def close(self):
        pass


def install():
    """Make ``import parallel_wavegan`` resolve to the reference tree."""
    if not available():
        raise RuntimeError("reference tree not present (expected on the GPU box)")
    import scipy.signal
    import scipy.signal.windows

    from . import slaney_mel

    if not hasattr(scipy.signal, "kaiser"):
        scipy.signal.kaiser = scipy.signal.windows.kaiser
    for name in ("h5py", "librosa", "soundfile", "kaldiio", "tensorboardX"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    lib = sys.modules["librosa"]
    if not hasattr(lib, "filters"):
        lib.filters = types.SimpleNamespace(
            mel=lambda sr, n_fft, n_mels=128, fmin=0.0, fmax=None, **kw: slaney_mel.mel(
                sr, n_fft, n_mels, fmin, fmax
            )
        )
    tb = sys.modules["tensorboardX"]
    if not hasattr(tb, "SummaryWriter"):
        tb.SummaryWriter = _NullWriter
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import parallel_wavegan  # noqa: F401

    return sys.modules["parallel_wavegan"]
