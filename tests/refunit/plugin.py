"""pytest plugin under which the reference's OWN unit tests (oracle/_ref/test/*.py, staged unedited by
oracle/make_ref.py) run against the drop-in package on an MI355X.

    python -m pytest -p tests.refunit.plugin oracle/_ref/test/test_layers.py

What the plugin does, and nothing else:
  * ``parallelwavegan_amd.compat.install()``: ``import parallel_wavegan...`` resolves to this engine;
  * ``torch.set_default_device("cuda")``: the reference's tests create their inputs with bare ``torch.randn(...)``
    (CPU in the reference's CI); this engine has no CPU path, so factory calls default to the GPU;
  * ``Tensor.numpy()`` on a device tensor copies to the host first (the tests call ``.numpy()`` on inputs and
    parameters they created on what is now the GPU);
  * ``MelSpectrogram``: the reference test feeds a float64 host array and casts the module to double; the engine
    computes in fp32 on the device, so the test's input is moved and cast at the module boundary
    (``parallelwavegan_amd.losses.MelSpectrogram`` itself stays strict).
The tests themselves are byte-identical to the reference's (sha256 manifest, tests/test_bench_host.py).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    import torch

    from parallelwavegan_amd import compat

    compat.install()
    if not torch.cuda.is_available():
        if config.option.collectonly:  # import / collection check in the build container
            return
        raise RuntimeError("the reference's unit tests run against the gfx950 kernels: no GPU visible")
    torch.set_default_device("cuda")
    orig_numpy = torch.Tensor.numpy

    def numpy(self, *a, **k):
        return orig_numpy(self.detach().cpu() if self.is_cuda else self, *a, **k)

    torch.Tensor.numpy = numpy

    # the one test that hands the engine a float64 HOST array (test_mel_loss.py: torch.from_numpy(x) into a module cast
    # with .to(dtype=torch.double)): input and buffers are brought to the device / fp32 at the module boundary
    from parallelwavegan_amd.losses import MelSpectrogram

    orig_forward = MelSpectrogram.forward

    def forward(self, x):
        if any(b.dtype != torch.float32 for b in self.buffers() if b.is_floating_point()):
            self.float()
        return orig_forward(self, x.to(device="cuda", dtype=torch.float32))

    MelSpectrogram.forward = forward

    # modules of the reference (and of this package) build constant buffers from numpy arrays
    # (``torch.from_numpy``: host tensors whatever the default device is) and the reference's tests never call
    # ``.to(device)``: buffers follow the default device like the parameters do
    orig_register = torch.nn.Module.register_buffer

    def register_buffer(self, name, tensor, *a, **k):
        if isinstance(tensor, torch.Tensor) and not tensor.is_cuda:
            tensor = tensor.to("cuda")
        return orig_register(self, name, tensor, *a, **k)

    torch.nn.Module.register_buffer = register_buffer
