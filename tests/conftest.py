import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def device():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _poison_lds(request):
    """PWG_TEST_POISON_LDS=1: fill every CU's LDS with NaN bit patterns before each GPU test
    (tools/probes/lds_poison.hip, built with hipcc -shared) so that any kernel whose result depends
    on stale LDS content -- e.g. 0 * unwritten-tile-element in a contraction -- fails deterministically
    instead of once in a while on a cold GPU."""
    if os.environ.get("PWG_TEST_POISON_LDS") and request.node.get_closest_marker("gpu") is not None:
        import ctypes

        import torch

        so = os.path.join(ROOT, "tools", "probes", "lds_poison.so")
        if os.path.exists(so) and torch.cuda.is_available():
            lib = ctypes.CDLL(so)
            sink = torch.zeros(4, device="cuda:0")
            lib.lds_poison(ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
    yield
