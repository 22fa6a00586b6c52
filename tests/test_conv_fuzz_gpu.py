"""GPU: seeded random sweep of the conv1d family (forward, data gradient, weight / bias gradient; plain,
strided, dilated, grouped and transposed; ragged lengths down to a single output column) against torch
CPU fp32 -- tools/fuzz_conv.py run for a fixed seed."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_random_conv_configurations(device):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_conv

    bad = fuzz_conv.run(120, 7)
    assert not bad, bad[:5]
