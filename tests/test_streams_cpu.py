"""Host logic of parallelwavegan_amd.streams (no GPU): when branches fork, and that run_branches degrades to in-order calls."""
import torch

from parallelwavegan_amd import ops, streams


def test_fork_only_inside_a_capture_unless_the_debugging_switch_is_set(monkeypatch):
    monkeypatch.setattr(streams, "EAGER_FORK", False)
    assert streams.fork_now() is False  # (no GPU here: never capturing)
    monkeypatch.setattr(streams, "EAGER_FORK", True)
    assert streams.fork_now() is True


def test_slot_add_without_a_device_is_a_plain_addition():
    """ops.slot_add (round 6): CPU tensors / `later is True` take the plain in-place addition; the cross-stream form is
    exercised by tests/test_ddp_graph_gpu.py."""
    slot = torch.ones(2, 3)
    ops.slot_add(slot, torch.full((6,), 2.0), True)
    assert torch.equal(slot, torch.full((2, 3), 3.0))


def test_run_branches_on_cpu_calls_the_branches_in_order():
    seen = []
    outs = streams.run_branches([lambda k=k: seen.append(k) or torch.full((2,), float(k)) for k in range(3)],
                                torch.device("cpu"), True, inputs=[torch.zeros(1)])
    assert seen == [0, 1, 2] and [o[0].item() for o in outs] == [0.0, 1.0, 2.0]
