"""Host logic of parallelwavegan_amd.streams (no GPU): when branches fork, and that run_branches degrades to in-order calls."""
import torch

from parallelwavegan_amd import ops, streams


def test_eager_fork_is_refused_while_a_data_parallel_reducer_is_active(monkeypatch):
    """Round 6: the eager fork (PWG_EAGER_BRANCH_STREAMS=1) is a single-process debugging mode; with gradient slots registered by a
    data-parallel reducer it is refused (profiles/r06_eager_nan_bisect.txt).  Without the switch nothing forks outside a capture."""
    monkeypatch.setattr(streams, "EAGER_FORK", False)
    assert streams.fork_now() is False
    monkeypatch.setattr(streams, "EAGER_FORK", True)
    monkeypatch.setattr(ops, "GRAD_SLOTS", {})
    assert streams.fork_now() is True
    monkeypatch.setattr(ops, "GRAD_SLOTS", {1234: ("slot", None, 0)})
    assert streams.fork_now() is False


def test_run_branches_on_cpu_calls_the_branches_in_order():
    seen = []
    outs = streams.run_branches([lambda k=k: seen.append(k) or torch.full((2,), float(k)) for k in range(3)],
                                torch.device("cpu"), True, inputs=[torch.zeros(1)])
    assert seen == [0, 1, 2] and [o[0].item() for o in outs] == [0.0, 1.0, 2.0]
