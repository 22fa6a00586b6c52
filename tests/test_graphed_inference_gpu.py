"""GPU: ``GraphedInference`` never replays stale weight images (a parameter update without ``reset()``
used to synthesise with the weights of capture time)."""
import pytest
import torch

from parallelwavegan_amd import models, ops
from parallelwavegan_amd.graphs import GraphedInference
from tests.util import max_abs

pytestmark = pytest.mark.gpu


def _small_hifigan(device):
    torch.manual_seed(7)
    g = models.HiFiGANGenerator(in_channels=80, out_channels=1, channels=64, kernel_size=7, upsample_scales=[4, 4],
                                upsample_kernel_sizes=[8, 8], resblock_kernel_sizes=[3, 7],
                                resblock_dilations=[[1, 3], [1, 3]])
    return g.to(device).eval()


def test_replay_follows_every_kind_of_parameter_update(device):
    g = _small_hifigan(device)
    run = GraphedInference(g)
    c = torch.randn(2, 80, 24, device=device)
    with torch.no_grad():
        y0 = run(c).clone()
        graph0 = run._graphs[next(iter(run._graphs))][0]
        y0b = run(c).clone()
        assert run._graphs[next(iter(run._graphs))][0] is graph0, "an unchanged model must replay, not capture again"
        assert torch.equal(y0, y0b) and max_abs(y0, g(c)) == 0.0

        # 1. in-place update through torch (bumps the version counter)
        p = g.output_conv[1].weight_v if hasattr(g.output_conv[1], "weight_v") else next(g.parameters())
        p.mul_(1.5)
        y1 = run(c).clone()
        assert run._graphs[next(iter(run._graphs))][0] is not graph0
        assert max_abs(y1, g(c)) == 0.0 and not torch.equal(y1, y0)

        # 2. update through raw pointers, as the fused optimizers do (no version bump; the engine's own epoch)
        graph1 = run._graphs[next(iter(run._graphs))][0]
        q = g.input_conv.bias
        torch.add(q.detach(), 0.25, out=torch.empty_like(q)).clone()  # (an unrelated op must not invalidate anything)
        assert run(c) is not None and run._graphs[next(iter(run._graphs))][0] is graph1
        q.data.add_(0.25)  # .data hides the update from the version counter of ``q``
        ops.bump_params([q])
        y2 = run(c).clone()
        assert run._graphs[next(iter(run._graphs))][0] is not graph1
        assert max_abs(y2, g(c)) == 0.0 and not torch.equal(y2, y1)

        # 3. load_state_dict / remove_weight_norm
        sd = {k: v.clone() for k, v in g.state_dict().items()}
        for k in sd:
            if k.endswith("weight_g"):
                sd[k] = sd[k] * 0.9
        g.load_state_dict(sd)
        y3 = run(c).clone()
        assert max_abs(y3, g(c)) == 0.0 and not torch.equal(y3, y2)
        g.remove_weight_norm()
        y4 = run(c).clone()
        assert max_abs(y4, g(c)) == 0.0 and max_abs(y4, y3) <= 1e-5


def test_branches_fork_onto_side_streams_only_inside_a_capture(device):
    """streams.fork_now (round 5): eager execution runs independent branches in order on the caller's
    stream; inside a hipGraph capture they are forked (and become parallel branches of the graph).  Results are the same
    either way."""
    from parallelwavegan_amd import streams

    cur = torch.cuda.current_stream(device)
    seen = []

    def branch(k):
        def fn():
            seen.append(torch.cuda.current_stream(device).cuda_stream)
            return torch.full((4,), float(k), device=device) * 2.0
        return fn

    outs = streams.run_branches([branch(k) for k in range(3)], device, True)
    assert seen == [cur.cuda_stream] * 3 and [o[0].item() for o in outs] == [0.0, 2.0, 4.0]
    streams.reserve(device)
    seen.clear()
    side = torch.cuda.Stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            origin = torch.cuda.current_stream(device).cuda_stream
            outs = streams.run_branches([branch(k) for k in range(3)], device, True)
    cur.wait_stream(side)
    g.replay()
    torch.cuda.synchronize()
    assert len(set(seen)) == 3 and origin not in seen, "three branches, three side streams, none the capturing stream"
    assert [o[0].item() for o in outs] == [0.0, 2.0, 4.0]


def test_branch_inputs_are_recorded_on_the_side_streams(device, monkeypatch):
    """Round 6 root cause of the eager multi-stream NaN (profiles/r06_eager_nan_bisect.txt): a branch INPUT allocated on
    the caller's stream and read on a side stream must be ``record_stream``'ed there, or the caching allocator hands
    its memory to the caller's next allocation of that size while the side stream's reader is still queued.  Here the
    side streams are kept busy (``torch.cuda._sleep``), the input is dropped, and the next allocation of the same
    size on the caller's stream must NOT receive the input's block; without ``inputs=`` it does (the control)."""
    from parallelwavegan_amd import streams

    monkeypatch.setattr(streams, "EAGER_FORK", True)
    n = 16 * 4097  # (the pooled input of HiFi-GAN's second scale discriminator)

    def attempt(declare_inputs):
        torch.cuda.synchronize()
        x = torch.ones(n, device=device)
        ptr = x.data_ptr()

        def branch():
            torch.cuda._sleep(200_000_000)  # ~0.1 s: the reader below stays queued behind it
            return x * 2.0

        outs = streams.run_branches([branch, branch], device, True, inputs=x if declare_inputs else None)
        del x, branch
        # a different stream than the joined caller's stream would not even wait for the readers; the allocator hands a
        # block of a stream's pool only to that stream, so ask on the caller's stream
        y = torch.empty(n, device=device)
        reused = y.data_ptr() == ptr
        torch.cuda.synchronize()
        assert all(bool((o == 2.0).all()) for o in outs)
        return reused

    assert attempt(False), "control: without inputs= the block goes straight back to the caller's pool"
    assert not attempt(True), "a recorded input's block must stay out of the pool while a side stream may read it"


def test_model_built_under_inference_mode_and_frozen_flag(device):
    """parameters created under ``torch.inference_mode()`` track no version counter (reading it raises);
    such a model must still be callable, and ``frozen=True`` skips the per-call walk over the tensors."""
    with torch.inference_mode():
        g = _small_hifigan(device)
        g.remove_weight_norm()
    assert any(p.is_inference() for p in g.parameters())
    c = torch.randn(1, 80, 20, device=device)
    with torch.inference_mode():
        ref = g(c).clone()
    run = GraphedInference(g)
    assert max_abs(run(c), ref) == 0.0 and max_abs(run(c), ref) == 0.0
    fast = GraphedInference(g, frozen=True)
    y = fast(c).clone()
    state = fast._state
    assert fast(c) is not None and fast._state is state and max_abs(y, ref) == 0.0
