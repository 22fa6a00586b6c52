"""hipGraph capture helpers (torch.cuda.graphs drives hipStreamBeginCapture on ROCm; every kernel of
libpwgkernels.so is launched on torch's current stream, so it is captured like any ATen kernel).

A GAN-vocoder forward is ~80 dependent launches of a few tens of microseconds each at batch 1; an
eager Python loop is host-bound there.  ``GraphedInference`` replays the whole generator forward
as one graph launch per utterance shape.
"""
import torch


class GraphedInference:
    """Capture ``model.forward`` for fixed input shapes; ``__call__`` copies the new inputs into the
    static buffers and replays.  One graph per distinct input shape (cached).

    The capture records the kernels that read the packed weight images of that moment.  Every call first compares the
    model's *parameter state* -- storage address, torch version counter and the engine's own epoch of every parameter
    and buffer (``ops.param_epoch``: the fused optimizers update through raw pointers), plus the global weight-cache
    epoch -- with the state at capture time; after ``load_state_dict``, an optimizer step, ``remove_weight_norm`` ...
    the stale graphs are dropped and the call captures again, so a replay can never synthesise with old weights
    (VERDICT r04).  :meth:`reset` does the same by hand."""

    def __init__(self, model, warmup=2, frozen=False):
        self.model = model
        self.warmup = warmup
        # frozen=True: latency-critical loops where the caller guarantees that no parameter changes between calls;
        # the parameter state is then taken once (at the first capture) instead of on every call
        self.frozen = frozen
        self._graphs = {}
        self._state = None

    def reset(self):
        """Drop the captured graphs (parameters changed)."""
        self._graphs = {}
        self._state = None

    def _param_state(self):
        from . import ops

        model = self.model
        if not isinstance(model, torch.nn.Module):
            return None
        if self.frozen and self._state is not None:
            return self._state  # the caller vouches for frozen weights: no per-call walk over the tensors
        st = [ops.PARAM_EPOCH[0], model.training]
        for t in model.parameters():
            st.append((t.data_ptr(), ops.tensor_version(t), ops.param_epoch(t)))
        for t in model.buffers():
            st.append((t.data_ptr(), ops.tensor_version(t)))
        return st

    def _capture(self, inputs):
        static_in = [t.clone() for t in inputs]
        if isinstance(self.model, torch.nn.Module) and any(getattr(m, "branch_streams", False) for m in self.model.modules()):
            from . import streams

            streams.reserve(static_in[0].device)  # branches fork only inside the capture: create their streams first
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(self.warmup):  # compiles nothing, but fills weight caches / sets kernel attributes
                self.model(*static_in)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(g):
            static_out = self.model(*static_in)
        return g, static_in, static_out

    @torch.no_grad()
    def __call__(self, *inputs):
        state = self._param_state()
        if state != self._state:
            if self._graphs:
                self._graphs = {}
            self._state = state
        key = tuple((tuple(t.shape), t.dtype) for t in inputs)
        if key not in self._graphs:
            self._graphs[key] = self._capture(inputs)
            # a forward in training mode may itself touch buffers (spectral-norm u / v): take the state AFTER the capture
            self._state = self._param_state()
        g, static_in, static_out = self._graphs[key]
        for s, t in zip(static_in, inputs):
            s.copy_(t, non_blocking=True)
        g.replay()
        return static_out
