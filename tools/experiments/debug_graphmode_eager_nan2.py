"""Debugging aid (round 5): where does the sporadic NaN of the eager branch-stream mode enter?  Finite-flags of every
discriminator feature map are taken WITHOUT host synchronisation (a) right after each discriminator forward and (b)
when the feature-matching loss reads the maps; printed after the run."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_train_full_shape_gpu import _build, load_golden  # noqa: E402
from tests.util import poison_empty, poison_lds  # noqa: E402

from parallelwavegan_amd import streams  # noqa: E402

streams.EAGER_FORK = True  # (debugging aid: the product forks only under capture)
if os.environ.get("NAN2_NO_INPUT_RECORD") == "1":
    # control of the round-6 fix: the branch inputs are NOT recorded on the side streams (the code as it was before)
    from parallelwavegan_amd.models import hifigan as _hifigan
    _rb = streams.run_branches

    def _rb_without_inputs(branches, device, enabled=True, inputs=None):
        return _rb(branches, device, enabled, None)

    streams.run_branches = _hifigan.run_branches = _rb_without_inputs
NODETAIL = os.environ.get("NAN2_NODETAIL") == "1"
dev = torch.device("cuda:0")
gold = load_golden("c3_train_full")
flags = []  # (label, device bool tensor)
detail = []
with poison_lds(), poison_empty():
    tr, batch, model, opt = _build("c3", gold, dev, use_hip_graph=True, graph_warmup_steps=100)
    disc = model["discriminator"]
    d_forward = disc.forward
    calls = [0]

    def fwd(x, *a, **k):
        out = d_forward(x, *a, **k)
        calls[0] += 1
        flags.append((f"step {tr.steps} D-call {calls[0]} input", torch.isfinite(x).all()))
        for i, maps in enumerate(out):
            for j, m in enumerate(maps or []):
                flags.append((f"step {tr.steps} D-call {calls[0]} produced disc {i} map {j} {tuple(m.shape)}", torch.isfinite(m).all()))
                if j == 0 and i in (0, 1, 2) and not NODETAIL:  # where inside the map are the non-finite values? (no host sync)
                    bad = ~torch.isfinite(m.reshape(-1))
                    b8 = bad.to(torch.int8)
                    detail.append((f"step {tr.steps} D-call {calls[0]} disc {i} map 0 ptr {m.data_ptr():#x} numel {m.numel()}",
                                   bad.sum(), torch.argmax(b8), m.numel() - 1 - torch.argmax(b8.flip(0)),
                                   bad.reshape(m.shape[0], -1).sum(dim=1), bad.reshape(-1, m.shape[-1]).sum(dim=1)[:256]))
        return out

    disc.forward = fwd
    if os.environ.get("NAN2_LAYER0") == "1":
        # round 6: what did the first layer of scale discriminator 1 see?  Flags taken on ITS stream right after its launch.
        conv0 = disc.msd.discriminators[1].layers[0][0]
        c0_forward = conv0.forward

        def c0(x, *a, **k):
            out = c0_forward(x, *a, **k)
            pw = conv0.prepared()
            sid = torch.cuda.current_stream().cuda_stream
            flags.append((f"step {tr.steps} L0 of disc 1 on stream {sid:#x}: INPUT x1", torch.isfinite(x).all()))
            flags.append((f"step {tr.steps} L0 of disc 1 on stream {sid:#x}: packed weight image", torch.isfinite(pw.fwd).all()))
            if pw.scale is not None:
                flags.append((f"step {tr.steps} L0 of disc 1 on stream {sid:#x}: weight-norm scale", torch.isfinite(pw.scale).all()))
            flags.append((f"step {tr.steps} L0 of disc 1 on stream {sid:#x}: bias", torch.isfinite(conv0.bias).all()))
            flags.append((f"step {tr.steps} L0 of disc 1 on stream {sid:#x}: PARAMETER weight_v", torch.isfinite(conv0.weight_v).all()))
            flags.append((f"step {tr.steps} L0 of disc 1 on stream {sid:#x}: PARAMETER weight_g", torch.isfinite(conv0.weight_g).all()))
            flags.append((f"step {tr.steps} L0 of disc 1 on stream {sid:#x}: OUTPUT", torch.isfinite(out).all()))
            return out

        conv0.forward = c0
    fm = tr.criterion["feat_match"]
    fm_forward = fm.forward

    def fmf(feats_hat, feats):
        for name, ff in (("hat", feats_hat), ("real", feats)):
            for i, maps in enumerate(ff):
                for j, m in enumerate(maps):
                    flags.append((f"step {tr.steps} FM-read {name} disc {i} map {j}", torch.isfinite(m).all()))
        return fm_forward(feats_hat, feats)

    fm.forward = fmf
    if os.environ.get("NAN2_WGRAD") == "1":
        # round 6: which operand of the first layer's weight gradient (scale discriminator 1) is non-finite, on its stream?
        from parallelwavegan_amd import ops as _ops
        import threading
        bw = _ops.conv1d_backward_weight_wn

        def bw_logged(desc, x, dy, v, g, *a, **k):
            hit = desc.c_in == 1 and desc.t_in == 4097
            if hit:
                sid = torch.cuda.current_stream().cuda_stream
                tag = f"step {tr.steps} wgrad of disc 1 L0 (thread {threading.get_ident() % 10000}, stream {sid:#x})"
                flags.append((tag + ": operand x (saved input)", torch.isfinite(x).all()))
                flags.append((tag + ": operand dy", torch.isfinite(dy).all()))
                flags.append((tag + ": v", torch.isfinite(v).all()))
                flags.append((tag + ": g", torch.isfinite(g).all()))
            out = bw(desc, x, dy, v, g, *a, **k)
            if hit:
                flags.append((tag + ": RESULT dv", torch.isfinite(out[0]).all()))
                flags.append((tag + ": RESULT dg", torch.isfinite(out[1]).all()))
                if out[2] is not None:
                    flags.append((tag + ": RESULT db", torch.isfinite(out[2]).all()))
            return out

        _ops.conv1d_backward_weight_wn = bw_logged
    if os.environ.get("NAN2_JOIN_AFTER_BACKWARD") == "1":
        # hypothesis (round 6): the optimizer on the caller's stream reads the LAST weight gradients a side stream produced
        # in the backward pass before they are written.  Make the caller's stream wait for every side stream after backward.
        t_backward = torch.Tensor.backward

        def backward_then_join(self_, *a, **k):
            r = t_backward(self_, *a, **k)
            cur_ = torch.cuda.current_stream()
            for pool in streams._POOL.values():
                for s_ in pool:
                    cur_.wait_stream(s_)
            return r

        torch.Tensor.backward = backward_then_join
    for _ in range(5):
        if os.environ.get("NAN2_LAYER0") == "1":
            c0m = disc.msd.discriminators[1].layers[0][0]
            for nm in ("weight_v", "weight_g", "bias"):
                p_ = getattr(c0m, nm)
                flags.append((f"BEFORE step {tr.steps + 1} (main stream): parameter {nm}", torch.isfinite(p_).all()))
                if p_.grad is not None:
                    flags.append((f"BEFORE step {tr.steps + 1} (main stream): .grad of {nm} left by the previous step", torch.isfinite(p_.grad).all()))
            st = tr.optimizer["discriminator"].state.get(c0m.weight_v, {})
            for k_, v_ in st.items():
                if torch.is_tensor(v_) and v_.is_floating_point() and v_.numel() > 1:
                    flags.append((f"BEFORE step {tr.steps + 1} (main stream): Adam state {k_} of weight_v", torch.isfinite(v_).all()))
        tr._train_step(batch)
    torch.cuda.synchronize()
bad = [lab for lab, f in flags if not bool(f.item())]
print(f"RESULT {len(bad)} non-finite of {len(flags)} checks")
for lab in bad[:24]:
    print("  ", lab)
for lab, n, first, last, per_item, per_row in detail:
    if int(n.item()):
        print("DETAIL", lab, "non-finite", int(n.item()), "first flat index", int(first.item()), "last", int(last.item()))
        print("   per batch item:", per_item.tolist())
        print("   per (item, channel) row, first 256 rows:", per_row.tolist())
