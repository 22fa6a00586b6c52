"""Randomised parity sweep of the conv1d family (forward, data gradient, weight/bias gradient) against
torch CPU fp32, with PWG_POISON_LDS=1 recommended (stale-LDS hazards).  GPU box only.
usage: fuzz_conv.py [n_cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from parallelwavegan_amd import ops

RTOL = 5e-5


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().double()
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)


def run(n_cases, seed):
    """Returns the list of failing cases (description, errors)."""
    rng = np.random.RandomState(seed)
    dev = torch.device("cuda:0")
    bad = []
    for case in range(n_cases):
        transposed = rng.rand() < 0.25
        groups = int(rng.choice([1, 1, 1, 2, 4, 16]))
        cin = groups * int(rng.choice([1, 2, 3, 8, 16, 20, 32, 64, 96]))
        cout = groups * int(rng.choice([1, 2, 4, 8, 16, 24, 32, 64, 128]))
        if cin > 1024 or cout > 1024:
            continue
        b = int(rng.choice([1, 2, 3, 5]))
        slope = [None, 0.1, 0.2, 0.0][rng.randint(4)]
        g = torch.Generator().manual_seed(case)
        if transposed:
            stride = int(rng.choice([1, 2, 3, 4, 5, 8]))
            k = int(rng.choice([stride, 2 * stride, 2 * stride + 1, stride + 3]))
            pad = int(rng.randint(0, max(1, k // 2)))
            out_pad = int(rng.randint(0, stride)) if stride > 1 else 0
            t = int(rng.choice([1, 5, 17, 33, 64, 100, 257]))
            t_out = (t - 1) * stride - 2 * pad + k + out_pad
            if t_out <= 0 or groups > 4:
                continue
            x = torch.randn(b, cin, t, generator=g, requires_grad=True)
            w = (torch.randn(cin, cout // groups, k, generator=g) / (cin // groups * k) ** 0.5).requires_grad_()
            bias = torch.randn(cout, generator=g, requires_grad=True)
            xa = F.leaky_relu(x, slope) if slope is not None else x
            y_ref = F.conv_transpose1d(xa, w, bias, stride=stride, padding=pad, output_padding=out_pad, groups=groups)
            desc = ops.make_conv_desc(b, cin, cout, t, t_out, k, stride, 1, pad, groups, transposed=True,
                                      pre_act="leaky_relu" if slope is not None else None, pre_slope=slope or 0.0)
        elif rng.rand() < 0.3:
            # (k, 1) Conv2d over (rows, width): the period-discriminator pattern, incl. long reductions over
            # few columns (split-K candidates)
            width = int(rng.choice([2, 3, 5, 7, 11]))
            cin = int(rng.choice([1, 32, 128, 512, 1024]))
            cout = int(rng.choice([1, 32, 128, 1024]))
            groups = 1
            k = int(rng.choice([3, 5]))
            stride = int(rng.choice([1, 3]))
            pad = (k - 1) // 2
            h = int(rng.choice([4, 10, 21, 51, 90]))
            h_out = (h + 2 * pad - k) // stride + 1
            if h_out <= 0:
                continue
            b = int(rng.choice([1, 2, 4]))
            x = torch.randn(b, cin, h, width, generator=g, requires_grad=True)
            w = (torch.randn(cout, cin, k, 1, generator=g) / (cin * k) ** 0.5).requires_grad_()
            bias = torch.randn(cout, generator=g, requires_grad=True)
            xa = F.leaky_relu(x, slope) if slope is not None else x
            y_ref = F.conv2d(xa, w, bias, stride=(stride, 1), padding=(pad, 0))
            desc = ops.make_conv_desc(b, cin, cout, h, h_out, k, stride, 1, pad, 1, width=width,
                                      pre_act="leaky_relu" if slope is not None else None, pre_slope=slope or 0.0)
            t = h
        else:
            k = int(rng.choice([1, 2, 3, 5, 7, 9, 11, 15, 41]))
            stride = int(rng.choice([1, 1, 1, 2, 3, 4]))
            dil = int(rng.choice([1, 1, 2, 3, 5, 9, 27])) if stride == 1 else 1
            pad = int(rng.choice([0, (k - 1) // 2 * dil, (k - 1) * dil]))
            t = int(rng.choice([(k - 1) * dil + 1, 31, 64, 97, 128, 200, 400, 777, 1500]))
            t_out = (t + 2 * pad - dil * (k - 1) - 1) // stride + 1
            if t_out <= 0 or t < (k - 1) * dil + 1 - 2 * pad:
                continue
            x = torch.randn(b, cin, t, generator=g, requires_grad=True)
            w = (torch.randn(cout, cin // groups, k, generator=g) / (cin // groups * k) ** 0.5).requires_grad_()
            bias = torch.randn(cout, generator=g, requires_grad=True)
            xa = F.leaky_relu(x, slope) if slope is not None else x
            y_ref = F.conv1d(xa, w, bias, stride=stride, padding=pad, dilation=dil, groups=groups)
            desc = ops.make_conv_desc(b, cin, cout, t, t_out, k, stride, dil, pad, groups,
                                      pre_act="leaky_relu" if slope is not None else None, pre_slope=slope or 0.0)
        dy = torch.randn(y_ref.shape, generator=g)
        y_ref.backward(dy)
        xd, wd, bd, dyd = (v.detach().to(dev).contiguous() for v in (x, w, bias, dy))
        if x.dim() == 4:  # device side works on (B, C, rows * width)
            xd, dyd = xd.reshape(b, cin, -1), dyd.reshape(b, cout, -1)
        tag = (f"case {case}: tr={int(transposed)} B={b} Cin={cin} Cout={cout} T={t}->{y_ref.shape[-1]} k={k} "
               f"s={stride} g={groups} slope={slope}")
        try:
            y = ops.conv1d_forward(desc, xd, ops.pack_weight(desc, wd), bd)
            dx = ops.conv1d_backward_data(desc, dyd, ops.pack_weight_bwd(desc, wd), xd)
            dw, db = ops.conv1d_backward_weight(desc, xd, dyd, tuple(w.shape))
            if x.dim() == 4:
                y, dx = y.reshape(y_ref.shape), dx.reshape(x.shape)
        except RuntimeError as e:
            if "unsupported" in str(e).lower() or "dilation with stride" in str(e):
                continue
            bad.append((tag, "exception " + str(e)[:120]))
            continue
        errs = dict(y=rel(y, y_ref), dx=rel(dx, x.grad), dw=rel(dw, w.grad), db=rel(db, bias.grad))
        worst = max(errs.values())
        if not (worst <= RTOL):
            bad.append((tag, {k2: f"{v:.2e}" for k2, v in errs.items()}))
    return bad


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    bad = run(n_cases, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f"{n_cases} cases, {len(bad)} failures")
    for t in bad[:30]:
        print("  FAIL", t)


if __name__ == "__main__":
    main()
