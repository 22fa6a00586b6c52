"""Adversarial (mse + hinge) and feature-matching losses against fixtures produced by the reference's
own classes (tests/golden/make_golden.py: adv_losses; losses/adversarial_loss.py, feat_match_loss.py).
CPU: the oracle restatement; GPU: the multi-tensor HIP reductions (values and gradients)."""
import numpy as np
import pytest
import torch

from oracle import torch_cpu
from tests.golden import synth
from tests.util import load_golden

VAL_TOL = 2e-6  # relative; sums of <= 200 fp32 terms
GRAD_TOL = 1e-7  # absolute; gradients are +-w/n or 2 w (x - c)/n with |.| <= 0.1


def _cat_grads(lists, last_only):
    ts = [o[-1] for o in lists] if last_only else [t for o in lists for t in o]
    return np.concatenate([(t.grad if t.grad is not None else torch.zeros_like(t)).detach().cpu().numpy().ravel()
                           for t in ts])


def _check(gold, fns, dev):
    seed = int(gold["meta"][0])

    def fresh(s):
        return [[t.clone().to(dev).requires_grad_() for t in o] for o in synth.adv_logits(s)]

    for lt in ("mse", "hinge"):
        for avg in (True, False):
            tag = f"{lt}_{int(avg)}"
            fake = fresh(seed + 1)
            g = fns["gen"](fake, avg, lt)
            g.backward()
            assert abs(g.item() - gold[f"gen_{tag}"]) <= VAL_TOL * max(1.0, abs(gold[f"gen_{tag}"])), tag
            assert np.abs(_cat_grads(fake, True) - gold[f"gen_{tag}_grad"]).max() <= GRAD_TOL, tag
            fake, real = fresh(seed + 1), fresh(seed)
            r, f = fns["dis"](fake, real, avg, lt)
            (r + 2.0 * f).backward()
            assert abs(r.item() - gold[f"dis_real_{tag}"]) <= VAL_TOL * max(1.0, abs(gold[f"dis_real_{tag}"])), tag
            assert abs(f.item() - gold[f"dis_fake_{tag}"]) <= VAL_TOL * max(1.0, abs(gold[f"dis_fake_{tag}"])), tag
            assert np.abs(_cat_grads(real, True) - gold[f"dis_{tag}_grad_real"]).max() <= GRAD_TOL, tag
            assert np.abs(_cat_grads(fake, True) - gold[f"dis_{tag}_grad_fake"]).max() <= GRAD_TOL, tag
    real = [[t.to(dev) for t in o] for o in synth.adv_logits(seed)]
    for al in (True, False):
        for ad in (True, False):
            for fin in (True, False):
                tag = f"{int(al)}{int(ad)}{int(fin)}"
                fake = fresh(seed + 1)
                fm = fns["fm"](fake, real, al, ad, fin)
                fm.backward()
                assert abs(fm.item() - gold[f"fm_{tag}"]) <= VAL_TOL * max(1.0, abs(gold[f"fm_{tag}"])), tag
                assert np.abs(_cat_grads(fake, False) - gold[f"fm_{tag}_grad"]).max() <= GRAD_TOL, tag


def test_oracle_matches_reference_fixture():
    fns = dict(gen=torch_cpu.generator_adversarial_loss, dis=torch_cpu.discriminator_adversarial_loss,
               fm=torch_cpu.feature_match_loss)
    _check(load_golden("adv_losses"), fns, torch.device("cpu"))


@pytest.mark.gpu
def test_hip_losses_match_reference_fixture(device):
    from parallelwavegan_amd import losses

    fns = dict(
        gen=lambda o, avg, lt: losses.GeneratorAdversarialLoss(average_by_discriminators=avg, loss_type=lt)(o),
        dis=lambda oh, o, avg, lt: losses.DiscriminatorAdversarialLoss(average_by_discriminators=avg, loss_type=lt)(oh, o),
        fm=lambda fh, f, al, ad, fin: losses.FeatureMatchLoss(average_by_layers=al, average_by_discriminators=ad,
                                                              include_final_outputs=fin)(fh, f))
    _check(load_golden("adv_losses"), fns, device)


@pytest.mark.gpu
def test_single_tensor_outputs_and_many_items(device):
    """Plain-tensor discriminator outputs (PWG's single discriminator) and > 64 items per reduction."""
    from parallelwavegan_amd import functional as Fn
    from parallelwavegan_amd import losses

    x = torch.randn(3, 1, 1000, generator=torch.Generator().manual_seed(1))
    xd = x.to(device).requires_grad_()
    xc = x.clone().requires_grad_()
    for lt in ("mse", "hinge"):
        r, f = losses.DiscriminatorAdversarialLoss(loss_type=lt)(xd, xd)
        rr, fr = torch_cpu.discriminator_adversarial_loss([xc], [xc], True, lt)
        assert abs(r.item() - rr.item()) <= 1e-6 and abs(f.item() - fr.item()) <= 1e-6
    g = torch.Generator().manual_seed(2)
    ts = [torch.randn(int(n), generator=g) for n in torch.randint(1, 30000, (150,), generator=g)]
    us = [torch.randn_like(t) for t in ts]
    td = [t.to(device).requires_grad_() for t in ts]
    ud = [u.to(device) for u in us]
    spec, flat = [], []
    for i, (a, b) in enumerate(zip(td, ud)):
        spec.append(("abs_diff" if i % 2 else "sq_diff", 0.5 + 0.01 * i, 0.0, i % 3))
        flat += [a, b]
    out = Fn.MultiReduceFn.apply(spec, 3, *flat)
    ref = torch.zeros(3, dtype=torch.float64)
    for i, (a, b) in enumerate(zip(ts, us)):
        d = (a.double() - b.double())
        ref[i % 3] += (0.5 + 0.01 * i) * (d.abs().sum() if i % 2 else (d * d).sum())
    assert torch.allclose(out.cpu().double(), ref, rtol=2e-6)
    (out * torch.tensor([1.0, 2.0, 3.0], device=device)).sum().backward()
    for i, (a, t, u) in enumerate(zip(td, ts, us)):
        d = t - u
        w = (0.5 + 0.01 * i) * (i % 3 + 1)
        exp = w * (torch.sign(d) if i % 2 else 2 * d)
        assert torch.allclose(a.grad.cpu(), exp, rtol=1e-6, atol=1e-7), i
    # determinism: the same reduction twice gives bit-identical sums
    assert torch.equal(out, Fn.MultiReduceFn.apply(spec, 3, *flat))
