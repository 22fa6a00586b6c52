"""GPU parity of the fused optimizers against torch.optim.Adam (CPU) and the RAdam restatement."""
import pytest
import torch

from oracle import radam as oracle_radam
from parallelwavegan_amd import optimizers

pytestmark = pytest.mark.gpu


def _params(seed, device):
    g = torch.Generator().manual_seed(seed)
    shapes = [(70000,), (64, 32, 7), (1,), (513, 80), (3,)]
    cpu = [torch.randn(s, generator=g).requires_grad_() for s in shapes]
    dev = [p.detach().clone().to(device).requires_grad_() for p in cpu]
    return cpu, dev, g


@pytest.mark.parametrize("kw", [dict(lr=2e-4, betas=(0.5, 0.9)), dict(lr=1e-3, eps=1e-7, amsgrad=True),
                                dict(lr=1e-3, weight_decay=0.01)])
def test_fused_adam_matches_torch_adam(kw, device):
    cpu, dev, g = _params(1, device)
    ref = torch.optim.Adam(cpu, **kw)
    opt = optimizers.Adam(dev, **kw)
    for step in range(5):
        for pc, pd in zip(cpu, dev):
            grad = torch.randn(pc.shape, generator=g) * (10.0 if step == 2 else 0.1)
            pc.grad = grad.clone()
            pd.grad = grad.to(device)
        ref.step()
        opt.step()
    for pc, pd in zip(cpu, dev):
        assert (pc.detach() - pd.detach().cpu()).abs().max().item() <= 2e-6 * (1 + pc.detach().abs().max().item())
    sd, sr = opt.state_dict(), ref.state_dict()
    assert sd["state"].keys() == sr["state"].keys()
    for k in sr["state"]:
        assert set(sd["state"][k]) == set(sr["state"][k])  # step / exp_avg / exp_avg_sq (/ max_exp_avg_sq)


def test_fused_radam_matches_reference_update_rule(device):
    cpu, dev, g = _params(2, device)
    opt = optimizers.RAdam(dev, lr=1e-4, eps=1e-6, weight_decay=0.0)
    m = [torch.zeros_like(p) for p in cpu]
    v = [torch.zeros_like(p) for p in cpu]
    with torch.no_grad():
        for step in range(1, 9):  # crosses the N_sma >= 5 rectification threshold (step 6 for beta2 = 0.999)
            for i, (pc, pd) in enumerate(zip(cpu, dev)):
                grad = torch.randn(pc.shape, generator=g)
                pd.grad = grad.to(device)
                oracle_radam.radam_step(pc, grad, m[i], v[i], step, lr=1e-4, eps=1e-6)
            opt.step()
    for pc, pd in zip(cpu, dev):
        assert (pc.detach() - pd.detach().cpu()).abs().max().item() <= 2e-6 * (1 + pc.detach().abs().max().item())


def test_clip_grad_norm_matches_torch(device):
    cpu, dev, g = _params(3, device)
    for pc, pd in zip(cpu, dev):
        grad = torch.randn(pc.shape, generator=g)
        pc.grad, pd.grad = grad.clone(), grad.to(device)
    total = torch.nn.utils.clip_grad_norm_(cpu, 10.0)
    out = optimizers.clip_grad_norm_([(p, p.grad) for p in dev], 10.0)
    assert abs(out[0].item() - total.item()) <= 1e-5 * total.item()
    for pc, pd in zip(cpu, dev):
        assert (pc.grad - pd.grad.cpu()).abs().max().item() <= 1e-6


@pytest.mark.parametrize("kw", [dict(lr=2e-4, betas=(0.8, 0.99), weight_decay=0.0), dict(lr=1e-3, weight_decay=0.05),
                                dict(lr=1e-3, amsgrad=True)])
def test_fused_adamw_matches_torch_adamw(kw, device):
    """`generator_optimizer_type: AdamW` (egs/yesno/voc1/conf/*.v1.debug.yaml of the reference): decoupled weight decay,
    torch's default 1e-2, same state-dict layout."""
    cpu, dev, g = _params(4, device)
    ref = torch.optim.AdamW(cpu, **kw)
    opt = optimizers.AdamW(dev, **kw)
    for step in range(5):
        for pc, pd in zip(cpu, dev):
            grad = torch.randn(pc.shape, generator=g) * (10.0 if step == 2 else 0.1)
            pc.grad = grad.clone()
            pd.grad = grad.to(device)
        ref.step()
        opt.step()
    for pc, pd in zip(cpu, dev):
        assert (pc.detach() - pd.detach().cpu()).abs().max().item() <= 2e-6 * (1 + pc.detach().abs().max().item())
    sd, sr = opt.state_dict(), ref.state_dict()
    assert sd["state"].keys() == sr["state"].keys()
    for k in sr["state"]:
        assert set(sd["state"][k]) == set(sr["state"][k])
    assert opt.param_groups[0]["weight_decay"] == ref.param_groups[0]["weight_decay"]
