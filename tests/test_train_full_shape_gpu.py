"""Training parity at each BASELINE configuration's OWN batch shape (C5 added in round 4).

C2 = Parallel WaveGAN.v1 (B 6 x 25600, RAdam, clip 10 / 1, multi-resolution STFT loss), C3 = HiFi-GAN V1
(B 16 x 8192, MSD + MPD, mel + feature-matching loss), C4 = multi-band MelGAN.v2 (B 64 x 16384, PQMF, full-band +
sub-band STFT losses), C5 = BASELINE configs[4], HiFi-GAN V1 LibriTTS 24 kHz (B 16 x 8400, upsample scales 5/5/4/3, mel loss
2048 / 300 / 1200 -- reference egs/libritts/voc1/conf/hifigan.v1.yaml:40,100-102,128-129 -- the data-parallel
workload of ``bench.py --gpus N``): ONE ``Trainer._train_step`` of the unmodified reference at exactly these shapes is stored in
``tests/golden/c{2,3,4,5}_train_full.npz`` (made by ``tests/golden/make_golden.py``: every logged loss, every
parameter's first-moment norm and <first update, first moment>).  The tile / split-K / slab plans the HIP engine
chooses at these sizes (``splits745``, ``tiles64``, ``tt64`` ... in profiles/r02_train_shapes_*.txt) are the ones the
benchmark times; the B = 2 fixtures never reach them.

Both tests run with the LDS NaN-poisoned before every MFMA launch (a contraction that touches an unwritten tile
element turns non-finite) and with every ``torch.empty`` NaN-filled (a workspace element nobody wrote does too);
the second test also replays the captured hipGraph of the step and requires it to follow the eager run.
"""
import os
import tempfile

import numpy as np
import pytest
import torch
import yaml

from parallelwavegan_amd.bin.train import Trainer
from parallelwavegan_amd.utils import build_from_config
from tests.golden import synth
from tests.util import load_golden, poison_empty, poison_lds

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONF = {"c2": "parallel_wavegan.v1", "c3": "hifigan.v1", "c4": "multi_band_melgan.v2", "c5": "hifigan.v1.libritts"}
SCALES = {"c2": (synth.PWG_G_SCALE, 1.4), "c3": (1.25, 1.0), "c4": (synth.MELGAN_G_SCALE, 1.2), "c5": (1.25, 1.0)}
# loss bars: 2e-4 relative (two CPU runs of the reference differ by ~5e-6 .. 1e-5) ...
LOSS_TOL = 2e-4
# ... except the spectral-convergence losses at these sizes.  The reference forms ||Y| - |X||_F / ||Y||_F with fp32
# torch.norm over 0.66 M (C2) to 13 M (C4) magnitudes; that accumulation alone is 4e-4 .. 6e-4 away from exact
# arithmetic at C4's sizes (measured here against float64, identical for 3 and 8 threads: torch's reduction order
# does not depend on the thread count), the ratio of the two norms 1e-5 .. 2e-4.  The HIP kernel sums 32 x 32 tiles
# and then the tiles, which is closer to exact.  So these losses get 5e-4 against the reference's fp32 value AND
# 2e-5 against the float64 evaluation of the same formula on the reference's generator output (fixture key sc64/*).
SC_TOL_FP32, SC_TOL_FP64 = 5e-4, 2e-5
SC64 = {"train/spectral_convergence_loss": "sc64/full", "train/sub_spectral_convergence_loss": "sc64/sub"}
# C4's fake loss is evaluated on the generator AFTER its first Adam step (lr 1e-3, which turns the rounding noise of
# near-zero gradient entries into +-lr steps, DESIGN s4): two runs of the REFERENCE at this shape (3 vs 8 threads,
# GOLDEN_THREADS=3 tests/golden/make_golden.py c4_train_full) give 0.035807 vs 0.035799 = 2.2e-4 apart; every other
# value of that pair agrees to <= 7e-6.  Bar = 3x the reference's own spread.
LOOSE = {("c4", "train/fake_loss"): 7e-4}
# Per-tensor first-moment norms (= (1 - beta1) |gradient| after one step) and <first update, first moment>.  Round 4:
# bars set per configuration at about 5x the measured deviation instead of one 3e-3 / 5e-3 for all ("would
# not catch a 0.2 % wgrad error in one layer"): measured worst tensors C2 2.5e-5, C3 1.4e-4, C5 2.0e-4.  C4 keeps 3e-3:
# its worst tensor (the generator's first convolution) is 2.7e-3 away, driven by the reference's own fp32
# spectral-convergence accumulation (see SC_TOL above); its discriminator is at 6e-4.
MOM_TOL = {"c2": 2e-4, "c3": 1e-3, "c5": 1e-3, "c4": 3e-3}
UPD_TOL = {"c2": 3e-4, "c3": 8e-4, "c5": 8e-4, "c4": 8e-3}  # measured 4.6e-5 / 1.1e-4 / 1.5e-4 / 4.3e-3 (C4: as above)
# Round 5 ("a 0.5 % wgrad error in C4's generator would pass"): C4's wide bars now apply to ONE tensor, the
# generator's first convolution `melgan.1.weight_v` (2.65e-3 / 4.16e-3 away: it sits right behind the spectral losses,
# whose fp32 accumulation in the reference is what it inherits); every other generator tensor is within 9.0e-4 /
# 1.1e-3 and every discriminator tensor within 6.7e-4 / 6.0e-4 of the reference, and is held to ~2x that.
TENSOR_TOL = {"c4": {"generator": {"default": (2e-3, 2.5e-3), "melgan.1.weight_v": (3e-3, 8e-3)},
                     "discriminator": {"default": (1.5e-3, 1.5e-3)}}}


def _bars(tag, key, names, which):
    """Per-tensor bars (which = 0: first-moment norms, 1: <update, moment>)."""
    per = TENSOR_TOL.get(tag, {}).get(key)
    if per is None:
        return np.full(len(names), (MOM_TOL, UPD_TOL)[which][tag])
    return np.array([per.get(n, per["default"])[which] for n in names])

def _build(tag, gold, dev, **overrides):
    with open(os.path.join(ROOT, "tests", "fixtures", "conf", CONF[tag] + ".yaml")) as f:
        conf = yaml.load(f, Loader=yaml.Loader)
    b, t, seed = (int(v) for v in gold["meta"])
    assert (b, t) == (conf["batch_size"], conf["batch_max_steps"])  # the recipe's own batch shape
    model, criterion, opt, sched = build_from_config(conf, dev)
    gs, ds = SCALES[tag]
    g, d = model["generator"], model["discriminator"]
    g.load_state_dict(synth.synth_state_dict(g.state_dict(), seed=seed, g_scale=gs))
    d.load_state_dict(synth.synth_state_dict(d.state_dict(), seed=seed + 1, g_scale=ds))
    conf.update(generator_train_start_steps=0, discriminator_train_start_steps=0, train_max_steps=10 ** 9,
                save_interval_steps=10 ** 9, eval_interval_steps=10 ** 9, log_interval_steps=10 ** 9, distributed=False,
                rank=0, outdir=tempfile.mkdtemp(), progress=False, record_loss_history=True)
    conf.update(overrides)
    acw = conf["generator_params"].get("aux_context_window", 0)
    c = synth.synth_input("c", (b, conf["num_mels"], t // conf["hop_size"] + 2 * acw), seed=seed).to(dev)
    y = (0.5 * synth.synth_input("y", (b, 1, t), seed=seed)).to(dev)
    x = (synth.synth_input("z", (b, 1, t), seed=seed).to(dev), c) if tag == "c2" else (c,)
    batch = (x, y)
    tr = Trainer(steps=1, epochs=0, data_loader={"train": [batch], "dev": [batch]}, sampler={"train": None, "dev": None},
                 model=model, criterion=criterion, optimizer=opt, scheduler=sched, config=conf, device=dev)
    tr.tqdm = None
    return tr, batch, model, opt


@pytest.mark.parametrize("tag", ["c2", "c3", "c4", "c5"])
def test_one_step_at_the_baseline_batch_shape_matches_the_reference(device, tag):
    gold = load_golden(f"{tag}_train_full")
    with poison_lds(), poison_empty():
        tr, batch, model, opt = _build(tag, gold, device)
        p0 = {key: {n: p.detach().clone() for n, p in model[key].named_parameters()} for key in model}
        tr._train_step(batch)
        torch.cuda.synchronize()
    (step, losses), = tr.loss_history()
    for k, v in losses.items():
        want = float(gold[f"step0/{k}"])
        rel = abs(v - want) / max(abs(want), 1e-3)
        print(f"[full-shape {tag}] {k}: got {v:.7g} want {want:.7g} rel {rel:.2e}")
        if k in SC64:
            exact = float(gold[SC64[k]])
            rel64 = abs(v - exact) / exact
            print(f"[full-shape {tag}] {k}: float64 evaluation {exact:.7g} rel {rel64:.2e} "
                  f"(the reference's fp32 value is {abs(want - exact) / exact:.2e} away)")
            assert np.isfinite(v) and rel <= SC_TOL_FP32 and rel64 <= SC_TOL_FP64, (tag, k, v, want, exact)
            continue
        tol = LOOSE.get((tag, k), LOSS_TOL)
        if k == "train/generator_loss" and any(n in losses for n in SC64):
            tol = SC_TOL_FP32  # (contains the spectral-convergence terms)
        assert np.isfinite(v) and rel <= tol, (tag, k, v, want)
    assert {f"step0/{k}" for k in losses} == {k for k in gold if k.startswith("step0/")}
    for key in ("generator", "discriminator"):
        names = {p: n for n, p in model[key].named_parameters()}
        gn = [str(n) for n in gold[f"momnorm_names/{key}"]]
        norms = {names[p]: float(s["exp_avg"].double().norm()) for p, s in opt[key].state.items()}
        assert sorted(norms) == gn
        got, want = np.array([norms[n] for n in gn]), gold[f"momnorm/{key}"]
        assert np.isfinite(got).all(), (tag, key)
        rel = np.abs(got - want) / (np.abs(want) + 1e-3 * np.abs(want).max())
        top = np.argsort(-rel)[:4]
        print(f"[full-shape {tag}] {key}: worst first-moment norm {gn[int(rel.argmax())]} rel {rel.max():.2e}"
              f"  (next: {', '.join('%s %.1e' % (gn[int(i)], rel[int(i)]) for i in top[1:])})")
        over = rel > _bars(tag, key, gn, 0)
        assert not over.any(), (tag, key, [(gn[int(i)], float(rel[int(i)])) for i in np.nonzero(over)[0]])
        dots = {names[p]: float(((p.detach() - p0[key][names[p]]).double() * s["exp_avg"].double()).sum())
                for p, s in opt[key].state.items()}
        got, want = np.array([dots[n] for n in gn]), gold[f"upddot/{key}"]
        rel = np.abs(got - want) / (np.abs(want) + 1e-3 * np.abs(want).max())
        top = np.argsort(-rel)[:4]
        print(f"[full-shape {tag}] {key}: worst <update, moment> {gn[int(rel.argmax())]} rel {rel.max():.2e}"
              f"  (next: {', '.join('%s %.1e' % (gn[int(i)], rel[int(i)]) for i in top[1:])})")
        over = rel > _bars(tag, key, gn, 1)
        assert not over.any(), (tag, key, "upddot", [(gn[int(i)], float(rel[int(i)])) for i in np.nonzero(over)[0]])


def _six_steps(tag, gold, device, **cfg):
    with poison_lds(), poison_empty():
        tr, batch, model, opt = _build(tag, gold, device, **cfg)
        for _ in range(6):
            tr._train_step(batch)
        torch.cuda.synchronize()
        n_graphs = len(tr._graphs)
        hist = tr.loss_history()
    del tr, model, opt
    torch.cuda.empty_cache()
    return hist, n_graphs


def _worst(ha, hb):
    """Per step: (largest relative deviation, its loss name) between two loss histories."""
    out = []
    for (sa, a), (sb, b) in zip(ha, hb):
        assert sa == sb and sorted(a) == sorted(b)
        for k in a:
            assert np.isfinite(a[k]) and np.isfinite(b[k]), (sa, k, a[k], b[k])
        k = max(a, key=lambda k_: abs(a[k_] - b[k_]) / max(abs(a[k_]), 1e-3))
        out.append((abs(a[k] - b[k]) / max(abs(a[k]), 1e-3), k.split("/")[-1]))
    return out


# plain-eager vs replay.  Until round 4 the three fp32-atomic reductions made Adam / RAdam amplify last-bit noise into
# percent-level drift by step 5 (bar 1e-1).  With ordered reductions -- and, since the end of round 5, the same launch
# plans in both modes (the concurrency hint of the captured step defaults to 1.0: planning for the whole chip measured
# faster) -- replay and plain eager agree BIT FOR BIT over all six steps at every configuration
# (profiles/r05_graph_eq_eager.txt; while HiFi-GAN's captured step planned with hint 0.5, i.e. other split-K / tile
# plans and another summation order, the two stayed within 1.3e-5).  A run with `conv_concurrency_hint` != 1 would need
# a tolerance here.
PLAIN_TOL = {"c2": 0.0, "c3": 0.0, "c4": 0.0, "c5": 0.0}


@pytest.mark.parametrize("tag", ["c2", "c3", "c4", "c5"])
def test_graph_replay_at_the_baseline_batch_shape_follows_eager(device, tag):
    """6 steps at the recipe's own batch shape, with poisoned LDS and NaN-filled ``torch.empty`` (the fills are part of
    the graph), three ways:
      A  eager launches under the graph mode's launch plans (``graph_warmup_steps`` > 6: never captured),
      B  2 eager steps, capture, 3 replays,
      C  plain eager (serial sub-networks, default plans).
    Since round 5 no kernel of the step uses floating-point atomics, so A and B -- same kernels, same plans, same
    inputs -- must agree BIT FOR BIT on every loss of all six steps (a loose bar cannot catch a
    stale pointer or a missed node), and so must a second run of A (run-to-run determinism).  B against C keeps a
    equality as well (same plans since the captured step's concurrency hint defaults to 1.0)."""
    gold = load_golden(f"{tag}_train_full")
    eager_plans, n0 = _six_steps(tag, gold, device, use_hip_graph=True, graph_warmup_steps=100)
    replay, n1 = _six_steps(tag, gold, device, use_hip_graph=True, graph_warmup_steps=2)
    again, _ = _six_steps(tag, gold, device, use_hip_graph=True, graph_warmup_steps=100)
    plain, n2 = _six_steps(tag, gold, device, use_hip_graph=False)
    assert (n0, n1, n2) == (0, 1, 0)
    assert len(eager_plans) == len(replay) == len(plain) == 6
    w_run = _worst(eager_plans, again)
    w_cap = _worst(eager_plans, replay)
    w_plain = _worst(plain, replay)
    print(f"[graph==eager {tag}] run-to-run {['%.1e' % w for w, _ in w_run]}  capture {['%.1e' % w for w, _ in w_cap]}  "
          f"plain-vs-replay {['%.1e:%s' % w for w in w_plain]}")
    assert all(w == 0.0 for w, _ in w_run), (tag, "two eager runs differ", w_run)
    assert all(w == 0.0 for w, _ in w_cap), (tag, "replay differs from eager launches of the same plans", w_cap)
    for i, (w, k) in enumerate(w_plain):
        assert w <= PLAIN_TOL[tag], (tag, i, k, w)
