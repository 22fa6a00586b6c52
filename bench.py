#!/usr/bin/env python
"""bench.py -- headline benchmark: HiFi-GAN V1 (22.05 kHz) generator inference RTF^-1
(audio samples / second) on MI355X, the training steps/s of BASELINE configs C2 / C3 / C4,
and the same-box CPU baseline.

Contract (one JSON line on stdout from rank 0):
  python bench.py --gpus N --steps K --warmup W
A "step" is one forward pass of the generator over one resident batch of
synthetic mel frames (B utterances x F frames, mel ~ N(0,1), random-init weights
of the V1 architecture, weight norm removed as bin/decode.py does).  N > 1 runs
one replica per GPU on its own batch (utterances are independent: "replicas
only", no data-path collective) -> weak scaling; value = total samples / max
time over ranks.  The training measurements shard minibatches (data parallel,
RCCL gradient all-reduce).

``--gpus N`` with N > 1 and no launcher environment (WORLD_SIZE unset) re-launches this
script once per GPU through parallelwavegan_amd.distributed.launch; under
``python -m torch.distributed.run`` (the driver's way) the ranks are used as given.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

FP32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
HBM_PEAK_GBS = 8000.0
CONF_DIR = os.path.join(ROOT, "tests", "fixtures", "conf")
# BASELINE.json configs[1..3] -> the reference recipe each is quoted on (tests/fixtures/make_conf.py)
# (+ configs[4] = c5: HiFi-GAN V1 LibriTTS 24 kHz, egs/libritts/voc1/conf/hifigan.v1.yaml -- the data-parallel workload)
TRAIN_CONFIGS = {"c2": "parallel_wavegan.v1", "c3": "hifigan.v1", "c4": "multi_band_melgan.v2",
                 "c5": "hifigan.v1.libritts"}
CORPUS = {"c2": "ljspeech", "c3": "ljspeech", "c4": "ljspeech", "c5": "libritts"}
# fwd = 1x, bwd = 2x, incl. the second no-grad G pass and the no-grad D(real) pass:
# per item 4 * G(32 frames) + 10 * D(8192 samples)  (what the REFERENCE executes per C3 step)
C3_REFERENCE_GFLOP_PER_ITEM = 4 * 19.65 + 10 * 12.08


def _fold_batch_on():
    """Is batch folding of weight-heavy, few-column layers active (layers/conv.py, PWG_FOLD_BATCH)?"""
    from parallelwavegan_amd.layers.conv import _ConvNd

    return _ConvNd.fold_batch


def load_conf(name):
    import yaml

    with open(os.path.join(CONF_DIR, name + ".yaml")) as f:
        return yaml.load(f, Loader=yaml.Loader)


def hifigan_macs_per_sample(cfg):
    """Algorithmic multiply-accumulates per output sample (1 199 424)."""
    up = 1
    for s in cfg["upsample_scales"]:
        up *= s
    ch = cfg["channels"]
    macs = cfg["in_channels"] * ch * cfg["kernel_size"] / up  # input conv runs at the frame rate
    rate = 1.0 / up
    for s, k in zip(cfg["upsample_scales"], cfg["upsample_kernel_sizes"]):
        macs += ch * (ch // 2) * k * rate  # ConvTranspose1d: Cin*Cout*k per INPUT sample
        ch //= 2
        rate *= s
        for ks, dil in zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilations"]):
            n_convs = len(dil) * (2 if cfg["use_additional_convs"] else 1)
            macs += n_convs * ch * ch * ks * rate
    macs += ch * cfg["out_channels"] * cfg["kernel_size"] * rate
    return macs


# ------------------------------------------------------------------------------------------------
# CPU baselines (the oracle = torch-CPU restatement of the reference's ATen call sequence; the
# reference package itself does not exist on the GPU box, hence kind "port")
# ------------------------------------------------------------------------------------------------
def cpu_baseline(g_params, budget_s=(6.0, 14.0)):
    """B=1 per call (bin/decode.py is utterance-at-a-time), at 100 AND 800 mel frames; ``value`` is
    the 800-frame rate -- the utterance length of the GPU workload."""
    from oracle import ref_run, torch_cpu
    from parallelwavegan_amd.models import HiFiGANGenerator

    cores = os.cpu_count() or 1
    use_ref = ref_run.available()  # the reference's own package: /root/reference or the staged copy oracle/_ref
    if use_ref:
        g_ref = ref_run.generator("HiFiGANGenerator", g_params)  # prepared as bin/decode.py:141-149 does
    else:
        g = HiFiGANGenerator(**g_params)
        g.remove_weight_norm()
        sd = {k: v.detach() for k, v in g.state_dict().items()}

    def once(c):
        t0 = time.time()
        y = g_ref(c) if use_ref else torch_cpu.hifigan_generator(sd, c, **g_params)
        return time.time() - t0, y

    per_len = {}
    with torch.no_grad():
        # torch's default (= all cores) oversubscribes these small convs on a many-core host, so
        # probe a few intra-op thread counts briefly and keep the fastest for the timed samples
        c100 = torch.randn(1, 80, 100)
        probe = {}
        for nt in sorted({min(cores, n) for n in (8, 16, 32, 64)}):
            torch.set_num_threads(nt)
            once(c100)
            probe[nt] = min(once(c100)[0], once(c100)[0])
        nthreads = min(probe, key=probe.get)
        torch.set_num_threads(nthreads)
        for frames, budget in zip((100, 800), budget_s):
            c = torch.randn(1, 80, frames)
            best, n, t_start = float("inf"), 0, time.time()
            while n < 3 or (time.time() - t_start < budget and n < 200):
                dt, y = once(c)
                best = min(best, dt)
                n += 1
            per_len[frames] = {"samples_per_s": y.numel() / best, "best_s": best, "calls": n,
                               "rtf": best / (y.numel() / 22050.0)}
    return {
        "value": per_len[800]["samples_per_s"],
        "unit": "samples/s",
        "cores": torch.get_num_threads(),
        "kind": "reference" if use_ref else "port",
        "frames_100": per_len[100],
        "frames_800": per_len[800],
        "sample": ("parallel_wavegan.models.HiFiGANGenerator of the UNMODIFIED reference package (staged byte for byte "
                   "by oracle/make_ref.py as oracle/_ref, weight norm removed as bin/decode.py does), " if use_ref else
                   "oracle.torch_cpu.hifigan_generator (torch-CPU restatement of the reference's ATen sequence, "
                   "pinned to reference fixtures; the staged reference package oracle/_ref is absent), ") +
                  f"B=1 x 800 frames, best of {per_len[800]['calls']} calls (100 frames: best of {per_len[100]['calls']}), "
                  f"{nthreads} of {cores} host threads (fastest of {sorted(probe)})",
    }


def cpu_train_baseline(conf, budget_steps=1, batch=None):
    """The reference's own ``Trainer._train_step`` (bin/train.py:189-340; staged copy oracle/_ref) -- or, when that is
    absent, oracle.train_step, its torch-CPU restatement -- at the recipe's OWN batch (B = 16 x 8192 for C3): one
    warm-up step, then ``budget_steps`` timed steps (~9 s each on the GPU boxes' hosts)."""
    from oracle import ref_run

    cores = os.cpu_count() or 1
    nthreads = min(cores, 32)
    torch.set_num_threads(nthreads)
    b = batch or conf["batch_size"]
    t = conf["batch_max_steps"]
    gen = torch.Generator(device="cpu").manual_seed(77)
    c, y = torch.randn(b, 80, t // conf["hop_size"], generator=gen), 0.3 * torch.randn(b, 1, t, generator=gen)
    use_ref = ref_run.available()
    if use_ref:
        tr = ref_run.trainer(conf, ((c,), y))

        def step():
            tr._train_step(((c,), y))
    else:
        from oracle.train_step import HiFiGANTrainState
        from parallelwavegan_amd.models import HiFiGANGenerator, HiFiGANMultiScaleMultiPeriodDiscriminator

        g = HiFiGANGenerator(**conf["generator_params"])
        d = HiFiGANMultiScaleMultiPeriodDiscriminator(**conf["discriminator_params"])
        st = HiFiGANTrainState({k: v.detach() for k, v in g.state_dict().items()},
                               {k: v.detach() for k, v in d.state_dict().items()}, conf["generator_params"],
                               conf["discriminator_params"], conf["mel_loss_params"])

        def step():
            st.step(c, y)
    step()  # warm-up
    t0 = time.time()
    for _ in range(budget_steps):
        step()
    dt = (time.time() - t0) / budget_steps
    return {
        "value": 1.0 / dt,
        "unit": f"steps/s at B={b} x {t} (the recipe's batch, run as is)",
        "batch": b,
        "cores": nthreads,
        "kind": "reference" if use_ref else "port",
        "sample": ("parallel_wavegan.bin.train.Trainer._train_step of the unmodified reference package (oracle/_ref), "
                   if use_ref else
                   "oracle.train_step.HiFiGANTrainState.step (restatement of the reference Trainer._train_step), ") +
                  f"B={b} x {t} samples, 1 warm-up + {budget_steps} timed steps ({dt:.2f} s each), "
                  f"{nthreads} of {cores} host threads",
    }


# ------------------------------------------------------------------------------------------------
# training steps/s of one BASELINE config
# ------------------------------------------------------------------------------------------------
def synthetic_batch(conf, batch, dev, rank):
    gen = torch.Generator(device="cpu").manual_seed(200 + rank)
    t, hop = conf["batch_max_steps"], conf["hop_size"]
    gp = conf["generator_params"]
    acw = gp.get("aux_context_window", 0)
    c = torch.randn(batch, conf["num_mels"], t // hop + 2 * acw, generator=gen).to(dev)
    y = (0.3 * torch.randn(batch, 1, t, generator=gen)).to(dev)
    if conf.get("generator_type", "ParallelWaveGANGenerator") == "ParallelWaveGANGenerator":
        return ((torch.randn(batch, 1, t, generator=gen).to(dev), c), y)  # Collater(use_noise_input=True)
    return ((c,), y)


def bench_train(args, tag, dev, rank, world, dist, steps, warmup, detail=True):
    """``steps`` timed optimisation steps (generator AND discriminator phase active) of the recipe
    ``TRAIN_CONFIGS[tag]`` at its own batch size / segment length per GPU, everything built by
    ``build_from_config`` from the reference YAML."""
    import tempfile

    from parallelwavegan_amd import ops
    from parallelwavegan_amd.bin.train import Trainer
    from parallelwavegan_amd.utils import build_from_config

    name = TRAIN_CONFIGS[tag]
    conf = load_conf(name)
    torch.manual_seed(4321)
    model, criterion, opt, sched = build_from_config(conf, dev)
    b = args.train_batch if (tag == "c3" and args.train_batch) else conf["batch_size"]
    # both phases active from the first step (steady state of the recipe after discriminator_train_start_steps)
    conf.update(generator_train_start_steps=0, discriminator_train_start_steps=0, train_max_steps=10 ** 9,
                save_interval_steps=10 ** 9, eval_interval_steps=10 ** 9, log_interval_steps=10 ** 9,
                distributed=world > 1 or os.environ.get("PWG_FORCE_DIST") == "1", rank=rank,
                outdir=tempfile.mkdtemp(), progress=False,
                use_hip_graph=not args.no_graph, graph_warmup_steps=2, record_loss_history=True,
                ddp_grad_groups=int(os.environ.get("PWG_DDP_GROUPS", "3")),
                reuse_real_discriminator_pass=os.environ.get("PWG_REUSE_REAL", "1") == "1")
    batch = synthetic_batch(conf, b, dev, rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n):
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            tr._train_step(batch)
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = tt.item()
        return el

    # hipGraph replay of the step (data parallel: graphs cut at the gradient exchanges); should the
    # capture fail on some stack, every rank sees the same exception and the run continues eagerly
    for use_graph in ([True, False] if conf["use_hip_graph"] else [False]):
        conf["use_hip_graph"] = use_graph
        tr = Trainer(steps=1, epochs=0, data_loader={"train": [batch], "dev": [batch]},
                     sampler={"train": None, "dev": None}, model=model, criterion=criterion, optimizer=opt,
                     scheduler=sched, config=conf, device=dev)
        tr.tqdm = None
        if os.environ.get("PWG_DBG_STFT"):  # debugging aid (tools/debug_bench_train.py)
            from tools.debug_bench_train import install_stft_debug

            install_stft_debug(tr)
        try:
            for _ in range(warmup):
                tr._train_step(batch)
            break
        except Exception as e:  # noqa: BLE001
            if not use_graph:
                raise
            print(f"[bench] hipGraph training step failed ({type(e).__name__}: {e}); falling back to eager",
                  file=sys.stderr, flush=True)
            for r in (tr.reducers or {}).values():
                r.remove()
    elapsed = timed(steps)
    # exposed communication = step time with the gradient all-reduces minus step time without them
    # (same graphs, same kernels; the ranks' parameters drift apart afterwards, which timing ignores)
    dist_info = None
    if tr.reducers:
        # the same steps again with all collectives off, and with only one of the two exchanges off: the
        # exposed (not overlapped) time of the generator's and of the discriminator's exchange separately
        def timed_without(keys):
            for k in keys:
                tr.reducers[k].skip_comm = True
            t = timed(steps)
            for k in keys:
                tr.reducers[k].skip_comm = False
            return t

        t_nocomm = timed_without(list(tr.reducers))
        t_without = {k: timed_without([k]) for k in tr.reducers}
        dist_info = {
            "backend": dist.get_backend(),
            "world_size": dist.get_world_size(),
            "exchanged_MB_per_step": {k: round(r.bytes / 1e6, 1) for k, r in tr.reducers.items()},
            "exchange_groups": {k: len(r.groups) for k, r in tr.reducers.items()},
            "buckets": {k: len(r.buckets) for k, r in tr.reducers.items()},
            "hook_copies_total": {k: r.copies for k, r in tr.reducers.items()},
            "slots_zeroed_total": {k: r.zero_fills for k, r in tr.reducers.items()},
            "ms_per_step_without_collectives": t_nocomm / steps * 1e3,
            "ms_per_step_without_exchange_of": {k: v / steps * 1e3 for k, v in t_without.items()},
            "rccl": rccl_log_summary(),
        }
    tr._flush_pending()
    # every loss of every step (warm-up included) was kept on the device (Trainer.loss_history): say WHICH loss
    # went non-finite at WHICH step instead of a bare flag
    hist = tr.loss_history()
    first_bad = None
    for st, losses in hist:
        bad = [k for k, v in losses.items() if not (v == v and abs(v) != float("inf"))]
        if bad:
            first_bad = {"step": st, "losses": bad, "values": {k: repr(losses[k]) for k in bad}}
            break
    finite = first_bad is None and all(v == v and abs(v) != float("inf") for v in tr.total_train_loss.values())
    last_losses = {k.split("/")[-1]: round(v, 6) for k, v in hist[-1][1].items()} if hist else None
    # one more (eager) step with per-kernel event timing; every rank runs it -- a data-parallel step
    # contains collectives -- but only rank 0 reports
    prof = None
    if detail:
        tr.config["use_hip_graph"] = False
        for m in model.values():  # serial launches: concurrent branches would inflate each kernel's event time
            for sub in m.modules():
                if hasattr(sub, "branch_streams"):
                    sub.branch_streams = False
        with ops.profile() as prof:
            tr._train_step(batch)
    out = None
    if rank == 0:
        ms = elapsed / steps * 1e3
        out = {
            "config": f"{tag}: {name} (egs/{CORPUS[tag]}/voc1/conf), B={b} x {conf['batch_max_steps']} samples per GPU, "
                      f"generator + discriminator phase, {conf.get('generator_optimizer_type', 'RAdam')}",
            # a step that produced a non-finite loss is not a measurement: no value
            "value": steps / elapsed if finite else None,
            "value_of_the_invalid_run": None if finite else steps / elapsed,
            "unit": "steps/s",
            "ms_per_step": ms,
            "steps": steps,
            "warmup": warmup,
            "batch_per_gpu": b,
            "global_batch": b * world,
            "segments_per_s": b * world * steps / elapsed,
            "scaling": "weak",
            "parallelism": f"dp{world}" if world > 1 else "single",
            "hip_graph": bool(tr._graphs),
            "fold_batch": bool(_fold_batch_on()),
            "losses_finite": finite,
            "first_nonfinite": first_bad,
            "steps_checked": len(hist),
            "last_step_losses": last_losses,
        }
        if dist_info is not None:
            dist_info["exposed_comm_ms"] = ms - dist_info["ms_per_step_without_collectives"]
            dist_info["exposed_comm_ms_per_exchange"] = {k: ms - v
                                                         for k, v in dist_info["ms_per_step_without_exchange_of"].items()}
            out["dist"] = dist_info
            out["exposed_comm_ms"] = dist_info["exposed_comm_ms"]
        if prof is not None:
            tot_ms = sum(v["ms"] for v in prof.results.values())
            executed = sum(v["flops"] for v in prof.results.values())
            mfma_ms = sum(v["ms"] for v in prof.results.values() if v["flops"])
            out["kernel_time_ms_per_step"] = round(tot_ms, 3)
            out["conv_launches_per_step"] = int(sum(v["launches"] for k, v in prof.results.items() if "conv1d" in k))
            out["launches_per_step"] = int(sum(v["launches"] for v in prof.results.values()))
            # EXECUTED flops: what the MFMA kernels of one step really computed (the engine skips the
            # discriminator weight gradients of the generator phase and the second D(real) pass)
            out["executed_TFLOP_per_step_per_gpu"] = executed / 1e12
            out["executed_TFLOPs_per_gpu"] = executed / (ms * 1e-3) / 1e12
            out["executed_frac_of_fp32_matrix_peak"] = executed / (ms * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS
            out["mfma_kernels_TFLOPs_while_running"] = executed / (mfma_ms * 1e-3) / 1e12 if mfma_ms else None
            if tag in ("c3", "c5"):
                # c5: the generator term from its own upsampling ladder, the discriminator term per sample as in c3
                t = conf["batch_max_steps"]
                ref = (4 * 2 * hifigan_macs_per_sample(conf["generator_params"]) * t + 10 * 12.08e9 * t / 8192) * b
                out["reference_TFLOP_per_step_per_gpu"] = ref / 1e12
                out["reference_flop_TFLOPs_per_gpu"] = ref / (ms * 1e-3) / 1e12
                out["reference_flop_frac_of_fp32_matrix_peak"] = ref / (ms * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS
            out["kernels"] = {
                k: {"ms_per_step": round(v["ms"], 3), "launches": v["launches"],
                    "TFLOPs": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["flops"] else None,
                    "GBps_algorithmic": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)}
                for k, v in sorted(prof.results.items(), key=lambda kv: -kv[1]["ms"])}
    for r in (tr.reducers or {}).values():
        r.remove()
    del tr, model, criterion, opt, sched
    torch.cuda.empty_cache()
    return out


def bench_pwg_inference(dev, steps=10, warmup=3, batch=16, frames=400):
    """BASELINE configs[0]/[1] generator: Parallel WaveGAN.v1 forward (noise + mel -> waveform), as a
    resident batch (graph replay) and as the single 80x100 mel of test_parallel_wavegan.py."""
    from parallelwavegan_amd import ops
    from parallelwavegan_amd.graphs import GraphedInference
    from parallelwavegan_amd.models import ParallelWaveGANGenerator

    conf = load_conf("parallel_wavegan.v1")
    gp = conf["generator_params"]
    torch.manual_seed(99)
    g = ParallelWaveGANGenerator(**gp)
    g.remove_weight_norm()
    g = g.to(dev).eval()
    acw, hop = gp["aux_context_window"], conf["hop_size"]
    out = {}
    for tag, (b, f, n) in {"batch": (batch, frames, steps), "B1_F100": (1, 100, 20)}.items():
        c = torch.randn(b, gp["aux_channels"], f + 2 * acw).to(dev)
        z = torch.randn(b, 1, f * hop).to(dev)
        run = GraphedInference(g)
        with torch.no_grad():
            for _ in range(warmup):
                y = run(z, c)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                y = run(z, c)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        assert torch.isfinite(y).all()
        out[tag] = {"batch": b, "frames": f, "ms": dt * 1e3, "samples_per_s": b * f * hop / dt}
        if tag == "batch":
            # parity of THIS workload (T = 102 400 per utterance): utterance B-1 of the timed batch against the oracle
            from oracle import torch_cpu

            sd = {k: v.detach().cpu() for k, v in g.state_dict().items()}
            torch.set_num_threads(min(os.cpu_count() or 1, 32))
            with torch.no_grad():
                ref = torch_cpu.pwg_generator(sd, z[b - 1:].cpu(), c[b - 1:].cpu(), **gp)
            err = (y[b - 1:].cpu() - ref).abs().max().item()
            out[tag]["parity"] = {"max_abs_vs_oracle": err, "utterances_checked": [b - 1], "tolerance": 1e-4,
                                  "oracle_abs_max": ref.abs().max().item(), "ok": err <= 1e-4}
            with ops.profile() as prof, torch.no_grad():
                g(z, c)
            fl = sum(v["flops"] for v in prof.results.values())
            out[tag]["MFLOP_per_sample_executed"] = fl / (b * f * hop) / 1e6
            out[tag]["TFLOPs"] = fl / dt / 1e12
            out[tag]["frac_of_fp32_matrix_peak"] = fl / dt / 1e12 / FP32_MATRIX_PEAK_TFLOPS
            out[tag]["kernels"] = {k: {"ms": round(v["ms"], 3), "launches": v["launches"],
                                       "TFLOPs": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["flops"] else None}
                                   for k, v in sorted(prof.results.items(), key=lambda kv: -kv[1]["ms"])}
        del run
    out["config"] = "c1/c2 generator: parallel_wavegan.v1, fp32, weight norm removed, hipGraph replay"
    del g
    torch.cuda.empty_cache()
    return out


class _MultiBandDecode(torch.nn.Module):
    """What ``MelGANGenerator.inference`` runs for a multi-band model (models/melgan.py:239-257): the generator, then
    ``pqmf.synthesis`` of its sub-band signals -- as a batched forward so that it can be captured like the others."""

    def __init__(self, g, pqmf):
        super().__init__()
        self.g, self.pqmf = g, pqmf

    def forward(self, c):
        return self.pqmf.synthesis(self.g(c))


def bench_mbmelgan_inference(dev, steps=10, warmup=3, batch=16, frames=800):
    """BASELINE configs[3]'s generator as ``bin/decode.py`` uses it: Multi-band MelGAN.v2 generator + PQMF synthesis
    (the one model whose published GPU decode figure maps onto a BASELINE config), as a resident batch
    and at batch 1 (100 / 800 frames), graph replay, with the timed batch's last utterance checked against the oracle."""
    from parallelwavegan_amd import ops
    from parallelwavegan_amd.graphs import GraphedInference
    from parallelwavegan_amd.layers import PQMF
    from parallelwavegan_amd.models import MelGANGenerator

    conf = load_conf("multi_band_melgan.v2")
    gp = conf["generator_params"]
    torch.manual_seed(77)
    g = MelGANGenerator(**gp)
    g.remove_weight_norm()
    model = _MultiBandDecode(g, PQMF(subbands=gp["out_channels"], **conf.get("pqmf_params", {}))).to(dev).eval()
    hop = conf["hop_size"]
    out = {}
    gen = torch.Generator(device="cpu").manual_seed(300)
    for tag, (b, f, n) in {"batch": (batch, frames, steps), "B1_F100": (1, 100, 20), "B1_F800": (1, 800, 20)}.items():
        c_host = torch.randn(b, gp["in_channels"], f, generator=gen)
        c = c_host.to(dev)
        run = GraphedInference(model)
        with torch.no_grad():
            for _ in range(warmup):
                y = run(c)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                y = run(c)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        assert torch.isfinite(y).all() and y.shape == (b, 1, f * hop), tuple(y.shape)
        out[tag] = {"batch": b, "frames": f, "ms": dt * 1e3, "samples_per_s": b * f * hop / dt,
                    "rtf": dt / (b * f * hop / float(conf["sampling_rate"]))}
        if tag == "batch":
            from oracle import torch_cpu

            sd = {k: v.detach().cpu() for k, v in g.state_dict().items()}
            torch.set_num_threads(min(os.cpu_count() or 1, 32))
            with torch.no_grad():
                ref = torch_cpu.pqmf_synthesis(torch_cpu.melgan_generator(sd, c_host[b - 1:], **gp),
                                               subbands=gp["out_channels"], **conf.get("pqmf_params", {}))
            err = (y[b - 1:].cpu() - ref).abs().max().item()
            out[tag]["parity"] = {"max_abs_vs_oracle": err, "utterances_checked": [b - 1], "tolerance": 1e-4,
                                  "oracle_abs_max": ref.abs().max().item(), "ok": err <= 1e-4}
            with ops.profile() as prof, torch.no_grad():
                model(c)
            fl = sum(v["flops"] for v in prof.results.values())
            by = sum(v["bytes"] for v in prof.results.values())
            out[tag]["MFLOP_per_sample_executed"] = fl / (b * f * hop) / 1e6
            out[tag]["TFLOPs"] = fl / dt / 1e12
            out[tag]["frac_of_fp32_matrix_peak"] = fl / dt / 1e12 / FP32_MATRIX_PEAK_TFLOPS
            out[tag]["algorithmic_GBps"] = by / dt / 1e9
            out[tag]["kernels"] = {k: {"ms": round(v["ms"], 3), "launches": v["launches"],
                                       "TFLOPs": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["flops"] else None,
                                       "GBps_algorithmic": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)}
                                   for k, v in sorted(prof.results.items(), key=lambda kv: -kv[1]["ms"])}
        del run
    out["config"] = ("c4 generator: multi_band_melgan.v2 + PQMF synthesis (MelGANGenerator.inference's work, "
                     "models/melgan.py:239-257), fp32, weight norm removed, hipGraph replay")
    del model, g
    torch.cuda.empty_cache()
    return out


def cpu_baseline_other_generators(budget_s=6.0):
    """The reference's own ``inference`` on the host cores for the two other generators SURVEY s8d names:
    Parallel WaveGAN.v1 on a random 80 x 100 mel (BASELINE configs[0]: the test_parallel_wavegan.py:157-198 path, a CPU
    run by definition) and Multi-band MelGAN.v2 (+ PQMF synthesis) at 100 and 800 frames.  ``kind`` = "reference" when
    the staged reference package is importable, else "port" (the oracle's restatement)."""
    from oracle import ref_run, torch_cpu

    cores = os.cpu_count() or 1
    use_ref = ref_run.available()
    res = {}

    def best_of(fn, budget):
        best, n, t_start = float("inf"), 0, time.time()
        while n < 3 or (time.time() - t_start < budget and n < 200):
            t0 = time.time()
            y = fn()
            best = min(best, time.time() - t0)
            n += 1
        return best, n, y

    def fastest_threads(fn):
        probe = {}
        for nt in sorted({min(cores, n) for n in (8, 16, 32, 64)}):
            torch.set_num_threads(nt)
            fn()
            probe[nt] = min(_time(fn), _time(fn))
        nt = min(probe, key=probe.get)
        torch.set_num_threads(nt)
        return nt

    def _time(fn):
        t0 = time.time()
        fn()
        return time.time() - t0

    with torch.no_grad():
        # ---- configs[0]: PWG.v1, c = randn(100 + 2 * acw, 80), the noise passed explicitly as decode.py does not need to
        conf = load_conf("parallel_wavegan.v1")
        gp, hop = conf["generator_params"], conf["hop_size"]
        acw = gp["aux_context_window"]
        gen = torch.Generator(device="cpu").manual_seed(5)
        c = torch.randn(100, gp["aux_channels"], generator=gen)  # inference() replicate-pads the context frames itself
        z = torch.randn(100 * hop, 1, generator=gen)
        if use_ref:
            g_ref = ref_run.generator("ParallelWaveGANGenerator", gp)
            fn = lambda: g_ref.inference(c, x=z)  # noqa: E731  (models/parallel_wavegan.py:229-261)
        else:
            from parallelwavegan_amd.models import ParallelWaveGANGenerator

            g0 = ParallelWaveGANGenerator(**gp)
            g0.remove_weight_norm()
            sd = {k: v.detach() for k, v in g0.state_dict().items()}
            cp = torch.nn.functional.pad(c.t().unsqueeze(0), (acw, acw), mode="replicate")
            fn = lambda: torch_cpu.pwg_generator(sd, z.t().unsqueeze(0), cp, **gp)  # noqa: E731
        nt = fastest_threads(fn)
        best, n, y = best_of(fn, budget_s)
        res["pwg_v1_c0"] = {"value": y.numel() / best, "unit": "samples/s", "best_s": best, "calls": n, "cores": nt,
                            "kind": "reference" if use_ref else "port",
                            "sample": "BASELINE configs[0]: ParallelWaveGANGenerator.inference on a random 80 x 100 mel "
                                      f"(context {acw} frames replicate-padded), B=1, best of {n}, {nt} of {cores} host threads"}
        # ---- Multi-band MelGAN.v2 + PQMF synthesis
        conf = load_conf("multi_band_melgan.v2")
        gp, hop = conf["generator_params"], conf["hop_size"]
        if use_ref:
            ref_run.ref_shim.install()
            from parallel_wavegan.layers import PQMF as RefPQMF

            m_ref = ref_run.generator("MelGANGenerator", gp)
            m_ref.pqmf = RefPQMF(subbands=gp["out_channels"], **conf.get("pqmf_params", {}))  # as utils.load_model:346-353
            mk = lambda c_: (lambda: m_ref.inference(c_))  # noqa: E731
        else:
            from parallelwavegan_amd.models import MelGANGenerator

            g1 = MelGANGenerator(**gp)
            g1.remove_weight_norm()
            sd1 = {k: v.detach() for k, v in g1.state_dict().items()}
            mk = lambda c_: (lambda: torch_cpu.pqmf_synthesis(  # noqa: E731
                torch_cpu.melgan_generator(sd1, c_.t().unsqueeze(0), **gp), subbands=gp["out_channels"],
                **conf.get("pqmf_params", {})))
        per = {}
        for frames, budget in ((100, budget_s / 2), (800, budget_s)):
            c = torch.randn(frames, gp["in_channels"], generator=gen)
            fn = mk(c)
            if frames == 100:
                nt = fastest_threads(fn)
            best, n, y = best_of(fn, budget)
            per[frames] = {"samples_per_s": y.numel() / best, "best_s": best, "calls": n}
        res["mb_melgan_v2"] = {"value": per[800]["samples_per_s"], "unit": "samples/s", "cores": nt,
                               "kind": "reference" if use_ref else "port", "frames_100": per[100], "frames_800": per[800],
                               "sample": "MelGANGenerator.inference (generator + PQMF synthesis), B=1 x 800 frames, best of "
                                         f"{per[800]['calls']}, {nt} of {cores} host threads"}
    return res


RCCL_LOG = {"path": None}


def enable_rccl_log(rank):
    """RCCL prints the topology it detected and, per collective size, the algorithm / protocol its tuner picks
    (NCCL_DEBUG=INFO, subsystems INIT + TUNING) into a per-process file that rank 0 parses after the run: the first
    multi-GPU line then PROVES how many ranks took part and that the 283 MB exchange did not fall back to a
    single-link ring.  Honours values the caller already exported."""
    import tempfile

    path = os.path.join(tempfile.gettempdir(), f"pwg_rccl_rank{rank}_{os.getpid()}.log")
    if os.environ.get("NCCL_DEBUG", "").upper() not in ("INFO", "TRACE"):  # (the GPU boxes export NCCL_DEBUG=VERSION)
        os.environ["NCCL_DEBUG"] = "INFO"
    os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,TUNING,GRAPH")
    os.environ.setdefault("NCCL_DEBUG_FILE", path)
    RCCL_LOG["path"] = os.environ["NCCL_DEBUG_FILE"]


def rccl_log_summary(max_lines=6):
    """What this rank's RCCL log says: ranks of the communicator, channels, transports seen, algorithm / protocol
    choices of the tuner.  None when there is no log (gloo run, logging redirected elsewhere)."""
    import re

    path = RCCL_LOG["path"]
    if not path or not os.path.exists(path):
        return None
    nranks, channels, algos, transports, samples, large = None, None, {}, set(), [], None
    try:
        with open(path, errors="replace") as f:
            for line in f:
                m = re.search(r"n[rR]anks (\d+)", line)
                if m:
                    nranks = int(m.group(1))
                m = re.search(r"(\d+) coll channels", line)
                if m:
                    channels = int(m.group(1))
                for t in ("P2P/IPC", "P2P/direct", "via SHM", "via NET", "P2P/CUMEM"):
                    if t in line:
                        transports.add(t)
                m = re.search(r"(AllReduce|Broadcast)[^\n]*?[Aa]lgo(?:rithm)?\s*[:=]?\s*(\w+)[^\n]*?[Pp]roto(?:col)?\s*[:=]?\s*(\w+)", line)
                if m:
                    algo = {"0": "Tree", "1": "Ring", "2": "CollNetDirect", "3": "CollNetChain", "4": "NVLS", "5": "NVLSTree",
                            "6": "PAT", "TREE": "Tree", "RING": "Ring"}.get(m.group(2), m.group(2))
                    proto = {"0": "LL", "1": "LL128", "2": "Simple", "SIMPLE": "Simple"}.get(m.group(3), m.group(3))
                    key = f"{m.group(1)}:{algo}/{proto}"
                    algos[key] = algos.get(key, 0) + 1
                    nb = re.search(r"(\d+) Bytes", line)
                    ch = re.search(r"channel\{Lo\.\.Hi\}=\{(\d+)\.\.(\d+)\}", line)
                    if m.group(1) == "AllReduce" and nb and (large is None or int(nb.group(1)) > large["bytes"]):
                        # the largest all-reduce seen = a full gradient bucket: ring or not, over how many channels
                        large = {"bytes": int(nb.group(1)), "algo": algo, "proto": proto,
                                 "channels": int(ch.group(2)) - int(ch.group(1)) + 1 if ch else None}
                    if len(samples) < max_lines:
                        samples.append(line.strip()[-200:])
    except OSError:
        return None
    return {"log": path, "nranks": nranks, "coll_channels": channels, "transports": sorted(transports),
            "algo_proto_counts": algos, "large_allreduce": large, "sample_lines": samples}


COMPACT_LIMIT = 1900  # the driver keeps a 2000-byte tail of stdout: the ONE stdout line must fit inside it


def _r(x, nd=4):
    return round(x, nd) if isinstance(x, float) else x


def compact_line(out):
    """The single stdout line: every key of the bench contract + roofline + cpu_baseline + parity + the per-config
    training steps/s, well under the driver's 2000-byte stdout tail.  The full record (per-kernel tables, loss
    histories, latency, distributed detail) goes to ``gpurun_out/bench_detail.json`` and to stderr."""
    rf, cb, par, tr = out.get("roofline"), out.get("cpu_baseline"), out.get("parity"), out.get("train")
    cfg = out["config"]
    line = {
        "metric": out["metric"], "value": _r(out["value"], 1), "unit": out["unit"], "n_gpus": out["n_gpus"],
        "steps": out["steps"], "warmup": out["warmup"], "ms_per_step": _r(out["ms_per_step"]),
        "higher_is_better": True, "scaling": out["scaling"], "vs_baseline": out["vs_baseline"], "dtype": out["dtype"],
        "data": out["data"],
        "config": {"workload": f"HiFi-GAN V1 generator fwd, LJSpeech 22.05 kHz, B{cfg['batch_per_gpu']} x "
                               f"{cfg['frames']} mel frames per GPU, fp32, random init",
                   "global_batch": cfg["batch_per_gpu"] * out["n_gpus"], "parallelism": cfg["parallelism"]},
    }
    if rf:
        line["roofline"] = {k: _r(rf.get(k), 4 if k == "frac" else 2) for k in
                            ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                             "avg_launch_us", "flops_per_launch", "algorithmic_bytes_per_launch")}
        line["roofline"]["launches"] = rf.get("launches_timed")
    if cb:
        line["cpu_baseline"] = {"value": _r(cb["value"], 1), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                "sample": f"reference HiFiGANGenerator B1 x 800 frames, best of {cb['frames_800']['calls']}"
                                if cb["kind"] == "reference" else "oracle restatement B1 x 800 frames"}
        oth = cb.get("others") or {}
        if "pwg_v1_c0" in oth:  # BASELINE configs[0] (PWG.v1, 80 x 100 mel, CPU) and MB-MelGAN.v2 + PQMF, same host
            line["cpu_baseline"]["pwg_v1_c0"] = _r(oth["pwg_v1_c0"]["value"], 1)
            line["cpu_baseline"]["mb_melgan_v2"] = _r(oth["mb_melgan_v2"]["value"], 1)
    lat = out.get("latency")
    if lat:  # B = 1 at 100 and 800 frames (ms per utterance, graph replay)
        line["lat_ms"] = {k.lower().replace("_", ""): _r(v["ms"], 3) for k, v in lat.items()}
    inf = {}
    cfgs = out.get("configs") or {}
    for key, short in (("c1_pwg_inference", "pwg"), ("c4_mbmelgan_inference", "mb")):
        rec = cfgs.get(key) or {}
        if "batch" in rec:  # M samples/s: resident batch, B1 x 100 frames; max|hip - oracle| of the timed batch
            inf[short] = [_r(rec["batch"]["samples_per_s"] / 1e6, 2), _r(rec["B1_F100"]["samples_per_s"] / 1e6, 2),
                          float(f"{rec['batch']['parity']['max_abs_vs_oracle']:.2g}")]
    if inf:
        line["infer_Msps_b16_b1_err"] = inf
    if par:
        line["parity"] = {"max_abs_vs_oracle": float(f"{par['max_abs_vs_oracle']:.3g}"), "tol": par["tolerance"],
                          "ok": par["ok"]}
    line["train_steps_per_s"] = out.get("train_steps_per_s")
    line["train_ok"] = out.get("train_ok")
    line["hip_graph"] = out.get("hip_graph")
    if tr:
        line["train"] = {"config": out.get("train_config"), "batch_per_gpu": tr.get("batch_per_gpu"),
                         "ms_per_step": _r(tr.get("ms_per_step"), 3),
                         "frac_of_fp32_peak": _r(tr.get("reference_flop_frac_of_fp32_matrix_peak"), 4)}
        tcb = tr.get("cpu_baseline")
        if tcb:
            line["train"]["cpu_baseline"] = {"value": _r(tcb["value"], 4), "unit": "steps/s", "batch": tcb.get("batch"),
                                             "cores": tcb["cores"], "kind": tcb["kind"]}
        d = tr.get("dist")
        if d:
            rc = d.get("rccl") or {}
            line["train"]["dist"] = {"backend": d["backend"], "world_size": d["world_size"],
                                     "exposed_comm_ms_per_exchange": {k: _r(v, 3) for k, v in
                                                                      d.get("exposed_comm_ms_per_exchange", {}).items()},
                                     "rccl": {"nranks": rc.get("nranks"), "large_allreduce": rc.get("large_allreduce")}}
    line["detail"] = out.get("detail_file")
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > COMPACT_LIMIT:  # never exceed the tail: drop the optional objects, least important first
        for key in ("detail", "train", "infer_Msps_b16_b1_err", "parity"):
            line.pop(key, None)
            text = json.dumps(line, separators=(",", ":"))
            if len(text) <= COMPACT_LIMIT:
                break
    return text


def write_detail(out):
    """Full record -> gpurun_out/bench_detail.json (merged back by gpurun) and stderr; returns the path or None."""
    text = json.dumps(out)
    print("[bench detail] " + text, file=sys.stderr, flush=True)
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, os.environ.get("PWG_BENCH_DETAIL", "bench_detail.json"))
        with open(path, "w") as f:
            f.write(text + "\n")
        return os.path.relpath(path, ROOT)
    except OSError:
        return None


def self_launch(args):
    """``python bench.py --gpus N`` without a launcher: one rank per GPU through the package's launcher."""
    from parallelwavegan_amd.distributed import launch

    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    return launch.spawn(cmd, args.gpus, master_addr="127.0.0.1", master_port=0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="utterances per step per GPU")
    ap.add_argument("--frames", type=int, default=800, help="mel frames per utterance (800 = 9.3 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the training-step measurements")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-utterance latency runs")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip BASELINE configs C1 (PWG inference), C2 and C4 (training); C3 is always measured")
    ap.add_argument("--train-steps", type=int, default=50)
    ap.add_argument("--train-warmup", type=int, default=10, help=">= 3 so that the hipGraph capture is not timed")
    ap.add_argument("--no-graph", action="store_true", help="run the training step eagerly (no hipGraph replay)")
    ap.add_argument("--train-batch", type=int, default=0, help="override the C3 batch per GPU (default: the recipe's 16)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    # stdout carries exactly ONE line, the result: whatever libraries print there (RCCL's version banner, gloo's
    # connection messages ...) is sent to stderr for the whole run; the JSON line goes to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    n_dev = max(torch.cuda.device_count(), 1)
    dev = torch.device("cuda", local_rank % n_dev)
    torch.cuda.set_device(dev)  # before the process group exists: RCCL binds to the current device
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm.  RCCL refuses two ranks on one device, so when there are fewer GPUs
        # than ranks (single-GPU smoke run of the multi-rank path) the collectives go through gloo.
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        backend = os.environ.get("PWG_DIST_BACKEND", "gloo" if local_world > n_dev else "nccl")
        if backend == "nccl":
            enable_rccl_log(rank)
        # (gloo's C++ side prints its connection banner to stdout: keep stdout for the one JSON line)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend)
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    force_dist = world == 1 and os.environ.get("PWG_FORCE_DIST") == "1"
    if force_dist:
        # single-GPU exercise of the data-parallel path on RCCL itself: a process group of one rank, the
        # trainer in distributed mode, every bucket all-reduce really issued (PWG_FORCE_COLLECTIVES)
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ["PWG_FORCE_COLLECTIVES"] = "1"
        if os.environ.get("PWG_DIST_BACKEND", "nccl") == "nccl":
            enable_rccl_log(0)
        dist.init_process_group(os.environ.get("PWG_DIST_BACKEND", "nccl"), rank=0, world_size=1)

    from parallelwavegan_amd import ops
    from parallelwavegan_amd.models import HiFiGANGenerator

    c3 = load_conf("hifigan.v1")
    g_params = c3["generator_params"]
    torch.manual_seed(1234)
    g = HiFiGANGenerator(**g_params)
    g.remove_weight_norm()  # as bin/decode.py:147
    g = g.to(dev).eval()
    gen = torch.Generator(device="cpu").manual_seed(100 + rank)
    c_host = torch.randn(args.batch, 80, args.frames, generator=gen)
    c = c_host.to(dev)
    samples_per_step = args.batch * args.frames * g.upsample_factor

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the forward is replayed as one hipGraph (the 3 MRF blocks of a stage are parallel branches)
    from parallelwavegan_amd.graphs import GraphedInference

    g.branch_streams = not args.no_graph
    run = g if args.no_graph else GraphedInference(g)
    with torch.no_grad():
        for _ in range(args.warmup):
            y = run(c)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = run(c)
        barrier()
        elapsed = time.perf_counter() - t0
    g.branch_streams = False
    assert torch.isfinite(y).all()
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    # ---- parity of THIS workload: utterances 0 and B-1 of the timed batch against the oracle
    parity = None
    if rank == 0:
        from oracle import torch_cpu

        sd = {k: v.detach().cpu() for k, v in g.state_dict().items()}
        idx = sorted({0, args.batch - 1})
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
        with torch.no_grad():
            ref = torch_cpu.hifigan_generator(sd, c_host[idx], **g_params)
        err = (y[idx].cpu() - ref).abs().max().item()
        parity = {"max_abs_vs_oracle": err, "utterances_checked": idx, "oracle_abs_max": ref.abs().max().item(),
                  "tolerance": 1e-4, "ok": err <= 1e-4}

    # ---- dominant-kernel roofline: HIP events recorded inside the library on the launch stream
    # around every kernel of `prof_steps` further steps (pwg_prof_*, include/pwg_kernels.h)
    roofline = None
    if rank == 0:
        prof_steps = min(args.steps, 3)
        with ops.profile() as prof, torch.no_grad():
            for _ in range(prof_steps):
                g(c)
        name, r = max(prof.results.items(), key=lambda kv: kv[1]["ms"])
        achieved = r["flops"] / (r["ms"] * 1e-3) / 1e12
        traffic, traffic_src = None, None
        for pmc_name in ("r06_pmc_hbm_traffic.json", "r05_pmc_hbm_traffic.json", "r04_pmc_hbm_traffic.json", "r03_pmc_hbm_traffic.json"):
            pmc_file = os.path.join(ROOT, "profiles", pmc_name)
            if os.path.exists(pmc_file) and args.batch == 16 and args.frames == 800:
                # HBM bytes per launch of this kernel on THIS workload, from separate rocprofv3 --pmc passes
                # (FETCH_SIZE, WRITE_SIZE; KiB units; read side calibrated on a known byte count) --
                # counters cannot be read from inside the process, so the committed summary is cited
                with open(pmc_file) as f:
                    pmc = json.load(f)
                if pmc.get("kernel") == name and abs(pmc.get("launches_per_forward", r["launches"] / prof_steps)
                                                      - r["launches"] / prof_steps) < 0.5:  # same launch mix
                    traffic, traffic_src = pmc["hbm_bytes_per_launch"], "profiles/" + pmc_name
                    break
        roofline = {
            "kernel": name,
            "bound": "mfma",
            "achieved": achieved,
            "peak": FP32_MATRIX_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / FP32_MATRIX_PEAK_TFLOPS,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "launches_per_step": r["launches"] / prof_steps,
            "launches_timed": r["launches"],
            "avg_launch_us": r["ms"] * 1e3 / r["launches"],
            "flops_per_launch": r["flops"] / r["launches"],
            "algorithmic_bytes_per_launch": r["bytes"] / r["launches"],
            "algorithmic_GBps": r["bytes"] / (r["ms"] * 1e-3) / 1e9,
            "kernel_ms_per_step": r["ms"] / prof_steps,
            "kernel_ms_note": "serial event timing of eager launches; the timed region replays a hipGraph whose "
                              "MRF branches overlap, so ms_per_step is ~4 % shorter than the sum of all kernels' "
                              "serial times (round 3: 67.9 vs 70.6 ms): frac is the conservative figure",
            "share_of_step_kernel_time": r["ms"] / sum(v["ms"] for v in prof.results.values()),
            # every kernel family of the forward pass (algorithmic FLOPs: the one-launch residual units are
            # credited with the two convolutions they replace, not with their recomputed halo)
            "kernels": {k_: {"ms_per_step": v["ms"] / prof_steps, "launches_per_step": v["launches"] / prof_steps,
                             "TFLOPs": v["flops"] / max(v["ms"], 1e-9) * 1e-9,
                             "algorithmic_bytes_per_launch": v["bytes"] / max(v["launches"], 1),
                             "frac_of_peak": v["flops"] / max(v["ms"], 1e-9) * 1e-9 / FP32_MATRIX_PEAK_TFLOPS}
                        for k_, v in sorted(prof.results.items(), key=lambda kv: -kv[1]["ms"])},
            "all_kernels_frac_of_peak": sum(v["flops"] for v in prof.results.values())
            / (sum(v["ms"] for v in prof.results.values()) * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS,
        }

    # single-utterance latency (bin/decode.py's regime: batch 1), same graph-replay path
    latency = None
    if rank == 0 and not args.no_latency:
        latency = {}
        g.branch_streams = not args.no_graph
        for frames in (100, 800):
            c1 = torch.randn(1, 80, frames, generator=gen).to(dev)
            run1 = g if args.no_graph else GraphedInference(g)
            with torch.no_grad():
                for _ in range(3):
                    run1(c1)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(20):
                    run1(c1)
                torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / 20
            latency[f"B1_F{frames}"] = {"ms": dt * 1e3, "samples_per_s": frames * g.upsample_factor / dt,
                                        "rtf": dt / (frames * g.upsample_factor / 22050.0)}
            del run1
        g.branch_streams = False
    del y, run
    torch.cuda.empty_cache()

    # N = 1: ``train`` = C3 (BASELINE's 1-GPU training config, LJSpeech) and C5 rides in ``configs``; N > 1: ``train`` =
    # C5 = BASELINE configs[4] (HiFi-GAN V1 LibriTTS 24 kHz, B = 16 x 8400 per GPU, minibatch shards + RCCL gradient
    # all-reduce), the workload the multi-GPU scaling is quoted on, and C3 rides in ``configs``.
    train, configs = None, {}
    main_tag = "c5" if world > 1 else "c3"
    if not args.no_train:
        train = bench_train(args, main_tag, dev, rank, world, dist, args.train_steps, args.train_warmup)
        others = ["c3" if main_tag == "c5" else "c5"] + ([] if args.no_extra_configs else ["c2", "c4"])
        for tag in others:
            try:  # >= 50 timed steps after 10
                configs[tag + "_train"] = bench_train(args, tag, dev, rank, world, dist, args.train_steps, args.train_warmup)
            except Exception as e:  # noqa: BLE001  (extra evidence must never take the headline line down)
                configs[tag + "_train"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and not args.no_extra_configs:
        try:
            configs["c1_pwg_inference"] = bench_pwg_inference(dev)
        except Exception as e:  # noqa: BLE001
            configs["c1_pwg_inference"] = {"error": f"{type(e).__name__}: {e}"}
        try:
            configs["c4_mbmelgan_inference"] = bench_mbmelgan_inference(dev)
        except Exception as e:  # noqa: BLE001
            configs["c4_mbmelgan_inference"] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        value = samples_per_step * world * args.steps / elapsed
        train_brief, train_ok = None, None
        if train is not None:
            runs = {main_tag: train, **{k[:2]: v for k, v in configs.items() if k.endswith("_train")}}
            train_brief = {k: (round(v["value"], 3) if v.get("value") else None) for k, v in sorted(runs.items())}
            # False as soon as ANY measured training configuration produced a non-finite loss (or failed to run)
            train_ok = all(v.get("losses_finite", False) for v in runs.values())
        out = {
            "metric": "HiFi-GAN V1 22.05 kHz generator inference RTF^-1 (audio samples/s)",
            "value": value,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"HiFi-GAN V1 generator forward (configs[2], LJSpeech 22.05 kHz), "
                            f"{args.batch} utterances x {args.frames} mel frames per GPU per step, fp32, "
                            f"weight norm removed, random-init weights",
                "batch_per_gpu": args.batch,
                "frames": args.frames,
                "samples_per_step_per_gpu": samples_per_step,
                "parallelism": f"replicas x{world}",
            },
            "rtf": 22050.0 / value,
            # the training results in brief, early in the line (the full objects follow under "train" / "configs")
            "train_steps_per_s": train_brief,
            "train_ok": train_ok,
            "hip_graph": not args.no_graph,
            "parity": parity,
            "latency": latency,
            "roofline": roofline,
        }
        if train is not None:
            out["train"] = train
        if configs:
            out["configs"] = configs
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(g_params)
            if not args.no_extra_configs:
                try:
                    out["cpu_baseline"]["others"] = cpu_baseline_other_generators()
                except Exception as e:  # noqa: BLE001
                    out["cpu_baseline"]["others"] = {"error": f"{type(e).__name__}: {e}"}
            if train is not None:
                train["cpu_baseline"] = cpu_train_baseline(load_conf("hifigan.v1"))
        out["train_config"] = main_tag
        # the per-config truth, not the flag: true only if EVERY training configuration really replayed a hipGraph
        runs_all = ([train] if train is not None else []) + [v for k, v in configs.items() if k.endswith("_train")]
        out["hip_graph"] = (not args.no_graph) and all(v.get("hip_graph", False) for v in runs_all)
        out["summary"] = {"infer_samples_per_s": round(value, 1), "roofline_frac": roofline and round(roofline["frac"], 4),
                          "train_config": main_tag, "train_steps_per_s": train_brief, "train_ok": train_ok,
                          "n_gpus": world}
        out["detail_file"] = "gpurun_out/" + os.environ.get("PWG_BENCH_DETAIL", "bench_detail.json")
        write_detail(out)
        sys.stdout.flush()
        os.write(real_stdout, (compact_line(out) + "\n").encode())
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
