"""Training runtime (drop-in surface of ``parallel_wavegan.bin.train``: ``Trainer``, ``Collater``).

``Trainer`` keeps the reference's constructor, ``run`` / ``save_checkpoint`` / ``load_checkpoint``
and checkpoint layout (/root/reference/parallel_wavegan/bin/train.py:49-187) and the loss
composition of ``_train_step`` (:189-340).  What differs is how a step is executed on an MI355X:

* every convolution, loss, reparametrisation and optimizer update is a HIP kernel from
  libpwgkernels.so (see parallelwavegan_amd.functional / optimizers.fused);
* loss scalars are kept on the device and fetched once per logging interval, instead of the
  reference's 7-9 ``.item()`` host syncs per step;
* during the generator phase the discriminator's parameters do not require grad, so its
  weight-gradient kernels (whose results the reference computes and then discards at :327)
  are never launched;
* data parallelism: ``distributed.GradReducer`` (bucketed RCCL all-reduce overlapped with the
  backward pass) instead of apex DDP.
"""
import argparse
import logging
import os
import sys
from collections import defaultdict

import torch

from ..distributed import GradReducer
from ..optimizers import clip_grad_norm_


class _NullWriter:
    def add_scalar(self, *a, **k):
        pass

    def close(self):
        pass


def _summary_writer(outdir):
    try:
        from tensorboardX import SummaryWriter

        return SummaryWriter(outdir)
    except Exception:  # tensorboardX is optional
        return _NullWriter()


class Trainer(object):
    """Customized trainer for GAN vocoders on MI355X."""

    def __init__(self, steps, epochs, data_loader, sampler, model, criterion, optimizer, scheduler, config,
                 device=torch.device("cpu")):
        self.steps = steps
        self.epochs = epochs
        self.data_loader = data_loader
        self.sampler = sampler
        self.model = model
        self.criterion = criterion
        self.optimizer = optimizer
        self.scheduler = scheduler
        self.config = config
        self.device = device
        self.writer = _summary_writer(config["outdir"]) if config.get("rank", 0) == 0 else _NullWriter()
        self.finish_train = False
        self.total_train_loss = defaultdict(float)
        self.total_eval_loss = defaultdict(float)
        gtype = config.get("generator_type", "ParallelWaveGANGenerator")
        if "VQVAE" in gtype or "Duration" in gtype:
            raise NotImplementedError(f"{gtype} is outside the accelerated hot path (SURVEY.md s2)")
        self._concurrency_hint = 1.0
        self._pending = []  # (name, device scalar) of the current logging interval
        # config["record_loss_history"]: keep every step's loss scalars on the device (tiny async copies, no host
        # sync) so that a run can say afterwards WHICH loss went non-finite at WHICH step (bench.py)
        self._loss_hist = [] if config.get("record_loss_history", False) else None
        self._capturing = False
        self._bank_stream = None
        self._graphs, self._graph_seen = {}, {}
        if config.get("use_hip_graph", False) and config.get("branch_streams", True):
            for m in self.model.values():  # independent sub-networks become parallel graph branches
                if hasattr(m, "branch_streams"):
                    m.branch_streams = True
            if torch.device(device).type == "cuda":
                from .. import streams

                streams.reserve(torch.device(device))  # (created before any capture)
                # one more stream: the discriminator's weight images are packed beside the generator's forward pass
                self._bank_stream = torch.cuda.Stream(device=torch.device(device))
            # parameters of the sub-networks receive their gradients on the branch streams by design (the joins
            # are explicit events, streams.py); torch >= 2.9 warns about that once per process
            quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
            if quiet is not None:
                quiet(False)
            if any(hasattr(m, "branch_streams") for m in self.model.values()):
                # the branches share the chip: a launch need not fill it alone (fewer, more efficient tiles and
                # fewer weight-gradient slabs; measured +2.7 % on the HiFi-GAN V1 step).  Applied around every
                # step (the hint is process-wide), see _train_step
                # (a model may state its own default: MelGAN's three scale branches plan for the whole chip)
                # Round 5: the default is 1.0 again.  The matrix kernels of a captured step run one after the other (their
                # serial times add up to the step, profiles/r05_c3_time_by_category.txt), so each should plan for the whole
                # chip: same-box A/B, three alternating runs: C3 52.21 -> 51.65 ms, C5 48.75 -> 47.95 ms; stand-alone
                # the period discriminators' 512 / 1024-channel layers are 14 - 17 % faster under hint 1.0
                # (tools/bench_dshapes.py).  Round 3's 0.5 predates the split-K fill rule and the per-shape slice counts.
                hints = [getattr(m, "branch_concurrency_hint", 1.0) for m in self.model.values() if hasattr(m, "branch_streams")]
                self._concurrency_hint = float(os.environ.get("PWG_CONCURRENCY_HINT",
                                                              config.get("conv_concurrency_hint", min(hints))))
        # discriminators whose feature maps stay in pre-activation form inside a training step (HiFi-GAN MPD / MSD,
        # MelGAN): needs a feature-matching loss that applies the activation itself (this package's; a foreign
        # criterion gets the ordinary post-activation maps)
        from ..losses.feat_match_loss import FeatureMatchLoss as _OwnFM

        # Default: on for the MelGAN discriminators (C4 30.85 -> 30.53 ms), off for HiFi-GAN's: there the activation on the
        # operand load of the 512 / 1024-channel layers costs more MFMA-loop time than the 143 removed activation-gradient
        # launches give back (C3 49.0 -> 49.6 ms, C5 45.7 -> 46.5 ms, three alternating runs each,
        # profiles/r04_deferred_activation_ab.txt).  Config key / PWG_DEFER_ACT=0|1 override.
        from ..models.melgan import MelGANMultiScaleDiscriminator as _MelD

        fm = criterion.get("feat_match") if hasattr(criterion, "get") else None
        want = config.get("deferred_discriminator_activation", os.environ.get("PWG_DEFER_ACT"))
        if want is None:
            want = isinstance(self._module("discriminator"), _MelD)
        self._defer_act = bool(want in (True, 1, "1")) and (fm is None or isinstance(fm, _OwnFM))
        # weight preparation of a whole model in two launches per parameter epoch (weight_bank.WeightBank)
        self._banks = {}
        if config.get("use_weight_bank", os.environ.get("PWG_WEIGHT_BANK", "1") == "1"):
            from ..weight_bank import WeightBank

            self._banks = {k: WeightBank(self._module(k)) for k in ("generator", "discriminator")}
        self.reducers = None
        if config.get("distributed", False):
            self.reducers = {}
            for k in ("generator", "discriminator"):
                m = self._module(k)
                # hipGraph mode replays the backward pass as one graph per exchange group, so that group
                # k's all-reduce overlaps the replay of group k+1 (independent sub-discriminators)
                groups = None
                n_groups = int(config.get("ddp_grad_groups", 3))
                if config.get("use_hip_graph", False) and n_groups > 1 and hasattr(m, "grad_groups"):
                    groups = m.grad_groups(n_groups)
                self.reducers[k] = GradReducer(list(m.parameters()), groups=groups,
                                               bucket_bytes=int(config.get("ddp_bucket_bytes", 64 << 20)))
                self.reducers[k].broadcast_parameters(list(m.parameters()) + list(m.buffers()))

    # ------------------------------------------------------------------ plumbing
    def _module(self, key):
        m = self.model[key]
        return m.module if hasattr(m, "module") else m

    def run(self):
        try:
            from tqdm import tqdm

            self.tqdm = tqdm(initial=self.steps, total=self.config["train_max_steps"], desc="[train]",
                             disable=self.config.get("rank", 0) != 0 or not self.config.get("progress", True))
        except Exception:
            self.tqdm = None
        while True:
            self._train_epoch()
            if self.finish_train:
                break
        if self.tqdm is not None:
            self.tqdm.close()
        logging.info("Finished training.")

    def save_checkpoint(self, checkpoint_path):
        state_dict = {
            "optimizer": {k: self.optimizer[k].state_dict() for k in ("generator", "discriminator")},
            "scheduler": {k: self.scheduler[k].state_dict() for k in ("generator", "discriminator")},
            "steps": self.steps,
            "epochs": self.epochs,
            "model": {k: self._module(k).state_dict() for k in ("generator", "discriminator")},
        }
        d = os.path.dirname(checkpoint_path)
        if d and not os.path.exists(d):
            os.makedirs(d)
        torch.save(state_dict, checkpoint_path)

    def load_checkpoint(self, checkpoint_path, load_only_params=False):
        state_dict = torch.load(checkpoint_path, map_location="cpu")
        self._module("generator").load_state_dict(state_dict["model"]["generator"])
        self._module("discriminator").load_state_dict(state_dict["model"]["discriminator"], strict=False)
        if not load_only_params:
            self.steps = state_dict["steps"]
            self.epochs = state_dict["epochs"]
            for k in ("generator", "discriminator"):
                self.optimizer[k].load_state_dict(state_dict["optimizer"][k])
                self.scheduler[k].load_state_dict(state_dict["scheduler"][k])

    def _log(self, name, value):
        """Record a loss scalar without synchronising the host (rank 0 only: the other ranks never
        flush, bin/train.py:378 of this file's ``_train_epoch``)."""
        if self.config.get("rank", 0) == 0:
            self._pending.append((name, value.detach()))

    def _flush_pending(self):
        for entry in self._graphs.values():  # losses accumulated on the device by graph replays
            if entry["accum"] is not None and entry["count"]:
                for n, v in zip(entry["names"], entry["accum"].cpu().tolist()):
                    self.total_train_loss[n] += v
                entry["accum"].zero_()
                entry["count"] = 0
        if not self._pending:
            return
        names = [n for n, _ in self._pending]
        vals = torch.stack([v.reshape(()) for _, v in self._pending]).cpu().tolist()  # ONE D2H copy
        for n, v in zip(names, vals):
            self.total_train_loss[n] += v
        self._pending = []

    def loss_history(self):
        """[(step, {loss name: value})] of every step since construction (``record_loss_history``); one host sync."""
        if not self._loss_hist:
            return []
        flat = torch.cat([v.reshape(-1) for _, _, v in self._loss_hist]).cpu().tolist()
        out, i = [], 0
        for step, names, _ in self._loss_hist:
            out.append((step, dict(zip(names, flat[i:i + len(names)]))))
            i += len(names)
        return out

    # ------------------------------------------------------------------ one optimisation step
    def _generator_forward(self, x):
        y_ = self.model["generator"](*x)
        y_mb_ = None
        if self.config["generator_params"]["out_channels"] > 1:
            y_mb_ = y_
            y_ = self.criterion["pqmf"].synthesis(y_mb_)
        return y_, y_mb_

    def _step_optimizer(self, key, loss, roots=None):
        """Backward, [yield (key, group) = gradient-exchange point(s)], clip, update.  The caller
        completes the exchange at the yields (eagerly: RCCL calls are never captured in a hipGraph).
        When the reducer has several exchange groups (hipGraph data-parallel mode), the backward pass is
        issued group by group -- ``inputs=`` restricted autograd calls over independent sub-networks --
        with a yield after each, so that every group becomes its own graph segment while capturing.
        ``roots``: tensors that separate the loss arithmetic from the sub-networks (the discriminators'
        logits): the loss is differentiated down to them ONCE and every group's backward starts there, so
        the shared loss nodes are neither re-executed nor re-captured per group.  Eager warm-up steps take
        the same grouped path, i.e. the captured kernel sequence has run eagerly before."""
        cfg = self.config
        opt = self.optimizer[key]
        params = list(self._module(key).parameters())
        for p in params:
            p.grad = None
        reducer = self.reducers[key] if self.reducers else None
        if reducer is not None:
            reducer.prepare()
        if reducer is not None and len(reducer.groups) > 1:
            last = len(reducer.groups) - 1
            starts, seeds = [loss], None
            if roots:
                roots = [t for t in roots if t.requires_grad]
                seeds = list(torch.autograd.grad(loss, roots, retain_graph=True, allow_unused=True))
                starts = [t for t, g in zip(roots, seeds) if g is not None]
                seeds = [g for g in seeds if g is not None]
            for gi, gparams in enumerate(reducer.groups):
                inputs = [p for p in gparams if p.requires_grad]
                if inputs:
                    torch.autograd.backward(starts, seeds, inputs=inputs, retain_graph=gi < last)
                reducer.zero_missing(gi)  # (recorded in this group's graph segment while capturing)
                yield (key, gi)
        else:
            loss.backward()
            if reducer is not None:
                reducer.zero_missing()
                yield (key, None)
        fused = self._is_fused(opt)
        if not fused:
            # an optimizer of torch's own (any `*_optimizer_type` the reference resolves from torch.optim: SGD, NAdam
            # ...): it reads `.grad` and knows nothing of bucket slots or a folded averaging factor (torch's Adam
            # family even asserts that an attribute called `grad_scale` is unset); the update itself is then ATen's
            scale = 1.0
            if reducer is not None:
                scale = 1.0 / reducer.world
                for p in params:
                    p.grad = reducer.flat_grads[p].clone()
            pairs = [(p, p.grad) for p in params if p.grad is not None]
            if scale != 1.0:
                for _, g in pairs:
                    g.mul_(scale)
            max_norm = cfg.get(f"{key}_grad_norm", -1)
            if max_norm > 0:
                clip_grad_norm_(pairs, max_norm)
            opt.step()
            if not self._capturing:
                self.scheduler[key].step()
            return
        if reducer is not None:
            opt.grad_scale = 1.0 / reducer.world
            opt.flat_grads = reducer.flat_grads
            pairs = [(p, reducer.flat_grads[p]) for p in params]
        else:
            opt.grad_scale = 1.0
            opt.flat_grads = None
            pairs = [(p, p.grad) for p in params if p.grad is not None]
        max_norm = cfg.get(f"{key}_grad_norm", -1)
        if max_norm > 0:
            if reducer is not None and reducer.world > 1:
                for _, g in pairs:  # clip on the averaged gradient, as the reference does after DDP
                    g.mul_(opt.grad_scale)
                opt.grad_scale = 1.0
            clip_grad_norm_(pairs, max_norm)
        opt.step()
        if not self._capturing:
            self.scheduler[key].step()

    # ------------------------------------------------------------------ hipGraph mode
    @staticmethod
    def _is_fused(opt):
        from ..optimizers.fused import _FusedBase

        return isinstance(opt, _FusedBase)

    def _graph_ok(self):
        cfg = self.config
        return (cfg.get("use_hip_graph", False)
                # the captured step replays the fused optimizers' device-scalar launches; torch's own optimizers run eagerly
                and all(self._is_fused(o) for o in self.optimizer.values())
                # models that take host-side random decisions per call (StyleMelGAN's random windows)
                and all(getattr(self._module(k), "hip_graph_safe", True) for k in ("generator", "discriminator")))

    def _phases(self):
        cfg = self.config
        return (self.steps > cfg.get("generator_train_start_steps", 0),
                self.steps > cfg["discriminator_train_start_steps"])

    def _train_step_graphed(self, batch):
        """Replay the whole optimisation step (G forward/backward/update, D forward/backward/update,
        ~1500 kernel launches) as hipGraph launches.  The first ``graph_warmup_steps`` steps run
        eagerly (they size every cache and the allocator), then the step is captured once per
        (active phases, batch shapes) signature.  Single GPU: ONE graph.  Data parallel: the step is
        cut at its gradient-exchange points into up to three graphs (... G backward | G update ... D
        backward | D update) that share one memory pool; the bucketed RCCL all-reduces run eagerly
        between the replays (collectives are never captured).  Per replay the host only copies the
        batch into the static buffers, refreshes the optimizers' device scalars and steps the LR
        schedulers."""
        x, y = self._parse_batch(batch)
        key = (self._phases(), tuple(tuple(t.shape) for t in x if t is not None), tuple(y.shape))
        entry = self._graphs.get(key)
        if entry is None:
            self._graph_seen[key] = self._graph_seen.get(key, 0) + 1
            if self._graph_seen[key] <= self.config.get("graph_warmup_steps", 2):
                return False  # run this step eagerly
            torch.cuda.synchronize()
            # pinned staging for the captured optimizer / clip launches (cannot be allocated while capturing)
            from ..optimizers.fused import reserve_clip_tables_for_capture

            for opt in self.optimizer.values():
                if hasattr(opt, "reserve_for_capture"):
                    opt.reserve_for_capture()
            reserve_clip_tables_for_capture()
            static_x = [None if t is None else t.clone() for t in x]
            static_y = y.clone()
            self._flush_pending()
            self._capturing = True
            # every cached weight image is dropped so that all weight-preparation kernels of the step
            # are recorded inside the graph, in order, and re-run at each replay
            from ..ops import bump_param_epoch

            bump_param_epoch()
            for r in (self.reducers or {}).values():
                r.defer = True  # hooks only fill the buckets; the exchange happens between the graphs
            segments = []  # [(graph, exchange key or None)]
            pool = torch.cuda.graph_pool_handle()
            step = self._device_step_iter(static_x, static_y)
            try:
                done = False
                while not done:
                    graph = torch.cuda.CUDAGraph()
                    exchange = None
                    # data parallel: RCCL's watchdog thread polls events while we capture; only THIS thread's
                    # calls are capture-restricted then (kernels launched by the autograd threads into the
                    # capturing stream are recorded either way)
                    mode = "thread_local" if self.reducers else "global"
                    with torch.cuda.graph(graph, pool=pool, capture_error_mode=mode):
                        try:
                            exchange = next(step)
                        except StopIteration:
                            done = True
                            names = [n for n, _ in self._pending]
                            vals = (torch.stack([v.reshape(()) for _, v in self._pending])
                                    if self._pending else None)
                    segments.append((graph, exchange))
                    if exchange is not None:
                        # capture pass: nothing ran, nothing to exchange -- and with ``defer`` set (above) finish()
                        # only resets the reducer's bookkeeping, it neither launches nor waits for a collective
                        assert self.reducers[exchange[0]].defer
                        self.reducers[exchange[0]].finish()
            finally:
                self._capturing = False
                for r in (self.reducers or {}).values():
                    r.defer = False
            self._pending = []
            accum = torch.zeros_like(vals) if vals is not None else None
            entry = dict(segments=segments, x=static_x, y=static_y, names=names, vals=vals, accum=accum, count=0)
            self._graphs[key] = entry
            # the capture pass executed nothing: fall through and replay it for this step
        for s, t in zip(entry["x"], x):
            if s is not None:
                s.copy_(t, non_blocking=True)
        entry["y"].copy_(y, non_blocking=True)
        gen_on, disc_on = key[0]
        for k, on in (("generator", gen_on), ("discriminator", disc_on)):
            if not on:
                continue
            # with gradient clipping the captured step has already averaged the gradients (see _step_optimizer)
            clipped = self.config.get(f"{k}_grad_norm", -1) > 0
            self.optimizer[k].grad_scale = (1.0 / self.reducers[k].world) if (self.reducers and not clipped) else 1.0
            self.optimizer[k].prepare()
        for r in (self.reducers or {}).values():
            r.begin_replay()
        for graph, exchange in entry["segments"]:
            graph.replay()
            if exchange is not None:
                # the segment filled this exchange group's buckets; its all-reduce runs on RCCL's stream
                # while the next segment (the next group's backward) replays
                k, gi = exchange
                r = self.reducers[k]
                if gi is None:
                    r.exchange_all()
                else:
                    r.exchange_group(gi)
                    if gi == len(r.groups) - 1:
                        r.wait_all()
        from ..ops import bump_param_epoch

        bump_param_epoch()
        if entry["accum"] is not None:
            entry["accum"].add_(entry["vals"])
            entry["count"] += 1
            if self._loss_hist is not None:
                self._loss_hist.append((self.steps, entry["names"], entry["vals"].clone()))
        if gen_on:
            self.scheduler["generator"].step()
        if disc_on:
            self.scheduler["discriminator"].step()
        return True

    def _train_step(self, batch):
        """One optimisation step.  Around it the discriminator runs in deferred-activation form
        (layers.activation.PreActivated: no activation-gradient launches in the backward pass); outside the step --
        evaluation, user code -- its feature maps are the reference's post-activation tensors."""
        if not self._defer_act:
            return self._train_step_hinted(batch)
        from ..layers.activation import set_deferred_activation

        set_deferred_activation(self._module("discriminator"), True)
        try:
            return self._train_step_hinted(batch)
        finally:
            set_deferred_activation(self._module("discriminator"), False)

    def _train_step_hinted(self, batch):
        if self._concurrency_hint != 1.0:
            from .. import _lib

            _lib.lib().pwg_set_concurrency_hint(self._concurrency_hint)
            try:
                self._train_step_inner(batch)
            finally:
                _lib.lib().pwg_set_concurrency_hint(1.0)
        else:
            self._train_step_inner(batch)

    def _train_step_inner(self, batch):
        if self._graph_ok() and self._train_step_graphed(batch):
            pass
        else:
            x, y = self._parse_batch(batch)
            first = len(self._pending)
            self._device_step(x, y)
            if self._loss_hist is not None and len(self._pending) > first:
                new = self._pending[first:]
                self._loss_hist.append((self.steps, [n for n, _ in new], torch.stack([v.reshape(()) for _, v in new])))
        self.steps += 1
        if getattr(self, "tqdm", None) is not None:
            self.tqdm.update(1)
        self._check_train_finish()

    def _device_step(self, x, y):
        """One optimisation step, eagerly: the gradient exchanges run where the step yields (the buckets of
        an exchange group go out from the hooks as they fill; the optimizer waits at the key's last yield)."""
        for key, gi in self._device_step_iter(x, y):
            r = self.reducers[key]
            if gi is None or gi == len(r.groups) - 1:
                r.finish()

    def _device_step_iter(self, x, y):
        """Generator over the step's gradient-exchange points (yields ("generator" | "discriminator",
        exchange group or None = all) after the respective backward when data parallel; yields nothing
        on a single GPU)."""
        cfg = self.config
        disc_on = self.steps > cfg["discriminator_train_start_steps"]
        p_real = None  # discriminator outputs on the real signal, shared by the two phases

        def prep(key, with_bwd):
            if key in self._banks and y.is_cuda:
                self._banks[key].ensure(with_bwd)

        def prep_beside(key, with_bwd):
            """``prep`` on a side stream while the caller's stream goes on (only inside a capture, where the fork is a
            branch of the graph: the discriminator's images -- 70 M parameters, 0.5 ms of HBM-bound packing in the HiFi-GAN
            V1 step -- are not needed before the generator's forward pass is over).  Returns the event to wait for, or None."""
            from ..streams import fork_now

            if not (key in self._banks and y.is_cuda and self._bank_stream is not None and fork_now()
                    and os.environ.get("PWG_BANK_BESIDE", "1") == "1"):
                prep(key, with_bwd)
                return None
            cur = torch.cuda.current_stream(y.device)
            fork = torch.cuda.Event()
            fork.record(cur)
            self._bank_stream.wait_event(fork)
            with torch.cuda.stream(self._bank_stream):
                prep(key, with_bwd)
                done = torch.cuda.Event()
                done.record(self._bank_stream)
            return done

        # ---------------- generator ----------------
        if self.steps > cfg.get("generator_train_start_steps", 0):
            prep("generator", True)
            d_images = prep_beside("discriminator", True) if disc_on else None
            y_, y_mb_ = self._generator_forward(x)
            gen_loss = 0.0
            if cfg["use_stft_loss"]:
                sc_loss, mag_loss = self.criterion["stft"](y_, y)
                gen_loss = gen_loss + sc_loss + mag_loss
                self._log("train/spectral_convergence_loss", sc_loss)
                self._log("train/log_stft_magnitude_loss", mag_loss)
            if cfg.get("use_subband_stft_loss", False):
                gen_loss = gen_loss * 0.5  # balance with the sub-band term (train.py:242-247)
                y_mb = self.criterion["pqmf"].analysis(y)
                sub_sc_loss, sub_mag_loss = self.criterion["sub_stft"](y_mb_, y_mb)
                gen_loss = gen_loss + 0.5 * (sub_sc_loss + sub_mag_loss)
                self._log("train/sub_spectral_convergence_loss", sub_sc_loss)
                self._log("train/sub_log_stft_magnitude_loss", sub_mag_loss)
            if cfg.get("use_mel_loss", False):
                mel_loss = self.criterion["mel"](y_, y)
                gen_loss = gen_loss + mel_loss
                self._log("train/mel_loss", mel_loss)
            gen_loss = gen_loss * cfg.get("lambda_aux", 1.0)
            if disc_on:
                if d_images is not None:
                    torch.cuda.current_stream(y.device).wait_event(d_images)
                # D acts as a fixed critic for G(c): no D weight gradients (they would be discarded)
                d_params = list(self._module("discriminator").parameters())
                for p in d_params:
                    p.requires_grad_(False)
                p_ = self.model["discriminator"](y_)
                for p in d_params:
                    p.requires_grad_(True)
                adv_loss = self.criterion["gen_adv"](p_)
                self._log("train/adversarial_loss", adv_loss)
                if cfg.get("use_feat_match_loss", False):
                    # The discriminator's weights do not change between this phase and the discriminator
                    # phase, so its pass over the real signal (feature-matching targets here) is
                    # evaluated once WITH autograd state and reused there -- except for sub-discriminators
                    # whose forward has a side effect (spectral-norm power iteration): those are
                    # re-evaluated, in the reference's order, exactly as often as the reference does.
                    if (cfg.get("reuse_real_discriminator_pass", True)
                            and hasattr(self._module("discriminator"), "stateful_outputs")):
                        p = p_real = self.model["discriminator"](y)
                    else:
                        with torch.no_grad():
                            p = self.model["discriminator"](y)
                    fm_loss = self.criterion["feat_match"](p_, p)  # (targets are detached inside)
                    self._log("train/feature_matching_loss", fm_loss)
                    adv_loss = adv_loss + cfg["lambda_feat_match"] * fm_loss
                gen_loss = gen_loss + cfg["lambda_adv"] * adv_loss
            self._log("train/generator_loss", gen_loss)
            yield from self._step_optimizer("generator", gen_loss)

        # ---------------- discriminator ----------------
        if disc_on:
            prep("discriminator", True)  # (no launch when the generator phase prepared them: D has not changed since)
            if cfg.get("update_prediction_after_generator_update", True):
                prep("generator", False)  # the generator was just updated; this pass needs forward images only
                with torch.no_grad():
                    y_, _ = self._generator_forward(x)
            if p_real is not None:
                redo = self._module("discriminator").stateful_outputs()
                fresh = self.model["discriminator"](y, only=redo) if redo else []
                p = [fresh[i] if i in redo else p_real[i] for i in range(len(p_real))]
            else:
                p = self.model["discriminator"](y)
            p_ = self.model["discriminator"](y_.detach())
            real_loss, fake_loss = self.criterion["dis_adv"](p_, p)
            dis_loss = real_loss + fake_loss
            self._log("train/real_loss", real_loss)
            self._log("train/fake_loss", fake_loss)
            self._log("train/discriminator_loss", dis_loss)
            # (the loss depends on the sub-discriminators only through their final outputs)
            roots = None
            if isinstance(p_, (list, tuple)) and p_ and isinstance(p_[0], (list, tuple)):
                roots = [out[-1] for out in list(p) + list(p_) if out is not None and torch.is_tensor(out[-1])]
            yield from self._step_optimizer("discriminator", dis_loss, roots=roots)

    def _train_epoch(self):
        for train_steps_per_epoch, batch in enumerate(self.data_loader["train"], 1):
            self._train_step(batch)
            if self.config.get("rank", 0) == 0:
                self._check_log_interval()
                self._check_eval_interval()
                self._check_save_interval()
            if self.finish_train:
                return
        self.epochs += 1
        self.train_steps_per_epoch = train_steps_per_epoch
        logging.info(f"(Steps: {self.steps}) Finished {self.epochs} epoch training "
                     f"({self.train_steps_per_epoch} steps per epoch).")
        if self.config.get("distributed", False) and self.sampler.get("train") is not None:
            self.sampler["train"].set_epoch(self.epochs)

    @torch.no_grad()
    def _eval_step(self, batch):
        cfg = self.config
        x, y = self._parse_batch(batch)
        y_, y_mb_ = self._generator_forward(x)
        aux_loss = 0.0
        if cfg["use_stft_loss"]:
            sc_loss, mag_loss = self.criterion["stft"](y_, y)
            aux_loss = aux_loss + sc_loss + mag_loss
            self.total_eval_loss["eval/spectral_convergence_loss"] += sc_loss.item()
            self.total_eval_loss["eval/log_stft_magnitude_loss"] += mag_loss.item()
        if cfg.get("use_subband_stft_loss", False):
            aux_loss = aux_loss * 0.5
            y_mb = self.criterion["pqmf"].analysis(y)
            sub_sc_loss, sub_mag_loss = self.criterion["sub_stft"](y_mb_, y_mb)
            aux_loss = aux_loss + 0.5 * (sub_sc_loss + sub_mag_loss)
        if cfg.get("use_mel_loss", False):
            mel_loss = self.criterion["mel"](y_, y)
            aux_loss = aux_loss + mel_loss
            self.total_eval_loss["eval/mel_loss"] += mel_loss.item()
        aux_loss = aux_loss * cfg.get("lambda_aux", 1.0)
        p_ = self.model["discriminator"](y_)
        adv_loss = self.criterion["gen_adv"](p_)
        gen_loss = aux_loss + cfg["lambda_adv"] * adv_loss
        p = self.model["discriminator"](y)
        if cfg.get("use_feat_match_loss", False):
            fm_loss = self.criterion["feat_match"](p_, p)
            self.total_eval_loss["eval/feature_matching_loss"] += fm_loss.item()
            gen_loss = gen_loss + cfg["lambda_adv"] * cfg["lambda_feat_match"] * fm_loss
        real_loss, fake_loss = self.criterion["dis_adv"](p_, p)
        self.total_eval_loss["eval/adversarial_loss"] += adv_loss.item()
        self.total_eval_loss["eval/generator_loss"] += float(gen_loss)
        self.total_eval_loss["eval/real_loss"] += real_loss.item()
        self.total_eval_loss["eval/fake_loss"] += fake_loss.item()
        self.total_eval_loss["eval/discriminator_loss"] += (real_loss + fake_loss).item()

    def _eval_epoch(self):
        logging.info(f"(Steps: {self.steps}) Start evaluation.")
        for key in self.model.keys():
            self.model[key].eval()
        n = 0
        for n, batch in enumerate(self.data_loader["dev"], 1):
            self._eval_step(batch)
        for key in self.total_eval_loss.keys():
            self.total_eval_loss[key] /= max(n, 1)
            logging.info(f"(Steps: {self.steps}) {key} = {self.total_eval_loss[key]:.4f}.")
        self._write_to_tensorboard(self.total_eval_loss)
        self.total_eval_loss = defaultdict(float)
        for key in self.model.keys():
            self.model[key].train()

    @torch.no_grad()
    def _parse_batch(self, batch):
        inputs, targets = batch

        def dev(t):
            return None if t is None else t.to(self.device, non_blocking=True)

        if isinstance(inputs, torch.Tensor):
            x = [dev(inputs)]
        elif isinstance(inputs, (tuple, list)):
            x = [dev(t) for t in inputs]
        else:
            raise ValueError(f"Not supported type ({type(inputs)}).")
        if isinstance(targets, torch.Tensor):
            y = dev(targets)
        elif isinstance(targets, (tuple, list)):
            y = [dev(t) for t in targets]
        else:
            raise ValueError(f"Not supported type ({type(targets)}).")
        return x, y

    def _write_to_tensorboard(self, loss):
        for key, value in loss.items():
            self.writer.add_scalar(key, value, self.steps)

    def _check_save_interval(self):
        if self.steps % self.config["save_interval_steps"] == 0:
            self.save_checkpoint(os.path.join(self.config["outdir"], f"checkpoint-{self.steps}steps.pkl"))
            logging.info(f"Successfully saved checkpoint @ {self.steps} steps.")

    def _check_eval_interval(self):
        if self.steps % self.config["eval_interval_steps"] == 0:
            self._eval_epoch()

    def _check_log_interval(self):
        if self.steps % self.config["log_interval_steps"] == 0:
            self._flush_pending()
            for key in self.total_train_loss.keys():
                self.total_train_loss[key] /= self.config["log_interval_steps"]
                logging.info(f"(Steps: {self.steps}) {key} = {self.total_train_loss[key]:.4f}.")
            self._write_to_tensorboard(self.total_train_loss)
            self.total_train_loss = defaultdict(float)

    def _check_train_finish(self):
        if self.steps >= self.config["train_max_steps"]:
            self.finish_train = True


class Collater(object):
    """Random-crop collate function for the mel -> waveform configurations (drop-in for the reference's
    ``Collater``, bin/train.py:646-896, mel2wav branch).

    Input: list of ``(audio[T], mel[T', C])`` numpy pairs (the dataset contract of SURVEY.md s2).
    Output: ``((z,) c), y`` with ``c`` (B, C, batch_max_frames + 2 * aux_context_window), ``y`` (B, 1,
    batch_max_steps) and, for Parallel WaveGAN (``use_noise_input``), ``z ~ N(0, 1)`` like ``y``.
    With ``use_f0_and_excitation`` (UHiFiGAN) the items are ``(audio, mel, f0[T'], excitation[T', hop])``
    and the inputs become ``(c, f0 (B, 1, F), excitation (B, 1, F * hop))`` cut with the same frame window.
    Utterances whose mel is not longer than the crop are dropped.  ``pin_memory=True`` returns pinned
    tensors so that the trainer's ``.to(device, non_blocking=True)`` overlaps the copy with compute.
    The variants for duration or global/local conditioning inputs belong to model families outside
    the accelerated hot path and raise.
    """

    def __init__(self, batch_max_steps=20480, hop_size=256, aux_context_window=2, use_noise_input=False,
                 use_f0_and_excitation=False, use_aux_input=True, use_duration=False, use_global_condition=False,
                 use_local_condition=False, pad_value=0, pin_memory=False):
        import numpy as np

        self._np = np
        if use_duration or use_global_condition or use_local_condition or not use_aux_input:
            raise NotImplementedError("only the mel -> waveform collation (configs C1-C5, UHiFiGAN) is provided")
        if use_f0_and_excitation and use_noise_input:
            raise NotImplementedError("noise input together with f0/excitation is not used by any model")
        self.use_f0_and_excitation = use_f0_and_excitation
        if batch_max_steps % hop_size != 0:
            batch_max_steps -= batch_max_steps % hop_size
        self.hop_size = hop_size
        self.batch_max_steps = batch_max_steps
        self.batch_max_frames = batch_max_steps // hop_size
        self.aux_context_window = aux_context_window
        self.use_noise_input = use_noise_input
        self.pin_memory = pin_memory
        self.start_offset = aux_context_window
        self.end_offset = -(self.batch_max_frames + aux_context_window)
        self.mel_threshold = self.batch_max_frames + 2 * aux_context_window

    def _adjust_length(self, x, c, *extra):
        np = self._np
        if len(x) < len(c) * self.hop_size:
            x = np.pad(x, (0, len(c) * self.hop_size - len(x)), mode="edge")
        assert len(x) == len(c) * self.hop_size, (len(x), len(c), self.hop_size)
        return (x, c) + tuple(extra)

    def __call__(self, batch):
        np = self._np
        items = [self._adjust_length(*b) for b in batch if len(b[1]) > self.mel_threshold]
        acw, frames = self.aux_context_window, self.batch_max_frames
        ys, cs, fs, es = [], [], [], []
        # (all start frames are drawn first, item by item, as the reference does: same numpy stream)
        starts = [np.random.randint(self.start_offset, len(it[1]) + self.end_offset) for it in items]
        for it, start in zip(items, starts):
            x, c = it[0], it[1]
            ys.append(x[start * self.hop_size: start * self.hop_size + self.batch_max_steps])
            cs.append(c[start - acw: start + frames + acw])
            if self.use_f0_and_excitation:
                fs.append(it[2][start - acw: start + frames + acw])
                es.append(it[3][start - acw: start + frames + acw])
        y = torch.from_numpy(np.ascontiguousarray(np.stack(ys), dtype=np.float32)).unsqueeze(1)
        c = torch.from_numpy(np.ascontiguousarray(np.stack(cs), dtype=np.float32)).transpose(2, 1).contiguous()
        inputs = (c,)
        if self.use_noise_input:
            inputs = (torch.randn(y.size()),) + inputs
        if self.use_f0_and_excitation:
            f = torch.from_numpy(np.ascontiguousarray(np.stack(fs), dtype=np.float32)).unsqueeze(1)
            e = torch.from_numpy(np.ascontiguousarray(np.stack(es), dtype=np.float32))
            inputs = inputs + (f, e.reshape(e.shape[0], 1, -1))
        if self.pin_memory:
            y = y.pin_memory()
            inputs = tuple(t.pin_memory() for t in inputs)
        return inputs, y


class DeviceCollater(object):
    """Collater whose corpus lives in HBM: the ``(audio[T], mel[T', C])`` pairs are
    uploaded once (288 GB per GPU hold thousands of hours of 22 kHz audio + mels), and a batch is ONE
    HIP gather launch (``pwg_gather_crop``) instead of numpy crops, ``np.stack``, ``torch.tensor`` and
    a pinned host-to-device copy per step (the reference's ``Collater`` + DataLoader path,
    bin/train.py:646-896,1125-1142).  Same crop rule and the same ``np.random.randint`` draw per item
    as :class:`Collater`, so with the same numpy seed both produce identical batches.

    ``collater(indices)`` takes utterance indices (what a sampler yields) and returns
    ``((z,) c), y`` on the device.
    """

    def __init__(self, pairs, device, batch_max_steps=20480, hop_size=256, aux_context_window=2,
                 use_noise_input=False):
        import numpy as np

        self._np = np
        if batch_max_steps % hop_size != 0:
            batch_max_steps -= batch_max_steps % hop_size
        self.hop_size, self.batch_max_steps = hop_size, batch_max_steps
        self.batch_max_frames = batch_max_steps // hop_size
        self.aux_context_window = aux_context_window
        self.use_noise_input = use_noise_input
        self.start_offset = aux_context_window
        self.end_offset = -(self.batch_max_frames + aux_context_window)
        self.mel_threshold = self.batch_max_frames + 2 * aux_context_window
        self.device = torch.device(device)
        audio_len = np.array([len(a) for a, _ in pairs], dtype=np.int64)
        self.mel_len = np.array([len(m) for _, m in pairs], dtype=np.int64)
        self.channels = int(pairs[0][1].shape[1])
        for (a, m) in pairs:
            # Collater._adjust_length: audio may be shorter than frames * hop (edge-padded), never longer
            assert len(a) <= len(m) * hop_size, (len(a), len(m), hop_size)
        audio_off = np.concatenate([[0], np.cumsum(audio_len)[:-1]])
        mel_off = np.concatenate([[0], np.cumsum(self.mel_len)[:-1]])
        dev = self.device
        self.audio = torch.from_numpy(np.concatenate([np.asarray(a, dtype=np.float32) for a, _ in pairs])).to(dev)
        self.mel = torch.from_numpy(np.concatenate([np.asarray(m, dtype=np.float32) for _, m in pairs])).to(dev)
        self.audio_off = torch.from_numpy(audio_off.astype(np.int64)).to(dev)
        self.audio_len = torch.from_numpy(audio_len).to(dev)
        self.mel_off = torch.from_numpy(mel_off.astype(np.int64)).to(dev)

    def __len__(self):
        return len(self.mel_len)

    def __call__(self, indices):
        import ctypes

        from .. import _lib
        from ..ops import _ptr, _stream

        np = self._np
        keep = [int(i) for i in indices if self.mel_len[int(i)] > self.mel_threshold]
        starts = [np.random.randint(self.start_offset, self.mel_len[i] + self.end_offset) for i in keep]
        b = len(keep)
        # the two small index vectors are the only host -> device traffic of a batch
        utt = torch.tensor(keep, dtype=torch.int32).to(self.device, non_blocking=True)
        start = torch.tensor(starts, dtype=torch.int32).to(self.device, non_blocking=True)
        frames_ctx = self.batch_max_frames + 2 * self.aux_context_window
        y = torch.empty((b, 1, self.batch_max_steps), device=self.device, dtype=torch.float32)
        c = torch.empty((b, self.channels, frames_ctx), device=self.device, dtype=torch.float32)
        _lib.check(_lib.lib().pwg_gather_crop(
            _ptr(self.audio), ctypes.c_void_p(self.audio_off.data_ptr()), ctypes.c_void_p(self.audio_len.data_ptr()),
            _ptr(self.mel), ctypes.c_void_p(self.mel_off.data_ptr()), ctypes.c_void_p(utt.data_ptr()),
            ctypes.c_void_p(start.data_ptr()), _ptr(y), _ptr(c), b, self.batch_max_steps, self.hop_size, frames_ctx,
            self.aux_context_window, self.channels, _stream()), "gather_crop")
        inputs = (c,)
        if self.use_noise_input:
            inputs = (torch.randn(y.size(), device=self.device),) + inputs
        return inputs, y


class _DeviceBatches(object):
    """Iterable of batches for one epoch from an HBM-resident corpus (:class:`DeviceCollater`): the index stream of a
    (Distributed)Sampler or a fresh permutation, cut into ``batch_size`` groups."""

    def __init__(self, collater, batch_size, sampler=None, shuffle=True):
        self.collater, self.batch_size, self.sampler, self.shuffle = collater, batch_size, sampler, shuffle

    def __len__(self):
        n = len(self.sampler) if self.sampler is not None else len(self.collater)
        return (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        if self.sampler is not None:
            idx = list(iter(self.sampler))
        else:
            idx = torch.randperm(len(self.collater)).tolist() if self.shuffle else list(range(len(self.collater)))
        for i in range(0, len(idx), self.batch_size):
            batch = self.collater(idx[i:i + self.batch_size])
            if batch[1].shape[0] > 0:
                yield batch


def main(argv=None):
    """``parallel-wavegan-train``: the command line of the reference's ``main()`` (bin/train.py:928-1546) for the
    mel -> waveform recipes on one or several MI355X.

    ``--train-dumpdir`` / ``--dev-dumpdir`` hold ``format: npy`` (``*-wave.npy`` + ``*-feats.npy``) or ``hdf5`` dumps;
    ``--config`` is a recipe YAML (objects are built by ``utils.build_from_config`` with the reference's defaulting
    rules).  One process per GPU: under ``parallelwavegan_amd.distributed.launch`` / ``torch.distributed.run`` the
    rank comes from ``--rank/--local_rank`` or ``LOCAL_RANK``, the process group is RCCL (``nccl`` backend; ``gloo`` when
    the ranks share a device or there is none), the training set is sharded by a ``DistributedSampler`` and gradients
    are averaged by ``GradReducer``.  Extra config keys of this engine: ``use_hip_graph`` (default true on a GPU),
    ``device_collater`` (default true: corpus resident in HBM, one gather launch per batch; falls back to the host
    DataLoader above ``device_collater_max_bytes``, default 64 GiB of dump files), ``ddp_grad_groups``."""
    import yaml

    from ..datasets import AudioMelDataset
    from ..utils import build_from_config

    p = argparse.ArgumentParser(description="Train a GAN vocoder on MI355X (see parallelwavegan_amd/bin/train.py).")
    for split in ("train", "dev"):
        p.add_argument(f"--{split}-wav-scp", default=None, type=str, help="kaldi-style wav.scp (not supported here)")
        p.add_argument(f"--{split}-feats-scp", default=None, type=str, help="kaldi-style feats.scp (not supported here)")
        p.add_argument(f"--{split}-segments", default=None, type=str, help="kaldi-style segments (not supported here)")
        p.add_argument(f"--{split}-dumpdir", default=None, type=str, help=f"directory including the {split} data")
    p.add_argument("--outdir", type=str, required=True, help="directory to save checkpoints")
    p.add_argument("--config", type=str, required=True, help="yaml format configuration file")
    p.add_argument("--pretrain", default="", type=str, nargs="?", help="checkpoint to load parameters from")
    p.add_argument("--resume", default="", type=str, nargs="?", help="checkpoint to resume training from")
    p.add_argument("--verbose", type=int, default=1, help="logging level; higher is more logging")
    p.add_argument("--rank", "--local_rank", default=None, type=int, help="local rank (default: $LOCAL_RANK or 0)")
    args = p.parse_args(argv)

    local_rank = args.rank if args.rank is not None else int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", str(local_rank)))
    args.rank, args.world_size, args.distributed = rank, world_size, world_size > 1
    if not torch.cuda.is_available():
        raise RuntimeError("parallel-wavegan-train of this package needs an MI355X: there is no CPU compute path")
    n_dev = torch.cuda.device_count()
    device = torch.device("cuda", local_rank % n_dev)
    torch.cuda.set_device(device)  # before the process group exists: RCCL binds to the current device
    if args.distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl") unless ranks of THIS node really share a device (single-GPU smoke runs of the multi-rank path):
        # the comparison is per node (LOCAL_WORLD_SIZE, exported by distributed/launch.py and torch.distributed.run),
        # not against the global world size -- 2 nodes x 8 GPUs is world 16 on 8 local devices and must stay on RCCL
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world_size)))
        backend = os.environ.get("PWG_DIST_BACKEND", "gloo" if local_world > n_dev else "nccl")
        torch.distributed.init_process_group(backend=backend, init_method="env://")
    if rank != 0:
        sys.stdout = open(os.devnull, "w")
    level = logging.DEBUG if args.verbose > 1 else logging.INFO if args.verbose > 0 else logging.WARN
    logging.basicConfig(level=level, stream=sys.stdout,
                        format="%(asctime)s (%(module)s:%(lineno)d) %(levelname)s: %(message)s")
    os.makedirs(args.outdir, exist_ok=True)
    for split in ("train", "dev"):
        if getattr(args, f"{split}_feats_scp") is not None or getattr(args, f"{split}_wav_scp") is not None:
            raise NotImplementedError("kaldi scp input is Kaldi glue (out of scope, SURVEY.md s2): use --*-dumpdir")
        if getattr(args, f"{split}_dumpdir") is None:
            raise ValueError(f"Please specify --{split}-dumpdir.")

    with open(args.config) as f:
        config = yaml.load(f, Loader=yaml.Loader)
    config.update(vars(args))
    from .. import __version__

    config["version"] = __version__
    config.setdefault("use_hip_graph", True)
    if rank == 0:
        with open(os.path.join(args.outdir, "config.yml"), "w") as f:
            yaml.dump(config, f, Dumper=yaml.Dumper)
    for key, value in config.items():
        logging.info(f"{key} = {value}")

    gtype = config.get("generator_type", "ParallelWaveGANGenerator")
    if "VQVAE" in gtype or "Duration" in gtype or config.get("use_local_condition") or config.get("use_global_condition"):
        raise NotImplementedError(f"{gtype} / conditioning inputs are outside the accelerated hot path (SURVEY.md s2)")
    if gtype == "UHiFiGANGenerator":
        raise NotImplementedError("f0 / excitation dumps: build the data loader around Collater(use_f0_and_excitation=True)")
    use_noise_input = "ParallelWaveGAN" in gtype
    acw = config["generator_params"].get("aux_context_window", 0)
    hop = config.get("hop_size")
    if config.get("remove_short_samples", True) or args.distributed:
        # Data parallel: always filter at dataset level.  With per-batch dropping (``remove_short_samples: false``) the
        # ranks could see different batch counts per epoch -- the next all-reduce would wait forever -- and different
        # batch shapes (a graph capture per shape).
        if args.distributed and not config.get("remove_short_samples", True):
            logging.warning("remove_short_samples=false is overridden in distributed runs: every rank must see the same "
                            "number of equally shaped batches")
        mel_length_threshold = config["batch_max_steps"] // hop + 2 * acw
    else:
        mel_length_threshold = None
    fmt = config.get("format", "hdf5")
    dataset = {split: AudioMelDataset(getattr(args, f"{split}_dumpdir"), format=fmt,
                                      mel_length_threshold=mel_length_threshold,
                                      allow_cache=config.get("allow_cache", False)) for split in ("train", "dev")}
    logging.info(f"The number of training files = {len(dataset['train'])}.")
    logging.info(f"The number of development files = {len(dataset['dev'])}.")

    sampler = {"train": None, "dev": None}
    if args.distributed:
        from torch.utils.data.distributed import DistributedSampler

        sampler = {"train": DistributedSampler(dataset["train"], num_replicas=world_size, rank=rank, shuffle=True),
                   "dev": DistributedSampler(dataset["dev"], num_replicas=world_size, rank=rank, shuffle=False)}
    ckw = dict(batch_max_steps=config["batch_max_steps"], hop_size=hop, aux_context_window=acw,
               use_noise_input=use_noise_input)
    use_device_collater = config.get("device_collater", True)
    if use_device_collater:
        # the HBM-resident corpus is uploaded by EVERY rank: fall back to the host DataLoader when it would not fit
        budget = int(config.get("device_collater_max_bytes", 64 << 30))
        total = sum(os.path.getsize(f) for split in ("train", "dev")
                    for f in set(dataset[split].audio_files) | set(dataset[split].mel_files))
        if total > budget:
            logging.warning(f"corpus of {total / 2 ** 30:.1f} GiB on disk exceeds device_collater_max_bytes "
                            f"({budget / 2 ** 30:.1f} GiB): using the host DataLoader / Collater path")
            use_device_collater = False
    if use_device_collater:
        data_loader = {split: _DeviceBatches(DeviceCollater([dataset[split][i] for i in range(len(dataset[split]))],
                                                            device, **ckw),
                                             config["batch_size"], sampler[split], shuffle=not args.distributed)
                       for split in ("train", "dev")}
    else:
        from torch.utils.data import DataLoader

        collater = Collater(pin_memory=False, **ckw)
        data_loader = {split: DataLoader(dataset=dataset[split], shuffle=not args.distributed, collate_fn=collater,
                                         batch_size=config["batch_size"], num_workers=config.get("num_workers", 0),
                                         sampler=sampler[split], pin_memory=config.get("pin_memory", False))
                       for split in ("train", "dev")}

    model, criterion, optimizer, scheduler = build_from_config(config, device)
    for part in (model, optimizer, scheduler):
        for v in part.values():
            logging.info(v)
    trainer = Trainer(steps=0, epochs=0, data_loader=data_loader, sampler=sampler, model=model, criterion=criterion,
                      optimizer=optimizer, scheduler=scheduler, config=config, device=device)
    if args.pretrain:
        trainer.load_checkpoint(args.pretrain, load_only_params=True)
        logging.info(f"Successfully load parameters from {args.pretrain}.")
    if args.resume:
        trainer.load_checkpoint(args.resume)
        logging.info(f"Successfully resumed from {args.resume}.")
    try:
        trainer.run()
    finally:
        if rank == 0:
            trainer.save_checkpoint(os.path.join(config["outdir"], f"checkpoint-{trainer.steps}steps.pkl"))
            logging.info(f"Successfully saved checkpoint @ {trainer.steps}steps.")
        if args.distributed:
            torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
