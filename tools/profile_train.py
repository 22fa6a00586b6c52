"""Per-launch breakdown of one HiFi-GAN V1 training step (events around every conv launch)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from parallelwavegan_amd import ops
import parallelwavegan_amd.functional as Fn

class A: pass
args = A(); args.train_steps = 2; args.train_warmup = 2; args.train_batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
recs = []
def wrap(name):
    orig = getattr(ops, name)
    def f(desc, *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = orig(desc, *a, **k); e1.record()
        recs.append((name, (desc.batch, desc.c_in, desc.c_out, desc.t_in, desc.t_out, desc.width, desc.kernel, desc.stride, desc.dilation, desc.groups, desc.transposed), e0, e1))
        return out
    setattr(ops, name, f)
import tempfile, time
# build trainer via bench internals
import types
out = None
orig_profile = ops.profile
# run warmup without wrapping, then wrap and run one step
class NoProf:
    def __enter__(self): self.results = {"x": dict(ms=1, launches=1, flops=0, bytes=0)}; return self
    def __exit__(self, *a): return False
state = {}
def patched_bench():
    from parallelwavegan_amd import losses, optimizers
    from parallelwavegan_amd.bin.train import Trainer
    from parallelwavegan_amd.models import HiFiGANGenerator, HiFiGANMultiScaleMultiPeriodDiscriminator
    torch.manual_seed(1)
    model = {"generator": HiFiGANGenerator(**bench.HIFIGAN_V1).to(dev), "discriminator": HiFiGANMultiScaleMultiPeriodDiscriminator(**bench.HIFIGAN_V1_D).to(dev)}
    criterion = {"gen_adv": losses.GeneratorAdversarialLoss(average_by_discriminators=False), "dis_adv": losses.DiscriminatorAdversarialLoss(average_by_discriminators=False),
                 "mel": losses.MelSpectrogramLoss(**bench.MEL_LOSS).to(dev), "feat_match": losses.FeatureMatchLoss(average_by_discriminators=False, average_by_layers=False, include_final_outputs=False)}
    opt = {k: optimizers.Adam(model[k].parameters(), lr=2e-4, betas=(0.5, 0.9)) for k in model}
    sched = {k: optimizers.lr_scheduler.MultiStepLR(opt[k], gamma=0.5, milestones=[200000]) for k in model}
    config = dict(generator_type="HiFiGANGenerator", generator_params=bench.HIFIGAN_V1, use_stft_loss=False, use_subband_stft_loss=False, use_mel_loss=True, use_feat_match_loss=True, lambda_aux=45.0, lambda_adv=1.0, lambda_feat_match=2.0, generator_grad_norm=-1, discriminator_grad_norm=-1, generator_train_start_steps=0, discriminator_train_start_steps=0, train_max_steps=10**9, save_interval_steps=10**9, eval_interval_steps=10**9, log_interval_steps=10**9, distributed=False, rank=0, outdir=tempfile.mkdtemp(), progress=False)
    b = args.train_batch
    c = torch.randn(b, 80, 32).to(dev); y = (0.3 * torch.randn(b, 1, 8192)).to(dev)
    batch = ((c,), y)
    tr = Trainer(steps=1, epochs=0, data_loader={"train": [batch], "dev": [batch]}, sampler={"train": None, "dev": None}, model=model, criterion=criterion, optimizer=opt, scheduler=sched, config=config, device=dev)
    tr.tqdm = None
    for _ in range(2): tr._train_step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); tr._train_step(batch); torch.cuda.synchronize(); print("step wall ms (unwrapped)", (time.perf_counter() - t0) * 1e3)
    for n in ("conv1d_forward", "conv1d_backward_data", "conv1d_backward_weight"): wrap(n)
    tr._train_step(batch); torch.cuda.synchronize()
patched_bench()
agg = collections.OrderedDict()
for name, d, e0, e1 in recs:
    ms = e0.elapsed_time(e1)
    k = (name, d)
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += ms
tot = sum(v[1] for v in agg.values())
print(f"total conv-launch time {tot:.1f} ms over {len(recs)} launches")
print("kind              B  Cin  Cout   Tin  Tout  W   K  s  d   g  T |  n     ms   ms/launch  TFLOP/s")
for (name, d), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    B, ci, co, ti, to, w, k, s, dil, g, tr_ = d
    fl = 2.0 * B * co * (ci // g) * k * (ti if tr_ else to) * w
    print(f"{name[7:]:16s} {B:2d} {ci:4d} {co:5d} {ti:5d} {to:5d} {w:2d} {k:3d} {s:2d} {dil:2d} {g:3d} {tr_:1d} | {n:2d} {ms:7.2f} {ms/n:8.3f}  {fl*n/ms/1e9:7.1f}")
