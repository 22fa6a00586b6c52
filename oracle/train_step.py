"""CPU restatement of ``Trainer._train_step`` (/root/reference/parallel_wavegan/bin/train.py:189-340)
for the HiFi-GAN configuration, on top of oracle.torch_cpu functionals + torch autograd + torch.optim.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Parameters live in plain dicts of leaf tensors
(reference state_dict keys); pinned against two steps of the reference's own Trainer by
tests/test_oracle_golden.py (fixture tests/golden/hifigan_v1_train.npz).
"""
import torch

from . import torch_cpu as O


class HiFiGANTrainState:
    def __init__(self, sd_g, sd_d, gen_params, dis_params, mel_params, lr=2.0e-4, betas=(0.5, 0.9),
                 lambda_aux=45.0, lambda_adv=1.0, lambda_feat_match=2.0):
        self.gen_params, self.dis_params = dict(gen_params), dict(dis_params)
        self.mel_params = {k: v for k, v in mel_params.items() if k != "window"}
        self.lambda_aux, self.lambda_adv, self.lambda_fm = lambda_aux, lambda_adv, lambda_feat_match
        buf = ("weight_u",)
        self.g = {k: v.clone().requires_grad_(True) for k, v in sd_g.items()}
        self.d = {}
        for k, v in sd_d.items():
            is_buf = k.endswith(buf) or (k.endswith("weight_v") and v.dim() == 1)
            self.d[k] = v.clone() if is_buf else v.clone().requires_grad_(True)
        self.opt_g = torch.optim.Adam([p for p in self.g.values() if p.requires_grad], lr=lr, betas=betas)
        self.opt_d = torch.optim.Adam([p for p in self.d.values() if p.requires_grad], lr=lr, betas=betas)

    def generator(self, c):
        return O.hifigan_generator(self.g, c, **self.gen_params)

    def discriminator(self, x):
        return O.hifigan_msmpd(self.d, x, training=True, **self.dis_params)

    def step(self, c, y):
        """One optimisation step with both phases active; returns the logged loss values."""
        log = {}
        # ---- generator (train.py:200-295)
        y_ = self.generator(c)
        mel_loss = O.mel_spectrogram_loss(y_, y, **self.mel_params)
        gen_loss = self.lambda_aux * mel_loss
        p_ = self.discriminator(y_)
        adv_loss = O.generator_adversarial_loss(p_, average_by_discriminators=False)
        with torch.no_grad():
            p = self.discriminator(y)
        fm_loss = O.feature_match_loss(p_, p, average_by_layers=False, average_by_discriminators=False)
        gen_loss = gen_loss + self.lambda_adv * (adv_loss + self.lambda_fm * fm_loss)
        log.update({"train/mel_loss": mel_loss.item(), "train/adversarial_loss": adv_loss.item(),
                    "train/feature_matching_loss": fm_loss.item(), "train/generator_loss": gen_loss.item()})
        self.opt_g.zero_grad()
        self.opt_d.zero_grad()
        gen_loss.backward()
        self.opt_g.step()
        # ---- discriminator (train.py:300-335)
        with torch.no_grad():
            y_ = self.generator(c)
        p = self.discriminator(y)
        p_ = self.discriminator(y_.detach())
        real_loss, fake_loss = O.discriminator_adversarial_loss(p_, p, average_by_discriminators=False)
        dis_loss = real_loss + fake_loss
        log.update({"train/real_loss": real_loss.item(), "train/fake_loss": fake_loss.item(),
                    "train/discriminator_loss": dis_loss.item()})
        self.opt_d.zero_grad()
        dis_loss.backward()
        self.opt_d.step()
        return log
