"""Feature matching loss (drop-in for parallel_wavegan.losses.feat_match_loss).

The reference walks discriminators x layers in Python and issues one ``F.l1_loss`` per feature map
(/root/reference/parallel_wavegan/losses/feat_match_loss.py:36-54; ~45 maps for HiFi-GAN's MSD+MPD).
Here the whole loss is ONE multi-tensor reduction (``pwg_multi_reduce_*``): every feature-map pair is
an item whose weight folds the map's 1/numel (the L1 mean), the per-discriminator 1/#layers and the
1/#discriminators averaging, so the result is a single device scalar from a single launch pair.
"""
import torch

from .. import functional as Fn


class FeatureMatchLoss(torch.nn.Module):
    def __init__(self, average_by_layers=True, average_by_discriminators=True, include_final_outputs=False):
        super().__init__()
        self.average_by_layers = average_by_layers
        self.average_by_discriminators = average_by_discriminators
        self.include_final_outputs = include_final_outputs

    def weighted_pairs(self, feats_hat, feats):
        """[(feat_hat, feat, weight)] such that the loss is sum weight * mean|feat_hat - feat|."""
        drop_last = 0 if self.include_final_outputs else 1
        disc_w = 1.0 / len(feats_hat) if self.average_by_discriminators else 1.0
        out = []
        for maps_hat, maps in zip(feats_hat, feats):
            n_maps = min(len(maps_hat), len(maps)) - drop_last
            w = disc_w / n_maps if self.average_by_layers else disc_w
            # deferred-activation form (layers.activation.PreActivated): every map but the last (the logits) is the
            # convolution output BEFORE its LeakyReLU; the reduction kernel applies it to both operands
            s_hat, s = getattr(maps_hat, "preact_slope", None), getattr(maps, "preact_slope", None)
            if s_hat != s:
                raise ValueError("feature maps and targets must both be in (or both out of) pre-activation form")
            for k in range(n_maps):
                pre = s_hat if (s_hat is not None and k < len(maps_hat) - 1) else None
                out.append((maps_hat[k], maps[k], w, pre))
        return out

    def forward(self, feats_hat, feats):
        """feats_hat / feats: list (discriminators) of lists (layers) of tensors; the targets ``feats``
        are treated as constants (the reference detaches them)."""
        spec, tensors = [], []
        for f_hat, f, w, pre in self.weighted_pairs(feats_hat, feats):
            if pre is None:
                spec.append(("abs_diff", w / f_hat.numel(), 0.0, 0))
            else:
                spec.append(("abs_diff_lrelu", w / f_hat.numel(), pre, 0))
            tensors += [f_hat, f.detach()]
        return Fn.MultiReduceFn.apply(spec, 1, *tensors)[0]
