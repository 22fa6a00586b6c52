"""Feature matching loss (drop-in for parallel_wavegan.losses.feat_match_loss)."""
import torch

from .. import functional as Fn


class FeatureMatchLoss(torch.nn.Module):
    """Sum over discriminators and layers of L1(feat_hat, feat.detach()) with the reference's
    averaging switches (losses/feat_match_loss.py:12-54)."""

    def __init__(self, average_by_layers=True, average_by_discriminators=True, include_final_outputs=False):
        super().__init__()
        self.average_by_layers = average_by_layers
        self.average_by_discriminators = average_by_discriminators
        self.include_final_outputs = include_final_outputs

    def forward(self, feats_hat, feats):
        feat_match_loss = 0.0
        for i, (feats_hat_, feats_) in enumerate(zip(feats_hat, feats)):
            feat_match_loss_ = 0.0
            if not self.include_final_outputs:
                feats_hat_ = feats_hat_[:-1]
                feats_ = feats_[:-1]
            for j, (feat_hat_, feat_) in enumerate(zip(feats_hat_, feats_)):
                feat_match_loss_ = feat_match_loss_ + Fn.l1_mean(feat_hat_, feat_.detach())
            if self.average_by_layers:
                feat_match_loss_ = feat_match_loss_ / (j + 1)
            feat_match_loss = feat_match_loss + feat_match_loss_
        if self.average_by_discriminators:
            feat_match_loss = feat_match_loss / (i + 1)
        return feat_match_loss
