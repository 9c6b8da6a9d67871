"""Cross-entropy + generalised Dice, the reference's default training criterion (losses/nnu_loss.py:6-19, selected by
losses/access_losses.py:47-48 for `--loss nnunet`), on the fused HIP loss kernel (value and gradient in two launches)."""
import torch
from torch import nn

from .. import functional as F_hip


class NNULoss(nn.Module):
    """Same constructor and return value as the reference: `(ce + dice, {'CE': ce, 'GDL': dice})`.

    `w_dice` / `w_ce` are stored and -- exactly as in losses/nnu_loss.py:16-19 -- not applied to the sum."""

    def __init__(self, class_weights, w_dice=1, w_ce=1):
        super().__init__()
        self.w_dice = w_dice
        self.w_ce = w_ce
        if class_weights is not None:
            self.register_buffer("class_weights", torch.as_tensor(class_weights, dtype=torch.float32), persistent=False)
        else:
            self.class_weights = None

    def forward(self, prediction, target):
        total, ce, gdl = F_hip.nnu_loss(prediction, target, self.class_weights)
        return total, {"CE": ce, "GDL": gdl}
