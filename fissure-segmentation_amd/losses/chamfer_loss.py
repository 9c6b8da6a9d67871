"""Chamfer loss on the HIP nearest-neighbour kernel (reference: losses/chamfer_loss.py:5-20, which
calls pytorch3d.loss.chamfer_distance with its defaults: squared L2, mean over points, both
directions added, mean over the batch -- train_pc_ae.py:85)."""
from torch import nn

from .. import functional as F_hip


def chamfer_distance(x, y):
    """x (B,N,3), y (B,M,3) -> (loss, None) like pytorch3d's (loss, loss_normals)."""
    d_xy, _ = F_hip.chamfer_nn(x, y)
    d_yx, _ = F_hip.chamfer_nn(y, x)
    return d_xy.mean(1).mean() + d_yx.mean(1).mean(), None


class ChamferLoss(nn.Module):
    def forward(self, prediction, target):
        if prediction.shape[1] == 3:  # (B, 3, N) layout
            prediction = prediction.transpose(1, 2)
        if target.shape[1] == 3:
            target = target.transpose(1, 2)
        assert prediction.shape[0] == target.shape[0] and prediction.shape[2] == target.shape[2]
        return chamfer_distance(prediction, target)[0]
