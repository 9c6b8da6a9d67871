"""PointNet segmentation (BASELINE config 1, the reference's CPU-runnable plumbing case):
models/point_net.py:11-100.  Pure point-wise Conv1d/BN stacks -- there is no neighbourhood op, hence
no HIP kernel on this model; it runs wherever its tensors live."""
import torch
from torch import nn

from .. import functional as F_hip
from ..norm import BatchNorm1d
from ..utils.model_utils import init_weights
from .point_seg_net import PointSegmentationModelBase


class MLPBlock(nn.Module):
    """Conv1d(no bias) -> BatchNorm1d -> LeakyReLU(0.01), repeated (point_net.py:11-30)."""

    def __init__(self, in_channel, num_neurons_list):
        super().__init__()
        mods, prev = [], in_channel
        for width in num_neurons_list:
            mods += [nn.Conv1d(prev, width, 1, bias=False), BatchNorm1d(width), nn.LeakyReLU()]
            prev = width
        self.layers = nn.ModuleList(mods)

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
        return x


class PointNetSeg(PointSegmentationModelBase):
    """point_net.py:55-100.  The T-Net branches of the reference are broken (its TNet.last_layer takes 32
    channels but receives 256, and forward refers to an undefined `tnet_feat`), so only the default
    configuration exists here and the flags raise."""

    def __init__(self, in_features, num_classes, spatial_transform=False, feature_transform=False, **kwargs):
        super().__init__(in_features, num_classes, spatial_transform=spatial_transform,
                         feature_transform=feature_transform)
        if spatial_transform or feature_transform:
            raise NotImplementedError("PointNetSeg: the T-Net variants do not run in the reference either")
        self.t_net_coord = None
        self.t_net_feat = None
        self.local_features = MLPBlock(in_features, [64, 64])
        self.global_features = nn.Sequential(MLPBlock(64, [64, 128, 1024]), nn.AdaptiveMaxPool1d(1))
        self.seg_branch = nn.Sequential(MLPBlock(64 + 1024, [256, 128, 64, 64]),
                                        nn.Conv1d(64, num_classes, 1, bias=True))
        self.init_weights()

    @F_hip.with_deferred_bn_counters
    def forward(self, x):
        local = self.local_features(x)
        glob = self.global_features(local)
        return self.seg_branch(torch.cat([local, glob.expand(-1, -1, local.shape[-1])], dim=1))

    def init_weights(self):
        self.apply(init_weights)
