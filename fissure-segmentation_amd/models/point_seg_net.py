"""Base class of the point segmentation nets (reference: models/point_seg_net.py:10-48)."""
import warnings
from abc import ABC, abstractmethod

import torch

from .modelio import LoadableModel, store_config_args


class PointSegmentationModelBase(LoadableModel, ABC):
    @store_config_args
    def __init__(self, in_features, num_classes, **kwargs):
        super().__init__()
        self.in_features = in_features
        self.num_classes = num_classes

    @abstractmethod
    def forward(self, x):
        ...

    def predict_full_pointcloud(self, pc, sample_points=1024, n_runs_min=50):
        """Test-time ensembling over random point subsets (point_seg_net.py:21-48): 4/5 of the runs
        on random `sample_points`-subsets, the rest mix still-unseen points with seen ones."""
        n_fill = n_runs_min // 5
        n_first = n_runs_min - n_fill
        n_pts = pc.shape[-1]
        acc = torch.zeros(pc.shape[0], self.num_classes, *pc.shape[2:], device=pc.device)
        for _ in range(n_first):
            pts = torch.randperm(n_pts, device=pc.device)[:sample_points]
            acc[..., pts] += torch.softmax(self(pc[..., pts]), dim=1)

        unseen = torch.nonzero(acc.sum(1) == 0)[..., 1]
        print(f'After {n_first} runs, {unseen.shape[0]} points have not been seen yet.')
        if unseen.shape[0] > 0:
            seen = torch.nonzero(acc.sum(1))[..., 1]
            n_mix = sample_points // 2
            pick = torch.randperm(n_fill * n_mix, device=pc.device) % len(unseen)
            for r in range(n_fill):
                lo = unseen[pick[r * n_mix:(r + 1) * n_mix]]
                rest = torch.randperm(len(seen), device=pc.device)[:sample_points - n_mix]
                pts = torch.cat((lo, rest), dim=0)
                acc[..., pts] += torch.softmax(self(pc[..., pts]), dim=1)
            if (acc.sum(1) == 0).any():
                warnings.warn('NOT ALL POINTS HAVE BEEN SEEN')
        return torch.softmax(acc, dim=1)
