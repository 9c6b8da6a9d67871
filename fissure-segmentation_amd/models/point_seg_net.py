"""Base class of the point segmentation nets (reference: models/point_seg_net.py:10-48)."""
import ctypes
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

    #: clouds per forward of the batched ensembling (40 + 10 runs of a 2048-point subset fit comfortably)
    ensemble_max_clouds = 64

    def _ensemble_batchable(self, pc):
        """The runs of the ensembling are independent clouds iff nothing couples the samples of a batch: eval mode with
        every BatchNorm on its running statistics.  Under autograd the sequential form is kept too."""
        if self.training or torch.is_grad_enabled() or not pc.is_cuda or pc.dim() != 3 or pc.dtype != torch.float32:
            return False
        if self.num_classes > 32:
            return False
        for m in self.modules():
            if isinstance(m, torch.nn.modules.batchnorm._NormBase) and (m.training or m.running_mean is None):
                return False
            if isinstance(m, torch.nn.modules.dropout._DropoutNd) and m.training:
                return False
        return True

    def _ensemble_pass(self, pc, pts, acc):
        """acc[..., pts[r]] += softmax(self(pc[..., pts[r]])) for all runs r, in run order: the subsets go through the net
        as one batch (chunks of `ensemble_max_clouds`), the accumulation is one call of fsg_ensemble_accumulate_f32"""
        from .. import _lib
        B, n_pts = pc.shape[0], pc.shape[-1]
        per = max(1, self.ensemble_max_clouds // max(B, 1))
        for r0 in range(0, pts.shape[0], per):
            chunk = pts[r0:r0 + per].contiguous()
            R, S = chunk.shape
            x = pc[:, :, chunk].permute(2, 0, 1, 3).reshape(R * B, pc.shape[1], S)    # run-major batch of subsets
            logits = self(x).contiguous()
            if logits.shape != (R * B, self.num_classes, S) or logits.dtype != torch.float32:
                raise RuntimeError(f"unexpected logits {tuple(logits.shape)} {logits.dtype} from {type(self).__name__}")
            ws = torch.empty(_lib.lib.fsg_ensemble_accumulate_workspace_bytes(R, n_pts) // 4, dtype=torch.int32,
                             device=pc.device)
            P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
            _lib.call("fsg_ensemble_accumulate_f32", P(logits), R, B, self.num_classes, S, P(chunk), n_pts, P(acc), P(ws),
                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))

    def predict_full_pointcloud(self, pc, sample_points=1024, n_runs_min=50):
        """Test-time ensembling over random point subsets (point_seg_net.py:21-48): 4/5 of the runs
        on random `sample_points`-subsets, the rest mix still-unseen points with seen ones.

        In eval mode (how train.py:213-214,383-386 call it) the runs are drawn exactly like the reference's loop draws
        them -- same generator calls in the same order -- but go through the net as one batch per phase; otherwise the
        sequential form below runs."""
        n_fill = n_runs_min // 5
        n_first = n_runs_min - n_fill
        n_pts = pc.shape[-1]
        batched = self._ensemble_batchable(pc)
        acc = torch.zeros(pc.shape[0], self.num_classes, *pc.shape[2:], device=pc.device)
        if batched:
            if n_first > 0:
                self._ensemble_pass(pc, torch.stack([torch.randperm(n_pts, device=pc.device)[:sample_points]
                                                     for _ in range(n_first)]), acc)
        else:
            for _ in range(n_first):
                pts = torch.randperm(n_pts, device=pc.device)[:sample_points]
                acc[..., pts] += torch.softmax(self(pc[..., pts]), dim=1)

        unseen = torch.nonzero(acc.sum(1) == 0)[..., 1]
        print(f'After {n_first} runs, {unseen.shape[0]} points have not been seen yet.')
        if unseen.shape[0] > 0:
            seen = torch.nonzero(acc.sum(1))[..., 1]
            n_mix = sample_points // 2
            pick = torch.randperm(n_fill * n_mix, device=pc.device) % len(unseen)
            runs = []
            for r in range(n_fill):
                lo = unseen[pick[r * n_mix:(r + 1) * n_mix]]
                rest = torch.randperm(len(seen), device=pc.device)[:sample_points - n_mix]
                pts = torch.cat((lo, rest), dim=0)
                if batched:
                    runs.append(pts)
                else:
                    acc[..., pts] += torch.softmax(self(pc[..., pts]), dim=1)
            if runs:
                self._ensemble_pass(pc, torch.stack(runs), acc)
            if (acc.sum(1) == 0).any():
                warnings.warn('NOT ALL POINTS HAVE BEEN SEEN')
        return torch.softmax(acc, dim=1)
