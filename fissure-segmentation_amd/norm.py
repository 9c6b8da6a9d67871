"""BatchNorm layers of the path.  MIOpen's batch-norm loses ~1e-2 absolute accuracy on channels whose
mean is large against their spread (measured on MI355X: 8.5e-3 vs 1.4e-5 for ATen's Welford kernels on
x ~ N(5, 0.05^2)), which breaks the 1e-4 parity bar of the deforming decoder; these subclasses keep the
parameter/buffer names and semantics of torch.nn.BatchNorm{1,2}d but call ATen's native kernels directly."""
import torch
from torch import nn


def _forward(self, x):
    self._check_input_dim(x)
    momentum = 0.0 if self.momentum is None else self.momentum
    if self.training and self.track_running_stats and self.num_batches_tracked is not None:
        from .functional import bump_bn_counter       # one multi-tensor add per forward inside deferred_bn_counters()
        momentum = bump_bn_counter(self)
    use_batch_stats = self.training or (self.running_mean is None and self.running_var is None)
    keep = not self.training or self.track_running_stats
    return torch.native_batch_norm(x, self.weight, self.bias, self.running_mean if keep else None,
                                   self.running_var if keep else None, use_batch_stats, momentum, self.eps)[0]


class BatchNorm1d(nn.BatchNorm1d):
    forward = _forward


class BatchNorm2d(nn.BatchNorm2d):
    forward = _forward
