"""Drop-in for the kNN part of the reference's utils/general_utils.py (:43-53, :315-327)."""
import torch

from .. import functional as F_hip


def knn(x, k, self_loop=False, return_dist=False):
    """Same contract as utils/general_utils.py:315 -- x: (B,C,N) -> idx (B,N,k) int64, neighbours in
    ascending distance; `self_loop=False` selects k+1 and drops the first column (:317,:320-322).
    The (B,N,N) matrix of `pairwise_dist` is never materialised (fused in fsg_knn_dense_f32)."""
    out = F_hip.knn_graph(x, k, fix_diag=True, drop_first=not self_loop, return_dist=return_dist)
    if return_dist:
        return out[0].long(), out[1]
    return out.long()


def pairwise_dist(x):
    """utils/general_utils.py:43-53 for callers that really want the dense (B,N,N) matrix (none on the
    hot path).  Plain device-side torch ops; kept only for API completeness."""
    if not x.is_cuda:
        raise RuntimeError("GPU tensors only")
    sq = x.pow(2).sum(2, keepdim=True)
    d = sq - 2.0 * torch.bmm(x, x.transpose(1, 2)) + sq.transpose(1, 2)
    ar = torch.arange(x.shape[1], device=x.device)
    d[:, ar, ar] = 0
    return d
