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


def farthest_point_sampling(kpts, num_points, start=None):
    """Drop-in for the pure-torch loop of dseg_ae_regularization.py:30-43 (used at :85 to pick the auto-encoder's input
    points of one object): `kpts (1,N,3)` -> `(kpts[:, ind, :], ind)` with `ind (num_points,) int64`.  The reference's
    `torch.argmax(dist)` runs over the flattened (B,N) distances, so it is only meaningful -- and only ever called -- with
    one cloud; B > 1 raises here.

    One `fsg_fps_f32` launch instead of `num_points` arg-max round trips.  The reference starts at a random point
    (`torch.randint(N, (1,))`); the kernel starts a segment at its first row, so the cloud is rotated to put the start
    there (`start` fixes it for tests).  Distances are the direct form (x-y)^2 as in the reference; ties (measure zero on
    real data) go to the first point after the start instead of the lowest index."""
    B, N, _ = kpts.size()
    if N <= num_points:
        if N < num_points:
            print(f'Tried to sample {num_points} from a point cloud with only {N}')
        return kpts, torch.arange(N)
    if B != 1:
        raise ValueError("farthest_point_sampling: one cloud at a time (the reference shares one index list and is "
                         "called with B = 1)")
    if not kpts.is_cuda:
        raise RuntimeError("farthest_point_sampling (HIP path) needs its input on the GPU")
    if start is None:
        start = int(torch.randint(N, (1,)))
    pts = torch.roll(kpts[0].to(torch.float32), -start, 0).contiguous()
    offset = torch.tensor([N], dtype=torch.int32, device=kpts.device)
    new_offset = torch.tensor([num_points], dtype=torch.int32, device=kpts.device)
    ind = (F_hip.fps(pts, offset, new_offset, num_points).long() + start) % N
    return kpts[:, ind, :], ind
