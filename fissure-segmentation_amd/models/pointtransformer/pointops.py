"""PointTransformer point operations on the HIP kernels -- the role `pointops_cuda` plays behind the
reference's models/pointtransformer/pointops.py.  Packed layout: xyz (n,3), feat (n,c), `offset` (b)
int32 cumulative segment ends.  Same function names and argument order as the reference."""
import torch

from ... import functional as F_hip

_knn_memo = None  # {(xyz ptr, new_xyz ptr, offsets ptr, nsample): (idx, dist)}, alive inside knn_cache()


class knn_cache:
    """Within one forward the reference queries the same (points, nsample) graph again and again (twice
    per layer, seg_model.py:38-39, and once per block of a level).  Inside this context the query
    runs once per distinct (xyz, new_xyz, offset, nsample)."""

    def __enter__(self):
        global _knn_memo
        self._outer, _knn_memo = _knn_memo, {}
        return self

    def __exit__(self, *exc):
        global _knn_memo
        _knn_memo = self._outer


_const_memo = {}


def device_ints(values, device, dtype=torch.int32):
    """Small constant integer tensor (segment ends, counts) on the device.  Memoised by value: the host-to-device copy
    happens once, so a training step with static shapes issues no copy at all and can be captured into a hipGraph."""
    key = (tuple(int(v) for v in values), str(device), dtype)
    t = _const_memo.get(key)
    if t is None:
        if len(_const_memo) > 4096:
            _const_memo.clear()
        t = torch.tensor(key[0], device=device, dtype=dtype)
        _const_memo[key] = t
    return t


def segment_ids(ends, device):
    """(n) int64: segment number of every packed row (memoised like device_ints)."""
    key = ("seg", tuple(int(v) for v in ends), str(device))
    t = _const_memo.get(key)
    if t is None:
        counts = device_ints([e - s for s, e in zip([0] + list(ends[:-1]), ends)], device, torch.int64)
        t = torch.repeat_interleave(torch.arange(len(ends), device=device), counts, output_size=int(ends[-1]))
        _const_memo[key] = t
    return t


def segment_mean_matrices(ends, device):
    """(A (b, n), S (n, b)) fp32, memoised like device_ints: A x = per-segment mean of the packed rows x, S m = the segment's row
    of m repeated for each of its points.  Two small products each way instead of index_add / divide / gather and their index
    backward (a sort among them) -- the head of the decoder works on a few dozen rows (seg_model.py:101-113 of the reference)."""
    key = ("segmean", tuple(int(v) for v in ends), str(device))
    t = _const_memo.get(key)
    if t is None:
        seg = segment_ids(ends, device)
        S = torch.zeros(int(ends[-1]), len(ends), device=device, dtype=torch.float32)
        S[torch.arange(int(ends[-1]), device=device), seg] = 1.0
        A = (S / S.sum(0, keepdim=True).clamp_min(1.0)).t().contiguous()
        t = (A, S)
        _const_memo[key] = t
    return t


def host_offsets(o):
    """Segment ends as a Python list without a device sync when the producer attached them."""
    cached = getattr(o, "_fsg_host", None)
    if cached is None:
        cached = [int(v) for v in o.tolist()]
        o._fsg_host = cached
    return cached


def with_host_offsets(o, values):
    o._fsg_host = [int(v) for v in values]
    return o


def furthestsampling(xyz, offset, new_offset):
    """pointops.py:16-39 -> idx (m) int32."""
    assert xyz.is_contiguous()
    return F_hip.fps(xyz, offset, new_offset, host_offsets(new_offset)[-1])


def knn_squared(nsample, xyz, new_xyz, offset, new_offset):
    """the query behind knnquery -> (idx (m,nsample) int32, SQUARED distance (m,nsample)); memoised inside knn_cache().
    Callers that only want the indices (the attention layers, queryandgroup) use this and skip the square-root launch."""
    if new_xyz is None:
        new_xyz = xyz
    assert xyz.is_contiguous() and new_xyz.is_contiguous()
    key = (xyz.data_ptr(), new_xyz.data_ptr(), offset.data_ptr(), new_offset.data_ptr(), xyz.shape[0],
           new_xyz.shape[0], nsample)
    if _knn_memo is not None and key in _knn_memo:
        return _knn_memo[key]
    out = F_hip.knn_segment(nsample, xyz, new_xyz, offset, new_offset)
    if _knn_memo is not None:
        _knn_memo[key] = out
    return out


def knnquery(nsample, xyz, new_xyz, offset, new_offset):
    """pointops.py:42-62 -> (idx (m,nsample) int32, distance (m,nsample)); not differentiable."""
    idx, d2 = knn_squared(nsample, xyz, new_xyz, offset, new_offset)
    return idx, torch.sqrt(d2)


def grouping(input, idx):
    """pointops.py:65-97: (n,c), (m,nsample) -> (m,nsample,c); backward scatter-adds."""
    return F_hip.group_gather(input, idx)


def queryandgroup(nsample, xyz, new_xyz, feat, idx, offset, new_offset, use_xyz=True):
    """pointops.py:100-123 -> (m,nsample,3+c) or (m,nsample,c)."""
    if new_xyz is None:
        new_xyz = xyz
    assert xyz.is_contiguous() and new_xyz.is_contiguous() and feat.is_contiguous()
    if idx is None:
        idx, _ = knn_squared(nsample, xyz, new_xyz, offset, new_offset)
    if use_xyz and feat.is_cuda and not (xyz.requires_grad or new_xyz.requires_grad):
        return F_hip.group_xyz_feat(xyz, new_xyz, feat, idx)      # gather, gather, subtract, concatenate: one launch
    grouped_feat = grouping(feat, idx)
    if not use_xyz:
        return grouped_feat
    grouped_xyz = grouping(xyz, idx) - new_xyz.unsqueeze(1)
    return torch.cat((grouped_xyz, grouped_feat), -1)


def aggregation(input, position, weight, idx):
    """pointops.py:161-195: out[i,c] = sum_j (input[idx[i,j],c] + position[i,j,c]) * weight[i,j,c % c']."""
    return F_hip.vec_attn(input, position, weight, idx)


def interpolation(xyz, new_xyz, feat, offset, new_offset, k=3):
    """pointops.py:198-215: inverse-distance interpolation from the k nearest coarse points."""
    assert xyz.is_contiguous() and new_xyz.is_contiguous() and feat.is_contiguous()
    idx, d2 = knn_squared(k, xyz, new_xyz, offset, new_offset)
    if k <= 8:       # one launch each way (fsg_interp_fwd_f32 / fsg_interp_bwd_f32) instead of eight element-wise / reduce ones
        return F_hip.interpolate(feat, idx, d2)
    w = 1.0 / (torch.sqrt(d2) + 1e-8)
    w = w / w.sum(dim=1, keepdim=True)
    return (grouping(feat, idx) * w.unsqueeze(-1)).sum(dim=1)
