"""Point-cloud augmentation and sampling in front of the hot path (reference: augmentations.py:52-113, data.py:435-460).

The reference builds its transforms with pytorch3d (`so3_exp_map`, `Transform3d`), which is not vendored and unpinned:
the two pieces used are restated here from their published behaviour (parity unpinned) --
  * `so3_exp_map(v)`: Rodrigues, R = I + sin(t)/t K + (1-cos(t))/t^2 K^2, K = hat(v), t = sqrt(max(|v|^2, eps)), eps 1e-4;
  * `Transform3d`: homogeneous 4x4 matrices in ROW-vector convention (`points_h @ M`), composition in call order:
    `.rotate(R).scale(s).translate(t)` maps x -> (x R) * s + t.
Random numbers are drawn exactly as `point_augmentation` draws them (three `torch.rand` calls in the same order, same shapes).
Points on the GPU are moved by one `fsg_sample_transform_f32` launch (csrc/augment.hip); `sample_and_augment` is
`PointDataset.__getitem__` for a whole batch on the device: random subset + augmentation in that same launch.
"""
import ctypes

import torch


def so3_exp_map(log_rot, eps=1e-4):
    """(B,3) axis * angle -> (B,3,3) rotation matrices (pytorch3d.transforms.so3_exp_map)"""
    if log_rot.dim() != 2 or log_rot.shape[1] != 3:
        raise ValueError("Input tensor shape has to be Nx3.")
    nrms = (log_rot * log_rot).sum(1)
    theta = torch.clamp(nrms, eps).sqrt()
    fac1 = theta.sin() / theta
    fac2 = (1.0 - theta.cos()) / (theta * theta)
    K = torch.zeros(log_rot.shape[0], 3, 3, dtype=log_rot.dtype, device=log_rot.device)
    x, y, z = log_rot.unbind(1)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -z, y, z, -x, -y, x
    eye = torch.eye(3, dtype=log_rot.dtype, device=log_rot.device)[None]
    return eye + fac1[:, None, None] * K + fac2[:, None, None] * torch.bmm(K, K)


class Transform3d:
    """The subset of pytorch3d.transforms.Transform3d the reference uses: batched 4x4 matrices, row-vector convention."""

    def __init__(self, device="cpu", dtype=torch.float32, matrix=None):
        self._matrix = torch.eye(4, dtype=dtype, device=device)[None] if matrix is None else matrix

    def _then(self, m):
        return Transform3d(matrix=torch.matmul(self._matrix, m))

    def rotate(self, R):
        m = torch.eye(4, dtype=R.dtype, device=R.device).repeat(R.shape[0], 1, 1)
        m[:, :3, :3] = R
        return self._then(m)

    def scale(self, s):
        m = torch.eye(4, dtype=s.dtype, device=s.device).repeat(s.shape[0], 1, 1)
        m[:, 0, 0], m[:, 1, 1], m[:, 2, 2] = s[:, 0], s[:, 1], s[:, 2]
        return self._then(m)

    def translate(self, t):
        m = torch.eye(4, dtype=t.dtype, device=t.device).repeat(t.shape[0], 1, 1)
        m[:, 3, :3] = t
        return self._then(m)

    def compose(self, *others):
        out = self
        for o in others:
            out = out._then(o.get_matrix())
        return out

    def inverse(self):
        return Transform3d(matrix=torch.linalg.inv(self._matrix))

    def get_matrix(self):
        return self._matrix

    def __len__(self):
        return self._matrix.shape[0]

    def affine_rows(self):
        """(B,12) row-major [A | t] of the column-vector form x' = A x + t (A = M[:3,:3]^T, t = M[3,:3])"""
        m = self._matrix
        return torch.cat([m[:, :3, :3].transpose(1, 2), m[:, 3, :3, None]], dim=2).reshape(-1, 12).contiguous()

    def transform_points(self, points):
        """points (B,N,3) or (N,3) on the GPU -> same shape, one fsg_sample_transform_f32 launch"""
        squeeze = points.dim() == 2
        p = points[None] if squeeze else points
        if self._matrix.shape[0] not in (1, p.shape[0]):
            raise ValueError(f"{self._matrix.shape[0]} transforms for {p.shape[0]} point sets")
        out = _sample_transform(p.transpose(1, 2), None, self.affine_rows().to(p.device).expand(p.shape[0], 12)
                                ).transpose(1, 2)     # raises on CPU tensors: no CPU fallback
        return out[0] if squeeze else out


def _sample_transform(x, sample, affine):
    """x (B,C,N) on the GPU -> (B,C,S): column subset (`sample` (B,S) int64 or None) and/or affine map of rows 0..2"""
    from . import _lib
    if not x.is_cuda:
        raise RuntimeError("the HIP sampling/augmentation path needs its input on the GPU")
    x = x.to(torch.float32).contiguous()
    B, C, N = x.shape
    S = N if sample is None else sample.shape[1]
    if sample is not None:
        sample = sample.to(device=x.device, dtype=torch.int64).contiguous()
        if sample.shape[0] != B:
            raise ValueError(f"sample must be (B,S), got {tuple(sample.shape)} for B={B}")
    if affine is not None:
        affine = affine.to(device=x.device, dtype=torch.float32).contiguous()
        if affine.shape != (B, 12):
            raise ValueError(f"affine must be (B,12), got {tuple(affine.shape)}")
    out = torch.empty(B, C, S, dtype=torch.float32, device=x.device)
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    with torch.cuda.device(x.device):
        _lib.call("fsg_sample_transform_f32", P(x), B, C, N, P(sample), S, P(affine), P(out),
                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    return out


def compose_transform(log_rotation_matrix, translation, scaling):
    """augmentations.py:78-88: rotate (axis-angle), then scale, then translate"""
    return Transform3d(device=log_rotation_matrix.device).rotate(so3_exp_map(log_rotation_matrix)) \
        .scale(scaling.expand(-1, 3)).translate(translation)


def transform_points(point_clouds, transforms):
    """augmentations.py:103-113: (B,3,N) rotated around the origin"""
    assert point_clouds.ndim == 3
    return transforms.transform_points(point_clouds.transpose(1, 2)).transpose(1, 2)


def transform_points_with_centering(point_clouds, transforms):
    """augmentations.py:91-100"""
    assert point_clouds.ndim == 3
    center = point_clouds.mean(2, keepdim=True)
    return transform_points(point_clouds - center, transforms) + center


def random_transform_parameters(n, device, rotation_amount=0.1, translation_amount=0.1, scale_amount=0.1):
    """the three draws of augmentations.py:62-73, in the same order"""
    vec = torch.rand(n, 3, device=device) * 2 - 1
    log_rot = vec / vec.norm(dim=1, keepdim=True) * torch.pi * rotation_amount
    translation = (torch.rand(n, 3, device=device) * 2 - 1) * translation_amount
    rescale = torch.ones(n, 1, device=device) - torch.rand(n, 1, device=device) * scale_amount
    return log_rot, translation, rescale


def point_augmentation(point_clouds, rotation_amount=0.1, translation_amount=0.1, scale_amount=0.1):
    """augmentations.py:52-75: random rotation (|angle| = rotation_amount*pi around a random axis), translation in
    [-translation_amount, translation_amount]^3, uniform scale in [1 - scale_amount, 1]; returns (points, transform)"""
    transforms = compose_transform(*random_transform_parameters(len(point_clouds), point_clouds.device, rotation_amount,
                                                                translation_amount, scale_amount))
    return transform_points(point_clouds, transforms), transforms


def random_subsets(B, N, S, device):
    """B independent uniformly random S-subsets of range(N), in random order (the batch form of data.py:449
    `torch.randperm(N)[:sample_points]`): ranks of one (B,N) draw of random keys -- one sort instead of B permutations"""
    return torch.rand(B, N, device=device).argsort(dim=1)[:, :S].contiguous()


def sample_and_augment(x, labels=None, sample_points=None, augment=True, binary=False, use_coords=True):
    """`PointDataset.__getitem__` (data.py:435-460) for a batch that already lives on the GPU: per item an independent
    augmentation of the coordinate rows 0..2 and an independent random subset of `sample_points` columns, in one launch.
    x (B,C,N); labels (B,N) or None -> (x_sampled (B,C,S), labels_sampled (B,S) or None, transforms or None)."""
    B, _, N = x.shape
    transforms = None
    if augment and use_coords:
        transforms = compose_transform(*random_transform_parameters(B, x.device))
    S = N if sample_points is None else min(sample_points, N)
    sample = random_subsets(B, N, S, x.device)
    out = _sample_transform(x, sample, None if transforms is None else transforms.affine_rows())
    lbl = None
    if labels is not None:
        lbl = torch.gather(labels, 1, sample)
        if binary:
            lbl = (lbl != 0).long()
    return out, lbl, transforms
