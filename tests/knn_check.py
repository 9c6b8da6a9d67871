"""Tie-aware comparison of kNN index sets (shared by the CPU and GPU parity tests).

Two fp32 implementations of d = xx_i - 2 x_i.x_j + xx_j (MKL bmm in the reference, a k-ordered fma
chain in the oracle / HIP kernels) can order two neighbours differently only when their distances
differ by less than the rounding error of the expanded form.  `well_separated_rows` marks the rows
where every consecutive gap among the k+2 nearest exact (fp64) distances exceeds that bound; on those
rows indices must be IDENTICAL; on the others the sorted exact distances of the two index lists must
agree within the bound."""
import numpy as np


def exact_dist(x, c_knn=None, fix_diag=True):
    x = np.asarray(x, np.float64)[:, :c_knn]
    sq = (x * x).sum(1)
    d = sq[:, :, None] - 2.0 * np.einsum("bci,bcj->bij", x, x) + sq[:, None, :]
    if fix_diag:
        i = np.arange(d.shape[1])
        d[:, i, i] = 0.0
    return d


def tol_rows(x, c_knn=None):
    x = np.asarray(x, np.float64)[:, :c_knn]
    C = x.shape[1]
    sq = (x * x).sum(1)
    return 16.0 * np.sqrt(C) * 6e-8 * (sq + sq.max(1, keepdims=True))  # (B, N)


def well_separated_rows(d, k, tol):
    srt = np.sort(d, -1)[..., :k + 2]
    return (np.diff(srt, axis=-1) > tol[..., None]).all(-1)


def assert_knn_equal(x, idx_a, idx_b, k, c_knn=None, fix_diag=True, drop_first=False, min_exact_frac=0.5):
    """idx_*: (B,N,k) integer arrays.  Returns the fraction of rows compared bit-exactly."""
    d = exact_dist(x, c_knn, fix_diag)
    tol = tol_rows(x, c_knn)
    sep = well_separated_rows(d, k + (1 if drop_first else 0), tol)
    a, b = np.asarray(idx_a, np.int64), np.asarray(idx_b, np.int64)
    assert a.shape == b.shape
    bad = (a != b).any(-1) & sep
    assert not bad.any(), f"{bad.sum()} well-separated rows differ, first at {np.argwhere(bad)[:3].tolist()}"
    da = np.sort(np.take_along_axis(d, a, -1), -1)
    db = np.sort(np.take_along_axis(d, b, -1), -1)
    assert (np.abs(da - db) <= tol[..., None]).all(), "near-tie rows: distances of the two neighbour lists differ"
    frac = float(sep.mean())
    assert frac >= min_exact_frac, f"only {frac:.2%} rows well separated -- fixture too degenerate"
    return frac
