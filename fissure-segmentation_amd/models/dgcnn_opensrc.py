"""Graph feature of the upstream DGCNN code path used by the PC-AE encoder
(reference: models/dgcnn_opensrc.py:34-66), on the HIP kernels."""
from .. import functional as F_hip


def knn(x, k):
    """(B,C,N) -> (B,N,k) int64: the k nearest points INCLUDING the point itself, no forced-zero
    diagonal -- dgcnn_opensrc.py:34-40 ranks by the negated distance with topk(largest)."""
    return F_hip.knn_graph(x, k, fix_diag=False).long()


def get_graph_feature(x, k=20, idx=None):
    """(B,C,N) -> (B,2C,N,k) = cat(x_j - x_i, x_i)   (dgcnn_opensrc.py:43-66)."""
    B, N = x.size(0), x.size(2)
    x = x.reshape(B, -1, N)
    if idx is None:
        idx = F_hip.knn_graph(x, k, fix_diag=False)
    return F_hip.edge_features(x, idx)
