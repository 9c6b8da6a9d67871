"""kNN on clouds with many duplicated points (the fill-up phase of predict_full_pointcloud draws unseen points with
replacement): exact parity with the oracle expected."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import fissure_segmentation_amd as fsg
from oracle import c_api
dev = torch.device("cuda:0")
for seed in range(6):
    rng = np.random.default_rng(seed)
    B, N, k = 2, 256, 8
    base = rng.uniform(-1, 1, (B, 3, 30 + 10 * seed)).astype(np.float32)
    pick = rng.integers(0, base.shape[2], (N,))
    x = base[:, :, pick].copy()
    x[:, :, 128:] = rng.uniform(-1, 1, (B, 3, 128)).astype(np.float32)
    xt = torch.from_numpy(x).to(dev)
    print("seed", seed, "distinct points in first half:", len(set(pick[:128].tolist())), flush=True)
    idx = fsg.functional.knn_graph(xt, k, c_knn=3, fix_diag=True)
    torch.cuda.synchronize()
    want = c_api.knn_dense(x, k, fix_diag=True)[0] if hasattr(c_api, "knn_dense") else None
    print("  ok", tuple(idx.shape), None if want is None else bool((idx.cpu().numpy() == want).all()), flush=True)
