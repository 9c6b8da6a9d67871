"""Golden vector for the Chamfer term of RegularizedMeshLoss (reference: losses/mesh_loss.py:24-33) at the size the reference
trains with: 2048 surface samples per mesh.  pytorch3d (sample_points_from_meshes, chamfer_distance) is absent from the
reference tree, so -- like chamfer_s701 in make_golden.py -- the value is the reference's own `pairwise_dist2`
(utils/general_utils.py:56-67) under pytorch3d's documented defaults (train_pc_ae.py:85); the samples are data of this script:
two noisy height fields over the unit square, the kind of surface `train_pc_ae.py --loss mesh` compares.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_mesh.py        (build container only: needs /root/reference)
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, os.path.join(REPO, "tests"))


def surface_samples(seed, B=2, n=2048):
    """(B, n, 3) fp32: uniform samples of z = a sin(3x) cos(2y) + noise over [-1, 1]^2, one amplitude per cloud"""
    import numpy as np
    rng = np.random.default_rng(seed)
    xy = rng.uniform(-1, 1, (B, n, 2))
    amp = rng.uniform(0.1, 0.4, (B, 1))
    z = amp * np.sin(3 * xy[..., 0]) * np.cos(2 * xy[..., 1]) + 0.01 * rng.standard_normal((B, n))
    return np.concatenate([xy, z[..., None]], -1).astype(np.float32)


def main():
    sys.dont_write_bytecode = True
    from make_golden import import_reference
    import_reference()
    import numpy as np
    import torch
    from golden_util import GOLDEN_DIR
    import utils.general_utils as r_gu

    torch.set_num_threads(8)
    a, b = surface_samples(711), surface_samples(712)
    at, bt = torch.from_numpy(a).requires_grad_(True), torch.from_numpy(b).requires_grad_(True)
    d = r_gu.pairwise_dist2(at, bt)
    loss = d.min(2).values.mean(1).mean() + d.min(1).values.mean(1).mean()
    loss.backward()
    np.savez_compressed(os.path.join(GOLDEN_DIR, "mesh_chamfer_s711.npz"), seed_pred=711, seed_targ=712,
                        loss=np.float64(loss.item()), grad_pred=at.grad.numpy())
    print("mesh_chamfer_s711: loss", loss.item())


if __name__ == "__main__":
    main()
