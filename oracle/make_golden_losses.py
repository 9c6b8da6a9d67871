"""Golden vectors for the segmentation loss (losses/nnu_loss.py:6-19 = CrossEntropyLoss(class_weights) + GDL(softmax,
batch_dice=True), losses/dice_loss.py:24-96), produced by running the REAL reference on CPU.  Build container only.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_losses.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import REPO, import_reference  # noqa: E402


def main():
    sys.dont_write_bytecode = True
    import_reference()
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import numpy as np
    import torch
    from golden_util import GOLDEN_DIR, seg_loss_case

    import losses.nnu_loss as r_nnu

    out = {}
    # (seed, B, classes, N, weighted)
    cases = [(301, 2, 4, 256, True), (302, 2, 4, 256, False), (303, 3, 2, 100, True), (304, 1, 6, 333, True),
             (305, 2, 4, 64, True)]
    out["cases"] = np.array([c[:4] + (int(c[4]),) for c in cases], dtype=np.int64)
    for seed, B, C, N, weighted in cases:
        logits, labels, w = seg_loss_case(seed, B, C, N, weighted, drop_class=(seed == 305))
        x = torch.from_numpy(logits).requires_grad_(True)
        loss_fn = r_nnu.NNULoss(None if w is None else torch.from_numpy(w))
        total, parts = loss_fn(x, torch.from_numpy(labels))
        total.backward()
        out[f"s{seed}_total"] = total.detach().numpy()
        out[f"s{seed}_ce"] = parts["CE"].detach().numpy()
        out[f"s{seed}_gdl"] = parts["GDL"].detach().numpy()
        out[f"s{seed}_grad"] = x.grad.numpy()
    np.savez_compressed(os.path.join(GOLDEN_DIR, "nnu_loss.npz"), **out)
    print("wrote nnu_loss", len(out), "arrays")


if __name__ == "__main__":
    main()
