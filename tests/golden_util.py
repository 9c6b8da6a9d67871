"""Deterministic weights/inputs shared by oracle/make_golden.py (which feeds them to the real
reference) and by the tests (which feed them to the oracle restatement and to the HIP path).
numpy's PCG64 stream is version-stable, so only seeds and outputs need to be committed."""
import os
import zlib

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def fill_state_dict(module, seed):
    """Overwrite every parameter/buffer of `module` with values seeded by (seed, key name):
    conv/linear weights ~ N(0, 1/sqrt(fan_in)), BN weight in [0.5,1.5] with some negative entries,
    biases / running_mean ~ 0.1 N(0,1), running_var in [0.5, 1.5]."""
    sd = module.state_dict()
    out = {}
    for name, t in sd.items():
        # one stream per key, so the values do not depend on registration order
        rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
        shape = tuple(t.shape)
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros_like(t)
            continue
        r = rng.standard_normal(shape).astype(np.float32)
        if name.endswith("running_var"):
            v = 0.5 + rng.random(shape).astype(np.float32)
        elif name.endswith("running_mean") or name.endswith("bias"):
            v = 0.1 * r
        elif t.dim() == 1:  # norm-layer scale: mostly positive, every 5th negative
            v = 0.5 + rng.random(shape).astype(np.float32)
            v[::5] *= -1.0
        else:
            fan_in = int(np.prod(shape[1:]))
            v = r / np.sqrt(max(fan_in, 1))
        out[name] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).reshape(shape)
    module.load_state_dict(out)
    return module


def cloud(seed, B, C, N):
    """coords U(-1,1) in channels 0:3, extra channels N(0,1) -- SURVEY 8(d) synthetic inputs."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, C, N)).astype(np.float32)
    nc = min(C, 3)
    x[:, :nc] = rng.uniform(-1, 1, (B, nc, N)).astype(np.float32)
    return x


def seg_loss_case(seed, B, C, N, weighted, drop_class=False):
    """logits (B,C,N) ~ 2*N(0,1), labels (B,N) int64, class weights (C,) or None; `drop_class` leaves the last class
    without any point (its volume term is then 1/1e-6, losses/dice_loss.py:71)."""
    rng = np.random.default_rng(seed)
    logits = (2.0 * rng.standard_normal((B, C, N))).astype(np.float32)
    labels = rng.integers(0, C - 1 if drop_class else C, (B, N)).astype(np.int64)
    w = (0.5 + rng.random(C)).astype(np.float32) if weighted else None
    return logits, labels, w


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
