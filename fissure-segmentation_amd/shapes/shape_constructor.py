"""Folding grids of the PC-AE decoder (reference: shapes/shape_constructor.py)."""
import math
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))


def get_plane_mesh(n=2025, xrange=(-1, 1), yrange=(-1, 1), device='cpu'):
    """sqrt(n) x sqrt(n) vertex grid + two triangles per cell (shape_constructor.py:8-23)."""
    s = int(math.sqrt(n))
    gx, gy = torch.meshgrid(torch.linspace(xrange[0], xrange[1], s, device=device),
                            torch.linspace(yrange[0], yrange[1], s, device=device), indexing='ij')
    points = torch.stack([gx.reshape(-1), gy.reshape(-1)], dim=1)
    cell = (torch.arange(s - 1, device=device)[:, None] * s + torch.arange(s - 1, device=device)[None, :]).reshape(-1)
    faces = torch.stack([torch.stack([cell, cell + 1, cell + s], 1),
                         torch.stack([cell + 1, cell + s, cell + 1 + s], 1)], 1).reshape(-1, 3)
    return points, faces


def get_plane():
    """Fixed 45 x 45 grid on [-0.3, 0.3]^2 (shape_constructor.py:35-40)."""
    a = np.linspace(-0.3, 0.3, 45)
    return np.stack(np.meshgrid(a, a, indexing='ij'), -1).reshape(-1, 2)


def _load(name):
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        path = os.path.join("shapes", name)  # the reference opens it relative to the cwd (:27,:31)
    return np.load(path)


def get_sphere():
    return _load("sphere.npy")


def get_gaussian():
    return _load("gaussian.npy")
