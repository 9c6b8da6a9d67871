"""ctypes bindings of oracle/fsg_oracle.c (numpy in, numpy out).  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfsg_oracle.so")

KNN_FIX_DIAG = 1
KNN_DROP_FIRST = 2


def build(force=False):
    src = os.path.join(_HERE, "fsg_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libfsg_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def knn_dense(x, k, c_knn=None, fix_diag=True, drop_first=False):
    """x: (B, C, N) float32 -> idx (B, N, k) int32, dist (B, N, k) float32."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, C, N = x.shape
    c_knn = C if c_knn is None else c_knn
    flags = (KNN_FIX_DIAG if fix_diag else 0) | (KNN_DROP_FIRST if drop_first else 0)
    idx = np.empty((B, N, k), np.int32)
    dist = np.empty((B, N, k), np.float32)
    rc = lib().orc_knn_dense_f32(_p(x), B, N, ctypes.c_long(C * N), ctypes.c_long(N), c_knn, k,
                                 flags, _p(idx), _p(dist))
    if rc != 0:
        raise ValueError("orc_knn_dense_f32: bad arguments")
    return idx, dist


def edge_features(x, idx):
    x = np.ascontiguousarray(x, dtype=np.float32)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    B, C, N = x.shape
    k = idx.shape[2]
    out = np.empty((B, 2 * C, N, k), np.float32)
    lib().orc_edge_features_f32(_p(x), _p(idx), B, C, N, k, _p(out))
    return out


def edge_features_bwd(grad_edge, idx):
    g = np.ascontiguousarray(grad_edge, dtype=np.float32)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    B, C2, N, k = g.shape
    out = np.empty((B, C2 // 2, N), np.float32)
    lib().orc_edge_features_bwd_f32(_p(g), _p(idx), B, C2 // 2, N, k, _p(out))
    return out


def chamfer_nn(x, y):
    """x: (B, N, 3), y: (B, M, 3) -> (dist (B,N) f32, argmin (B,N) i32)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    B, N, _ = x.shape
    M = y.shape[1]
    d = np.empty((B, N), np.float32)
    a = np.empty((B, N), np.int32)
    lib().orc_chamfer_nn_f32(_p(x), _p(y), B, N, M, _p(d), _p(a))
    return d, a


def knn_segment(xyz, new_xyz, offset, new_offset, nsample):
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    new_xyz = np.ascontiguousarray(new_xyz, dtype=np.float32)
    offset = np.ascontiguousarray(offset, dtype=np.int32)
    new_offset = np.ascontiguousarray(new_offset, dtype=np.int32)
    m = new_xyz.shape[0]
    idx = np.empty((m, nsample), np.int32)
    d2 = np.empty((m, nsample), np.float32)
    lib().orc_knn_segment_f32(_p(xyz), _p(new_xyz), _p(offset), _p(new_offset), len(offset), nsample,
                              _p(idx), _p(d2))
    return idx, d2


def fps(xyz, offset, new_offset):
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    offset = np.ascontiguousarray(offset, dtype=np.int32)
    new_offset = np.ascontiguousarray(new_offset, dtype=np.int32)
    idx = np.zeros((int(new_offset[-1]),), np.int32)
    lib().orc_fps_f32(_p(xyz), _p(offset), _p(new_offset), len(offset), _p(idx))
    return idx
