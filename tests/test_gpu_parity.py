"""GPU parity tests (pytest -m gpu): the HIP path, called through the C ABI of libfsg_hip.so, against
the CPU oracle and the committed golden vectors of the real reference.
Bars: integer/index outputs bit-exact (kNN also on distance bits), floating point within 1e-4."""
import glob
import json
import os
import sys

import numpy as np
import pytest
import torch

from golden_util import GOLDEN_DIR, cloud, fill_state_dict, load, seg_loss_case
from knn_check import assert_knn_equal
from oracle import c_api, ref_cpu

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-4, atol=1e-4)


@pytest.fixture(scope="module")
def fsg():
    import fissure_segmentation_amd as pkg
    return pkg


def G(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def N_(t):
    return t.detach().cpu().numpy()


def N(t):
    return t.detach().cpu().numpy()


# --------------------------------------------------------------------------- dense kNN
@pytest.mark.parametrize("B,C,Np,k,c_knn,fix,drop", [
    (2, 3, 256, 20, None, True, False), (2, 3, 256, 20, None, True, True), (2, 3, 256, 20, None, False, False),
    (3, 64, 300, 40, None, True, False), (2, 15, 513, 16, 3, True, False), (1, 3, 2048, 20, None, True, True),
    (1, 64, 2048, 20, None, True, False), (2, 3, 77, 63, None, True, True), (1, 7, 64, 64, None, True, False),
    (1, 3, 5000, 40, None, True, False), (1, 5, 8192, 40, 3, True, True), (1, 3, 1, 1, None, True, False)])
def test_knn_dense_bit_exact_vs_c_oracle(fsg, device, B, C, Np, k, c_knn, fix, drop):
    x = cloud(1000 + Np + C, B, C, Np)
    idx, dist = fsg.functional.knn_graph(G(x, device), k, c_knn=c_knn, fix_diag=fix, drop_first=drop, return_dist=True)
    ridx, rdist = c_api.knn_dense(x, k, c_knn=c_knn, fix_diag=fix, drop_first=drop)
    assert np.array_equal(N(idx), ridx)
    assert np.array_equal(N(dist).view(np.uint32), rdist.view(np.uint32))  # same bits


@pytest.mark.parametrize("B,C,Np,k,drop", [(8, 64, 2048, 20, False), (8, 3, 2048, 20, True), (4, 64, 8192, 40, False),
                                             (4, 3, 8192, 40, True)])
def test_knn_dense_baseline_batches_bit_exact_vs_c_oracle(fsg, device, B, C, Np, k, drop):
    """The graph builds of BASELINE configs 2 and 4 at their FULL per-GPU batch (8 x 2048, k = 20; 4 x 8192, k = 40; feature
    space and coordinates) straight against the C oracle -- indices and distance bits -- now that its brute-force search runs on
    all host cores (until round 4 the full batches were only compared with the independent rows kernel)."""
    x = cloud(7000 + Np + C, B, C, Np)
    idx, dist = fsg.functional.knn_graph(G(x, device), k, fix_diag=True, drop_first=drop, return_dist=True)
    ridx, rdist = c_api.knn_dense(x, k, c_knn=None, fix_diag=True, drop_first=drop)
    assert np.array_equal(N(idx), ridx)
    assert np.array_equal(N(dist).view(np.uint32), rdist.view(np.uint32))


@pytest.mark.parametrize("B,C,Np,k,c_knn,fix,drop", [
    (8, 64, 2048, 20, None, True, False), (8, 3, 2048, 20, None, True, False), (8, 3, 2048, 20, None, True, True),
    (4, 64, 8192, 40, None, True, False), (4, 3, 8192, 40, None, True, True), (2, 128, 2048, 20, None, False, False),
    (2, 15, 1000, 47, 3, True, True), (1, 100, 4097, 33, None, False, False), (3, 3, 1025, 20, None, True, False)])
def test_knn_mfma_kernel_equals_rows_kernel_at_full_size(fsg, device, B, C, Np, k, c_knn, fix, drop):
    """BASELINE config 2 / 4 sizes: the matrix-core kernel against the independent "rows in LDS" kernel (which is
    bit-exact with the C oracle on every size the oracle can reach): same indices, same distance bits."""
    x = G(cloud(5000 + Np + C, B, C, Np), device)
    r = fsg.functional.knn_graph(x, k, c_knn=c_knn, fix_diag=fix, drop_first=drop, return_dist=True,
                                 force_rows_kernel=True)
    # default entry (coarse-sweep + exact-refine kernel inside its envelope), two-phase kernel (2097152), its 512-candidate-chunk
    # variant (2048), and the one superseded design kept as an independent cross-check: the first MFMA kernel (8,
    # libfsg_hip_experiments.so)
    for dbg in (0, 2097152, 2048, 8):
        a = fsg.functional.knn_graph(x, k, c_knn=c_knn, fix_diag=fix, drop_first=drop, return_dist=True, _debug_flags=dbg)
        assert torch.equal(a[0], r[0]), dbg
        assert torch.equal(a[1].view(torch.int32), r[1].view(torch.int32)), dbg


def _split_inputs(kind, seed, B, C, Np):
    g = np.random.default_rng(seed)
    if kind == "uniform":
        return g.uniform(-1, 1, (B, C, Np)).astype(np.float32)
    if kind == "biased":       # what an EdgeConv emits: large common mean, small spread (cancellation in d = xx - 2 dot + xx)
        return (1.9 + 0.5 * g.standard_normal((B, C, Np))).astype(np.float32)
    if kind == "lowdim":       # features that are a smooth function of 3-D positions: neighbours far closer than the cloud radius
        p = g.uniform(0, 1, (B, 3, Np))
        w = g.standard_normal((C, 3))
        return (np.tanh(np.einsum("cd,bdn->bcn", w, p)) + 1.0).astype(np.float32)
    if kind == "lattice":      # every point twice on a coarse lattice: massive ties, lists overflow -> slow path
        p = g.integers(0, 6, (B, C, Np // 2)).astype(np.float32)
        return np.concatenate([p, p], 2)
    if kind == "far":          # a tight cloud far from the origin: |x|^2 ~ 1e4 times the neighbour distances
        return (100.0 + 0.01 * g.standard_normal((B, C, Np))).astype(np.float32)
    if kind == "outlier":      # one point far outside the range the fp16 image was scaled for: marked, the cloud goes the slow path
        x = g.uniform(-1, 1, (B, C, Np)).astype(np.float32)
        x[0, :, Np // 3] = 3.0e5
        return x
    if kind == "outlier0":     # the outlier sits INSIDE the centre / scale sample: everything else collapses below fp16's range
        x = g.uniform(-1, 1, (B, C, Np)).astype(np.float32)
        x[:, :, 3] = -7.0e6
        return x
    if kind == "tiny":         # coordinates far below fp16's range before the power-of-two scaling
        return (1e-9 * g.uniform(-1, 1, (B, C, Np))).astype(np.float32)
    if kind == "sorted":       # spatially sorted cloud: the sampled centre sits in one corner
        x = g.uniform(-1, 1, (B, C, Np)).astype(np.float32)
        return np.take_along_axis(x, np.argsort(x[:, :1], axis=2).repeat(C, 1), axis=2)
    raise ValueError(kind)


@pytest.mark.parametrize("B,C,Np,k,c_knn,fix,drop,kind,flags", [
    (2, 64, 2048, 20, None, True, False, "uniform", 0), (2, 64, 2048, 20, None, True, False, "biased", 0),
    (2, 64, 2048, 20, None, True, False, "lowdim", 0), (2, 3, 2048, 20, None, True, True, "uniform", 0),
    (2, 6, 2048, 20, 3, True, False, "uniform", 0), (2, 10, 1024, 16, None, False, False, "uniform", 0),
    (2, 24, 1500, 20, None, True, True, "lowdim", 0), (1, 33, 4096, 63, None, True, True, "biased", 0),
    (1, 64, 1024, 64, None, True, False, "lowdim", 0), (2, 3, 1024, 20, None, True, False, "lattice", 0),
    (1, 16, 1100, 8, None, True, False, "far", 0), (1, 64, 2048, 20, None, True, False, "uniform", 4194304),
    (1, 3, 1030, 40, None, True, True, "uniform", 4194304), (2, 8, 1100, 60, None, True, True, "uniform", 0), (2, 16, 1024, 20, None, True, False, "outlier", 0),
    (2, 16, 1024, 20, None, True, False, "tiny", 0), (1, 24, 1024, 20, None, True, False, "outlier0", 0), (2, 40, 2048, 20, None, True, True, "sorted", 0),
    (2, 64, 2048, 20, None, True, False, "biased", 1073741824), (2, 24, 1500, 20, None, True, True, "lowdim", 1073741824),
    (1, 64, 1024, 64, None, True, False, "lowdim", 1073741824), (2, 128, 2048, 20, None, True, False, "lowdim", 0),
    (1, 100, 1500, 33, None, True, True, "biased", 0), (1, 128, 1024, 20, None, True, False, "uniform", 4194304)])
def test_knn_split_kernel_bit_exact_vs_c_oracle(fsg, device, B, C, Np, k, c_knn, fix, drop, kind, flags):
    """fsg_knn_dense_ws_f32's coarse-sweep + exact-refine kernel (csrc/knn_split.hip: fp16 MFMA products on the centred, scaled
    points -- or three bf16 products on the points as they are: flag 1073741824, and always up to 4 channels -- only NOMINATE
    candidates under a rigorous error bound; every ranked distance is the oracle's fp32 fma chain): indices and distance bits
    equal the C oracle's on inputs that stress the bound (common mean, manifold features, far-from-origin clouds, ties, an
    outlier beyond the fp16 range, values below it, a spatially sorted cloud); flag 4194304 sends every query through the
    kernel's slow exact path."""
    x = _split_inputs(kind, 77 + Np + C, B, C, Np)
    idx, dist = fsg.functional.knn_graph(G(x, device), k, c_knn=c_knn, fix_diag=fix, drop_first=drop, return_dist=True,
                                         _debug_flags=flags)
    ridx, rdist = c_api.knn_dense(x, k, c_knn=c_knn, fix_diag=fix, drop_first=drop)
    assert np.array_equal(N(idx), ridx)
    assert np.array_equal(N(dist).view(np.uint32), rdist.view(np.uint32))  # same bits


@pytest.mark.parametrize("seed", range(12))
def test_knn_split_kernel_random_affine_clouds_vs_c_oracle(fsg, device, seed):
    """The error bound of the coarse sweeps is a claim about ARBITRARY inputs: random per-channel scales over six decades, random
    offsets up to 30 spreads away from the origin, constant channels, duplicated points, clustered clouds and anisotropic
    features, random channel counts / k / flags -- indices and distance bits must equal the C oracle's every time."""
    g = np.random.default_rng(9000 + seed)
    B, Np = 2, int(g.choice([1024, 1280, 1536]))
    C = int(g.choice([3, 5, 16, 24, 40, 64, 96]))
    k = int(g.choice([8, 20, 40]))
    x = g.standard_normal((B, C, Np))
    if seed % 3 == 0:      # clusters: most neighbours are much closer than the cloud is wide
        centres = 5.0 * g.standard_normal((B, C, 8))
        x = centres[:, :, g.integers(0, 8, Np)] + 0.05 * x
    scale = 10.0 ** g.uniform(-3, 3, (1, C, 1)) if seed % 2 else 10.0 ** g.uniform(-3, 3)
    shift = g.uniform(-30, 30, (1, C, 1)) * (seed % 4 != 0)
    x = (x + shift) * scale
    if C > 4:
        x[:, g.integers(0, C)] = 1.5                      # a constant channel
    dup = g.integers(0, Np, 40)
    x[:, :, dup[:20]] = x[:, :, dup[20:]]                  # duplicated points: exact ties
    x = x.astype(np.float32)
    fix, drop = bool(seed % 2), bool((seed // 2) % 2)
    idx, dist = fsg.functional.knn_graph(G(x, device), k, fix_diag=fix, drop_first=drop, return_dist=True)
    ridx, rdist = c_api.knn_dense(x, k, fix_diag=fix, drop_first=drop)
    assert np.array_equal(N(idx), ridx)
    assert np.array_equal(N(dist).view(np.uint32), rdist.view(np.uint32))


@pytest.mark.parametrize("B,C,Np,k,kind", [(8, 64, 2048, 20, "biased"), (8, 64, 2048, 20, "lowdim"), (4, 64, 8192, 40, "lowdim"),
                                           (4, 3, 8192, 40, "uniform"), (32, 3, 2048, 40, "uniform"), (2, 3, 2048, 20, "far"),
                                           (3, 48, 5000, 33, "biased"), (8, 128, 4096, 20, "lowdim")])
def test_knn_split_kernel_equals_two_phase_kernel_at_full_size(fsg, device, B, C, Np, k, kind):
    """BASELINE config 2 / 4 / 5 graph sizes: the default entry against the two-phase matrix-core kernel (flag 2097152), which
    reproduces the C oracle's bits on every size the oracle reaches."""
    x = G(_split_inputs(kind, 5 + Np, B, C, Np), device)
    a = fsg.functional.knn_graph(x, k, return_dist=True)
    r = fsg.functional.knn_graph(x, k, return_dist=True, _debug_flags=2097152)
    assert torch.equal(a[0], r[0])
    assert torch.equal(a[1].view(torch.int32), r[1].view(torch.int32))


@pytest.mark.parametrize("B,C,Np,k,rows", [(8, 3, 2048, 20, 128), (2, 3, 1024, 16, 128), (2, 4, 2048, 20, 64), (1, 2, 1500, 8, 256),
                                          (2, 3, 4096, 20, 128)])
def test_knn_graph_emits_first_edgeconv_rows(fsg, device, B, C, Np, k, rows):
    """fsg_knn_dense_ws_pq_f32: the graph build over the coordinates also hands out the per-point rows of the first EdgeConv's
    decomposed conv, pq = x^T W^T (models/dgcnn.py:212-243) -- fused into the build's first launch at N = 2048 (the no-prep
    path), a small launch of its own elsewhere.  The graph must be the plain build's, bit for bit; the rows are a K <= 4 fma
    chain: 1e-6 of their scale against float64."""
    F_hip = fsg.functional
    g = np.random.default_rng(B * 100 + C + Np)
    x = g.uniform(-1, 1, (B, C, Np)).astype(np.float32)
    w = g.standard_normal((rows, C)).astype(np.float32)
    xt, wt = G(x, device), G(w, device)
    idx, pq = F_hip.knn_graph(xt, k, fix_diag=True, pq_weight=wt)
    assert pq is not None and tuple(pq.shape) == (B, Np, rows)
    ref_idx = F_hip.knn_graph(xt, k, fix_diag=True)
    assert torch.equal(idx, ref_idx)
    io, _ = c_api.knn_dense(x, k, fix_diag=True)
    assert np.array_equal(N(idx), io)
    want = np.einsum("bcn,rc->bnr", x.astype(np.float64), w.astype(np.float64))
    assert np.abs(N(pq).astype(np.float64) - want).max() <= 1e-6 * max(1.0, np.abs(want).max())
    # a cloud that is NOT just its coordinates (features behind them): no by-product, the caller runs the product itself
    x6 = G(np.concatenate([x[:, :min(C, 3)], x[:, :min(C, 3)]], 1), device)
    idx6, none = F_hip.knn_graph(x6, k, c_knn=min(C, 3), fix_diag=True, pq_weight=G(g.standard_normal((rows, x6.shape[1])).astype(np.float32), device))
    assert none is None and idx6.shape == (B, Np, k)


def test_knn_massive_ties_take_the_slow_exact_path(fsg, device):
    """2000 identical points + a few distinct ones: every distance ties, far more than 128 survivors per row."""
    x = np.zeros((2, 3, 2100), np.float32)
    x[:, :, 2000:] = np.random.default_rng(0).uniform(-1, 1, (2, 3, 100)).astype(np.float32)
    for k, drop in ((20, False), (40, True), (63, True)):
        ridx, rdist = c_api.knn_dense(x, k, drop_first=drop)
        for dbg in (0, 2097152):   # coarse-sweep + exact-refine kernel; two-phase kernel
            idx, dist = fsg.functional.knn_graph(G(x, device), k, drop_first=drop, return_dist=True, _debug_flags=dbg)
            assert np.array_equal(N(idx), ridx) and np.array_equal(N(dist).view(np.uint32), rdist.view(np.uint32)), dbg


def test_knn_dense_ties_and_duplicates(fsg, device):
    """all-equal points (every distance ties) and duplicated points: lowest index first, like the oracle."""
    x = np.zeros((1, 3, 130), np.float32)
    x[0, :, 64:] = 1.0
    x[0, :, 100] = x[0, :, 3]
    for drop in (False, True):
        idx = fsg.functional.knn_graph(G(x, device), 9, drop_first=drop)
        assert np.array_equal(N(idx), c_api.knn_dense(x, 9, drop_first=drop)[0])


def test_knn_dense_strided_slice_no_copy(fsg, device):
    x = cloud(7, 2, 9, 200)
    xt = G(x, device)
    a = fsg.functional.knn_graph(xt[:, :3], 12)           # strided view
    b = fsg.functional.knn_graph(xt, 12, c_knn=3)
    assert torch.equal(a, b) and np.array_equal(N(a), c_api.knn_dense(x, 12, c_knn=3)[0])


def test_knn_dense_errors(fsg, device):
    x = torch.zeros(1, 3, 16, device=device)
    with pytest.raises(RuntimeError):
        fsg.functional.knn_graph(x, 17)
    with pytest.raises(RuntimeError):
        fsg.functional.knn_graph(x, 16, drop_first=True)
    with pytest.raises(RuntimeError):
        fsg.functional.knn_graph(torch.zeros(1, 3, 16), 4)  # CPU tensor: no fallback


KNN_FIXTURES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "knn_s*.npz")))


@pytest.mark.parametrize("name", KNN_FIXTURES)
def test_knn_reference_api_vs_golden(fsg, device, name):
    from fissure_segmentation_amd.models import dgcnn_opensrc
    from fissure_segmentation_amd.utils.general_utils import knn
    g = load(name)
    B, C, Np, k = int(g["B"]), int(g["C"]), int(g["N"]), int(g["k"])
    x = cloud(int(g["seed"]), B, C, Np)
    xt = G(x, device)
    for sl in (1, 0):
        idx, dist = knn(xt, k, self_loop=bool(sl), return_dist=True)
        assert idx.dtype == torch.int64 and idx.shape == (B, Np, k)
        assert_knn_equal(x, N(idx), g[f"idx_self{sl}"], k, drop_first=not sl)
        np.testing.assert_allclose(np.sort(N(dist), -1), np.sort(g[f"dist_self{sl}"], -1), rtol=1e-4, atol=2e-4)
    assert_knn_equal(x, N(dgcnn_opensrc.knn(xt, k)), g["idx_open"], k, fix_diag=False)
    if "idx_coords_self1" in g:
        assert_knn_equal(x, N(knn(xt[:, :3], k, self_loop=True)), g["idx_coords_self1"], k, c_knn=3)


def test_knn_full_size_properties(fsg, device):
    """BASELINE config 2 and 4 sizes: size-independent properties instead of the (slow) oracle."""
    for (B, C, Np, k) in [(8, 64, 2048, 20), (2, 3, 8192, 40)]:
        x = G(cloud(31, B, C, Np), device)
        idx, dist = fsg.functional.knn_graph(x, k, return_dist=True)
        assert int(idx.min()) >= 0 and int(idx.max()) < Np
        assert bool((dist[..., 1:] >= dist[..., :-1]).all())                       # ascending
        assert bool((idx[..., 0] == torch.arange(Np, device=device)).all())        # self first (d forced to 0)
        srt = torch.sort(idx, -1)[0]
        assert bool((srt[..., 1:] != srt[..., :-1]).all())                         # no repeats
        # distances agree with a direct recomputation for the returned pairs
        xi = x.transpose(1, 2)
        nb = torch.gather(xi.unsqueeze(1).expand(-1, Np, -1, -1), 2, idx.long().unsqueeze(-1).expand(-1, -1, -1, C)[:, :, :, :]) \
            if Np <= 2048 else None
        if nb is not None:
            d = (nb - xi.unsqueeze(2)).pow(2).sum(-1)
            torch.testing.assert_close(d, dist, rtol=1e-3, atol=2e-3)
        # dropping self == shifting by one when all distances are distinct
        idx2 = fsg.functional.knn_graph(x, k - 1, drop_first=True)
        assert torch.equal(idx2, idx[..., 1:])


# --------------------------------------------------------------------------- edge features
def test_edge_features_exact_and_grad(fsg, device):
    g = load("edge_feat_s201")
    x = cloud(201, 2, 5, 64)
    xt = G(x, device).requires_grad_(True)
    idx = G(g["idx"].astype(np.int32), device)
    from fissure_segmentation_amd.models.dgcnn import create_neighbor_features
    from fissure_segmentation_amd.models.dgcnn_opensrc import get_graph_feature
    e = create_neighbor_features(xt, 4, fixed_knn_graph=idx)
    assert np.array_equal(N(e), g["edge"])
    assert np.array_equal(N(get_graph_feature(xt.detach(), 4, idx)), g["edge_open"])
    gr = np.random.default_rng(int(g["gseed"])).standard_normal(e.shape).astype(np.float32)
    e.backward(G(gr, device))
    np.testing.assert_allclose(N(xt.grad), g["grad_x"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,C,Np,k", [(2, 64, 333, 20), (1, 3, 2048, 40), (3, 13, 50, 7)])
def test_edge_features_vs_c_oracle(fsg, device, B, C, Np, k):
    x = cloud(5, B, C, Np)
    idx, _ = c_api.knn_dense(x, k)
    xt = G(x, device).requires_grad_(True)
    e = fsg.functional.edge_features(xt, G(idx, device))
    assert np.array_equal(N(e), c_api.edge_features(x, idx))
    gr = np.random.default_rng(9).standard_normal(e.shape).astype(np.float32)
    e.backward(G(gr, device))
    np.testing.assert_allclose(N(xt.grad), c_api.edge_features_bwd(gr, idx), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B,Np,k", [(8, 2048, 20), (4, 8192, 40), (2, 9000, 8), (3, 77, 5), (1, 1, 1)])
def test_reverse_graph_csr(fsg, device, B, Np, k):
    """CSR by destination (fsg_graph_reverse_csr): both builders (16 workgroups per cloud with a workspace, one without)
    must hold exactly the in-edges (source << 6 | slot) of every destination, ascending inside a row -- with the workspace
    also the hub rows above the 1024-entry LDS sort capacity (ranked from a copy in the workspace); the one-workgroup
    builder without a workspace leaves those as filled: the graph, and the backward that walks it, is reproducible."""
    import ctypes
    rng = np.random.default_rng(B * Np + k)
    idx = rng.integers(0, Np, (B, Np, k)).astype(np.int32)
    idx[:, :, 0] = np.arange(Np)                       # self loops like a real kNN graph
    if Np > 10:
        idx[0, : Np // 2, 1] = 3                       # a hub: in-degree far above k
    flat = idx.reshape(B, -1)
    src = (np.arange(Np * k) // k) << 6 | (np.arange(Np * k) % k)
    it = G(idx, device)
    for use_ws in (True, False):
        rowptr = torch.empty(B, Np + 1, dtype=torch.int32, device=device)
        col = torch.full((B, Np * k), -1, dtype=torch.int32, device=device)
        ws = torch.empty(fsg._lib.lib.fsg_graph_reverse_csr_workspace_bytes(B, Np, k) // 4 + 1, dtype=torch.int32, device=device)
        fsg._lib.call("fsg_graph_reverse_csr", ctypes.c_void_p(it.data_ptr()), B, Np, k, ctypes.c_void_p(rowptr.data_ptr()),
                      ctypes.c_void_p(col.data_ptr()), ctypes.c_void_p(ws.data_ptr() if use_ws else 0),
                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        rp, cl = N(rowptr), N(col)
        for b in range(B):
            deg = np.bincount(flat[b], minlength=Np)
            assert np.array_equal(rp[b], np.concatenate([[0], np.cumsum(deg)])), use_ws
            order = np.argsort(flat[b], kind="stable")
            want = src[order]                          # in-edges grouped by destination, ascending inside a group
            got = cl[b].copy()
            if not use_ws:
                for j in np.nonzero(deg > 1024)[0]:    # no workspace: hubs above the LDS sort capacity keep any order
                    got[rp[b, j]:rp[b, j + 1]].sort()
            assert np.array_equal(got, want), use_ws


# --------------------------------------------------------------------------- Chamfer
@pytest.mark.parametrize("B,Np,M", [(2, 512, 384), (1, 1, 5), (3, 1500, 2048), (1, 4096, 4096)])
def test_chamfer_nn_exact(fsg, device, B, Np, M):
    rng = np.random.default_rng(Np + M)
    a = rng.uniform(-1, 1, (B, Np, 3)).astype(np.float32)
    b = rng.uniform(-1, 1, (B, M, 3)).astype(np.float32)
    b[:, M // 2] = b[:, 0]  # a tie: lowest index must win
    d, arg = fsg.functional.chamfer_nn(G(a, device), G(b, device))
    rd, rarg = c_api.chamfer_nn(a, b)
    assert np.array_equal(N(arg), rarg) and np.array_equal(N(d).view(np.uint32), rd.view(np.uint32))


def test_chamfer_loss_vs_golden(fsg, device):
    from fissure_segmentation_amd.losses.chamfer_loss import ChamferLoss
    g = load("chamfer_s701")
    rng = np.random.default_rng(701)
    a = rng.uniform(-1, 1, (2, 512, 3)).astype(np.float32)
    b = rng.uniform(-1, 1, (2, 384, 3)).astype(np.float32)
    at, bt = G(a, device).requires_grad_(True), G(b, device).requires_grad_(True)
    loss = ChamferLoss()(at, bt)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * float(g["loss"])
    np.testing.assert_allclose(N(at.grad), g["grad_a"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(N(bt.grad), g["grad_b"], rtol=1e-4, atol=1e-6)
    l2 = ChamferLoss()(G(a, device).transpose(1, 2), G(b, device).transpose(1, 2))  # (B,3,N) layout
    assert abs(l2.item() - float(g["loss"])) <= 1e-5 * float(g["loss"])
    with pytest.raises(AssertionError):
        ChamferLoss()(G(a, device), G(b[:1], device))


def test_mesh_loss_chamfer_term_vs_golden(fsg, device):
    """RegularizedMeshLoss (losses/mesh_loss.py:24-33 of the reference) at the size `train_pc_ae.py --loss mesh` trains with:
    2048 surface samples per mesh; value and gradient against the reference's pairwise_dist2 (oracle/make_golden_mesh.py),
    `sampler=` / `sample_points()` hooks, (loss, components) contract of model_trainer.py:180-185"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    from make_golden_mesh import surface_samples
    from fissure_segmentation_amd.losses.access_losses import get_loss_fn
    g = load("mesh_chamfer_s711")
    a, b = surface_samples(int(g["seed_pred"])), surface_samples(int(g["seed_targ"]))
    crit = get_loss_fn("mesh", term_weights=[1., 0., 0., 0.])
    at = G(a, device).requires_grad_(True)
    loss, parts = crit(at, G(b, device))
    loss.backward()
    assert set(parts) == {"Chamfer"} and parts["Chamfer"].item() == loss.item()
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * float(g["loss"])
    # the reference's expanded form |x|^2 - 2xy + |y|^2 and the direct (x - y)^2 form pick another of two near-equidistant
    # targets on a few rows (3 of 4096 here): rows must agree at 1e-4 except for <= 0.2 percent of them, whole tensor 5e-3 in norm
    err = np.abs(N(at.grad) - g["grad_pred"])
    rows_off = (err > 1e-4 * np.abs(g["grad_pred"]) + 1e-7).any(-1)
    assert rows_off.mean() <= 2e-3 and np.linalg.norm(err) <= 5e-3 * np.linalg.norm(g["grad_pred"])
    # bit-exact against the C oracle's direct-form distances (the kernel's own arithmetic contract)
    d1, _ = c_api.chamfer_nn(a, b)
    d2, _ = c_api.chamfer_nn(b, a)
    want = np.float32(d1.mean(1, dtype=np.float64).mean() + d2.mean(1, dtype=np.float64).mean())
    assert abs(loss.item() - float(want)) <= 1e-6 * float(want)
    # weight, (B,3,n) layout, sampler hook and sample_points() objects
    l2, _ = get_loss_fn("mesh", term_weights=[2.5, 0., 0., 0.])(G(a, device).transpose(1, 2), G(b, device).transpose(1, 2))
    assert abs(l2.item() - 2.5 * float(g["loss"])) <= 1e-5 * 2.5 * float(g["loss"])

    class Surf:
        def __init__(self, pts):
            self.pts = pts

        def sample_points(self, n):
            assert n == 2048
            return self.pts
    l3, _ = crit(Surf(G(a, device)), Surf(G(b, device)))
    assert l3.item() == loss.item()
    from fissure_segmentation_amd.losses.mesh_loss import RegularizedMeshLoss
    l4, _ = RegularizedMeshLoss(1., 0., 0., 0., n_samples=7, sampler=lambda m, n: m.pts)(Surf(G(a, device)), Surf(G(b, device)))
    assert l4.item() == loss.item()
    with pytest.raises(NotImplementedError, match="Laplacian"):
        RegularizedMeshLoss(1., 0., 0., 0.1)(G(a, device), G(b, device))


# --------------------------------------------------------------------------- segmentation loss
def _nnu_case(g, row):
    seed, B, C, Np, wt = (int(v) for v in row)
    lg, lb, w = seg_loss_case(seed, B, C, Np, bool(wt), drop_class=(seed == 305))
    return seed, lg, lb, w


@pytest.mark.parametrize("layout", ["class_major", "point_major"])
def test_nnu_loss_vs_golden(fsg, device, layout):
    """fused CE + generalised Dice (value and gradient) against the reference's own outputs (tests/golden/nnu_loss.npz,
    made by oracle/make_golden_losses.py from losses/nnu_loss.py) -- fp32, tolerance 1e-4 relative (north star)."""
    from fissure_segmentation_amd.losses.nnu_loss import NNULoss
    g = load("nnu_loss")
    for row in g["cases"]:
        seed, lg, lb, w = _nnu_case(g, row)
        x = G(lg, device)
        if layout == "point_major":
            x = x.transpose(1, 2).contiguous().transpose(1, 2)      # (B,C,N) view of (B,N,C) memory
        x.requires_grad_(True)
        crit = NNULoss(None if w is None else torch.from_numpy(w)).to(device)
        total, parts = crit(x, G(lb, device))
        total.backward()
        for got, key in [(total, "total"), (parts["CE"], "ce"), (parts["GDL"], "gdl")]:
            ref = float(g[f"s{seed}_{key}"])
            assert abs(got.item() - ref) <= 1e-4 * max(abs(ref), 1e-3), (seed, key, got.item(), ref)
        gr = g[f"s{seed}_grad"]
        assert x.grad.stride() == x.stride()
        assert np.abs(N(x.grad) - gr).max() <= 1e-4 * np.abs(gr).max(), seed


def test_nnu_loss_vs_oracle_full_size(fsg, device):
    """BASELINE config 2 shape (8 x 4 x 2048) and a 6-class, ragged one against the CPU restatement in fp64; the result
    must be reproducible bit for bit (no atomics) and scale with the incoming gradient."""
    from oracle import ref_cpu
    for seed, B, C, Np in [(311, 8, 4, 2048), (312, 3, 6, 1000), (313, 2, 17, 130), (314, 1, 32, 70)]:
        lg, lb, w = seg_loss_case(seed, B, C, Np, True)
        xr = torch.from_numpy(lg).double().requires_grad_(True)
        tr, cr, gr = ref_cpu.nnu_loss(xr, torch.from_numpy(lb), torch.from_numpy(w).double())
        (3.0 * tr).backward()
        x = G(lg, device).requires_grad_(True)
        t, c, gd = fsg.functional.nnu_loss(x, G(lb, device), G(w, device))
        (3.0 * t).backward()
        for got, ref in [(t, tr), (c, cr), (gd, gr)]:
            assert abs(got.item() - ref.item()) <= 1e-5 * max(abs(ref.item()), 1e-3)
        ref_g = xr.grad.numpy()
        assert np.abs(N(x.grad) - ref_g).max() <= 1e-4 * np.abs(ref_g).max()
        x2 = G(lg, device).requires_grad_(True)
        t2, _, _ = fsg.functional.nnu_loss(x2, G(lb, device), G(w, device))
        (3.0 * t2).backward()
        assert torch.equal(t, t2) and torch.equal(x.grad, x2.grad)
    with pytest.raises(ValueError):
        fsg.functional.nnu_loss(G(lg, device), G(lb[:, :5], device))


# --------------------------------------------------------------------------- packed-cloud primitives
def packed(seed, sizes, c=0):
    rng = np.random.default_rng(seed)
    n = int(sum(sizes))
    xyz = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    feat = rng.standard_normal((n, c)).astype(np.float32) if c else None
    return xyz, feat, np.cumsum(sizes).astype(np.int32)


@pytest.mark.parametrize("sizes,ns", [([300, 200, 257], 16), ([8, 8], 16), ([2048], 8), ([5, 1, 64], 3), ([100], 33),
                                      ([3000, 100, 1025], 16), ([1, 2, 3, 40], 32), ([2048] * 8, 8), ([512] * 8, 16)])
def test_knn_segment_exact(fsg, device, sizes, ns):
    xyz, _, off = packed(11, sizes)
    xyz[7 % len(xyz)] = xyz[0]                      # duplicate points: ties go to the lower index
    if len(xyz) > 1500:
        xyz[1100:1400] = np.round(xyz[1100:1400] * 4) / 4   # a lattice: many exactly equal distances
    idx, d2 = fsg.functional.knn_segment(ns, G(xyz, device), G(xyz, device), G(off, device), G(off, device))
    ridx, rd2 = c_api.knn_segment(xyz, xyz, off, off, ns)
    assert np.array_equal(N(idx), ridx) and np.array_equal(N(d2).view(np.uint32), rd2.view(np.uint32))


def test_knn_segment_cross_sets_and_fps(fsg, device):
    sizes = [512, 300, 129]
    xyz, _, off = packed(12, sizes)
    new_off = np.cumsum([s // 4 for s in sizes]).astype(np.int32)
    fi = fsg.functional.fps(G(xyz, device), G(off, device), G(new_off, device), int(new_off[-1]))
    rfi = c_api.fps(xyz, off, new_off)
    assert np.array_equal(N(fi), rfi)
    q = xyz[rfi]
    idx, d2 = fsg.functional.knn_segment(16, G(xyz, device), G(q, device), G(off, device), G(new_off, device))
    ridx, rd2 = c_api.knn_segment(xyz, q, off, new_off, 16)
    assert np.array_equal(N(idx), ridx) and np.array_equal(N(d2).view(np.uint32), rd2.view(np.uint32))
    # the interpolation query (pointops.py:198-215): fine points ask for their 3 nearest coarse points
    idx, d2 = fsg.functional.knn_segment(3, G(q, device), G(xyz, device), G(new_off, device), G(off, device))
    ridx, rd2 = c_api.knn_segment(q, xyz, new_off, off, 3)
    assert np.array_equal(N(idx), ridx) and np.array_equal(N(d2).view(np.uint32), rd2.view(np.uint32))


@pytest.mark.parametrize("I,J,K,sa,sb", [(256, 256, 256, "ik", "jk"), (64, 512, 512, "ik", "kj"), (96, 32, 16384, "ki", "kj"),
                                         (64, 35, 65536, "ki", "kj"), (1000, 13, 7, "ik", "jk"), (1, 4, 32, "ik", "jk"),
                                         (192, 64, 4096, "ki", "kj"), (77, 130, 259, "ik", "kj"), (256, 256, 768, "ik", "kj")])
def test_small_gemm(fsg, device, I, J, K, sa, sb):
    """fsg_gemm_small_f32 with every operand orientation (x W^T, dY W, dY^T X incl. reductions over all rows split across
    workgroups) against fp64; fp32 fma chains -> ~1e-6 relative per term; bit-reproducible."""
    rng = np.random.default_rng(I + J + K)
    a = rng.standard_normal((I, K) if sa == "ik" else (K, I)).astype(np.float32)
    b = rng.standard_normal((K, J) if sb == "kj" else (J, K)).astype(np.float32)
    bias = rng.standard_normal(J).astype(np.float32)
    at, bt = G(a, device), G(b, device)
    args = (at, K if sa == "ik" else 1, 1 if sa == "ik" else I, bt, J if sb == "kj" else 1, 1 if sb == "kj" else K)
    c = fsg.functional.gemm_small(*args, G(bias, device), I, J, K)
    ref = (a if sa == "ik" else a.T).astype(np.float64) @ (b if sb == "kj" else b.T).astype(np.float64) + bias
    assert np.abs(N(c) - ref).max() <= 3e-7 * np.sqrt(K) * max(1.0, np.abs(ref).max())
    assert torch.equal(c, fsg.functional.gemm_small(*args, G(bias, device), I, J, K))
    # the row-sum by-product (fsg_gemm_small_rowsum_f32: the bias gradient next to a weight gradient): same product, + sum_k A(i, k)
    c2, rs = fsg.functional.gemm_small(*args, G(bias, device), I, J, K, rowsum=True)
    assert torch.equal(c2, c)
    rref = (a if sa == "ik" else a.T).astype(np.float64).sum(1)
    assert np.abs(N(rs) - rref).max() <= 3e-7 * np.sqrt(K) * max(1.0, np.abs(rref).max()) + 1e-5
    assert torch.equal(rs, fsg.functional.gemm_small(*args, G(bias, device), I, J, K, rowsum=True)[1])


def test_linear_pm_routing_and_grads(fsg, device):
    """linear_pm sends exactly the products whose output fits one 256x256 vendor macro-tile to the small kernel"""
    for M, Nn, K, n_small in [(256, 256, 256, 3), (256, 768, 256, 1), (16384, 96, 32, 1), (1024, 384, 128, 0)]:
        rng = np.random.default_rng(M)
        x, w, b = rng.standard_normal((M, K)), rng.standard_normal((Nn, K)) / np.sqrt(K), rng.standard_normal(Nn)
        g = rng.standard_normal((M, Nn))
        xt, wt, bt = (G(v.astype(np.float32), device).requires_grad_(True) for v in (x, w, b))
        fsg._lib.start_timing()
        y = fsg.functional.linear_pm(xt, wt, bt)
        y.backward(G(g.astype(np.float32), device))
        timed = fsg._lib.stop_timing()
        assert sum(len(timed.get(f"fsg_gemm_small_{e}f32", [])) for e in ("", "rowsum_", "deferred_")) == n_small, (M, Nn, K)
        for got, ref in [(y, x @ w.T + b), (xt.grad, g @ w), (wt.grad, g.T @ x), (bt.grad, g.sum(0))]:
            assert np.abs(N(got) - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())
        # the weight gradient of a leaf parameter leaves its split sum to the end of the backward pass (one launch for all deferred
        # products: fsg_gemm_small_reduce_many_f32); same partial products, summed in split order (fp32 rounding apart)
        if "fsg_gemm_small_deferred_f32" in timed:
            assert len(timed.get("fsg_gemm_small_reduce_many_f32", [])) == 1
            fsg.functional.set_deferred_weight_grads(False)
            try:
                x2, w2, b2 = (v.detach().clone().requires_grad_(True) for v in (xt, wt, bt))
                fsg.functional.linear_pm(x2, w2, b2).backward(G(g.astype(np.float32), device))
            finally:
                fsg.functional.set_deferred_weight_grads(True)
            close = lambda a, b: float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())  # noqa: E731
            assert close(w2.grad, wt.grad) and close(b2.grad, bt.grad)
            # a weight that is NOT a leaf (its gradient is read by the producer's backward) is reduced at once
            x3, w3 = xt.detach().clone().requires_grad_(True), wt.detach().clone().requires_grad_(True)
            fsg._lib.start_timing()
            fsg.functional.linear_pm(x3, w3 * 1.0, None).backward(G(g.astype(np.float32), device))
            assert "fsg_gemm_small_deferred_f32" not in fsg._lib.stop_timing()
            assert close(w3.grad, wt.grad)
            # ... and so is one that ACCUMULATES into a gradient that is already there (AccumulateGrad adds on the spot)
            first = wt.grad.clone()
            fsg._lib.start_timing()
            fsg.functional.linear_pm(xt, wt, bt).backward(G(g.astype(np.float32), device))
            assert "fsg_gemm_small_deferred_f32" not in fsg._lib.stop_timing()
            assert close(wt.grad, 2 * first)


@pytest.mark.parametrize("M,C,relu,res,train", [(16384, 32, True, False, True), (4096, 64, True, True, True),
                                                (65536, 64, True, False, True), (257, 128, False, False, True),
                                                (256, 256, True, True, True), (64, 512, True, True, True),
                                                (1000, 64, True, True, False), (3, 32, True, False, True)])
def test_bn_rows_fused_vs_torch(fsg, device, M, C, relu, res, train):
    """fsg_bn_rows_{fwd,bwd}: [relu](BatchNorm1d(x) [+ residual]) against torch's BatchNorm1d in fp64 (output, all
    gradients, running statistics, num_batches_tracked)."""
    rng = np.random.default_rng(M + C)
    x = (rng.standard_normal((M, C)) * rng.uniform(0.05, 3, C) + rng.uniform(-5, 5, C)).astype(np.float32)
    r = rng.standard_normal((M, C)).astype(np.float32)
    g = rng.standard_normal((M, C)).astype(np.float32)
    ref = torch.nn.BatchNorm1d(C).double()
    ref.weight.data = torch.from_numpy(rng.uniform(-2, 2, C)); ref.bias.data = torch.from_numpy(rng.standard_normal(C))
    ref.running_mean.data = torch.from_numpy(rng.standard_normal(C)); ref.running_var.data = torch.from_numpy(rng.uniform(0.5, 2, C))
    from fissure_segmentation_amd.norm import BatchNorm1d
    bn = BatchNorm1d(C)
    bn.load_state_dict({k: (v.float() if v.is_floating_point() else v) for k, v in ref.state_dict().items()})
    bn = bn.to(device)
    ref.train(train), bn.train(train)
    xr, rr = torch.from_numpy(x).double().requires_grad_(True), torch.from_numpy(r).double().requires_grad_(True)
    yr = ref(xr) + (rr if res else 0)
    yr = torch.relu(yr) if relu else yr
    yr.backward(torch.from_numpy(g).double())
    xt, rt = G(x, device).requires_grad_(True), G(r, device).requires_grad_(True)
    y = fsg.functional.bn_rows(xt, bn, relu=relu, residual=rt if res else None)
    y.backward(G(g, device))
    tol = 2e-5 * max(1.0, float(yr.abs().max()))
    assert np.abs(N(y) - yr.detach().numpy()).max() <= tol
    for got, want in [(xt.grad, xr.grad), (bn.weight.grad, ref.weight.grad), (bn.bias.grad, ref.bias.grad)] + \
                     ([(rt.grad, rr.grad)] if res else []):
        want = want.numpy()
        assert np.abs(N(got) - want).max() <= 1e-4 * max(1e-3, np.abs(want).max()), (got.shape,)
    np.testing.assert_allclose(N(bn.running_mean), ref.running_mean.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(N(bn.running_var), ref.running_var.numpy(), rtol=1e-5, atol=1e-6)
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked)


@pytest.mark.parametrize("sizes", [[2048] * 8, [512, 100, 7, 513], [3000, 64, 2049], [1, 2, 65]])
def test_fps_exact(fsg, device, sizes):
    """farthest point sampling (pointops.py:16-39), a quarter of every segment: single-wave register path (<= 512 and
    <= 2048 points) and the workgroup path, bit-exact incl. ties (duplicated points -> lowest index)."""
    xyz, _, off = packed(21, sizes)
    xyz[len(xyz) // 2] = xyz[0]
    xyz[-1] = xyz[len(xyz) // 3]
    new_off = np.cumsum([max(1, s // 4) for s in sizes]).astype(np.int32)
    fi = fsg.functional.fps(G(xyz, device), G(off, device), G(new_off, device), int(new_off[-1]))
    assert np.array_equal(N(fi), c_api.fps(xyz, off, new_off))


def test_knn_segment_config3_sizes(fsg, device):
    """BASELINE config 3 shapes: 8 clouds x 2048 points; level-1 graph (nsample 8), the TransitionDown query of 512
    sampled points per cloud (nsample 16) and the level-2 graph -- bit-exact against the C oracle."""
    xyz, _, off = packed(13, [2048] * 8)
    new_off = np.cumsum([512] * 8).astype(np.int32)
    sel = np.concatenate([np.arange(512) * 4 + 2048 * b for b in range(8)])
    q = np.ascontiguousarray(xyz[sel])
    for ns, a, b_, oa, ob in [(8, xyz, xyz, off, off), (16, xyz, q, off, new_off), (16, q, q, new_off, new_off)]:
        idx, d2 = fsg.functional.knn_segment(ns, G(a, device), G(b_, device), G(oa, device), G(ob, device))
        ridx, rd2 = c_api.knn_segment(a, b_, oa, ob, ns)
        assert np.array_equal(N(idx), ridx) and np.array_equal(N(d2).view(np.uint32), rd2.view(np.uint32))


def test_group_gather_and_vec_attn(fsg, device):
    rng = np.random.default_rng(5)
    n, ns, c, cw = 200, 8, 32, 4
    v = torch.tensor(rng.standard_normal((n, c)), dtype=torch.float32, device=device, requires_grad=True)
    pos = torch.tensor(rng.standard_normal((n, ns, c)), dtype=torch.float32, device=device, requires_grad=True)
    w = torch.tensor(rng.random((n, ns, cw)), dtype=torch.float32, device=device, requires_grad=True)
    idx = torch.tensor(rng.integers(0, n, (n, ns)), dtype=torch.int32, device=device)
    gg = fsg.functional.group_gather(v, idx)
    assert torch.equal(gg, v[idx.long()])
    out = fsg.functional.vec_attn(v, pos, w, idx)
    ref = ((v[idx.long()] + pos).view(n, ns, c // cw, cw) * w.unsqueeze(2)).sum(1).view(n, c)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)
    go = torch.tensor(rng.standard_normal((n, c)), dtype=torch.float32, device=device)
    g1 = torch.autograd.grad(out, (v, pos, w), go)
    g2 = torch.autograd.grad(ref, (v, pos, w), go)
    for a, b in zip(g1, g2):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
    g3 = torch.autograd.grad(gg, v, torch.ones_like(gg))[0]
    torch.testing.assert_close(g3, torch.autograd.grad(v[idx.long()], v, torch.ones_like(gg))[0])


# --------------------------------------------------------------------------- models vs golden + oracle
def run_model(net, x, gseed, device):
    xt = G(x, device).requires_grad_(True)
    y = net(xt)
    gr = np.random.default_rng(gseed).standard_normal(tuple(y.shape)).astype(np.float32)
    y.backward(G(gr, device))
    return y, xt.grad


def check_against_golden(net, g, y, gx, out_key, loose=False, out_tol=None):
    """north_star bar: outputs within 1e-4.  Gradients are compared in norm against the scale of their tensor:
    (i) near-ties in max-pools (over k neighbours, over N points) route the gradient to a different element on
    different fp32 summation orders, which changes isolated entries by O(1) without changing the function;
    (ii) several parameters have a mathematically ZERO gradient (a bias in front of a train-mode BatchNorm or a
    softmax), whose computed value is rounding noise on both sides; (iii) `loose` marks the spatial-transformer
    fixture, whose MLP runs train-mode BatchNorm over a batch of TWO samples and amplifies 1e-7 input
    differences to 1e-3."""
    np.testing.assert_allclose(N(y), g[out_key], **(out_tol or TOL))
    ref_gx = g["grad_x"]
    assert np.linalg.norm(N(gx) - ref_gx) <= (3e-2 if loose else 5e-3) * np.linalg.norm(ref_gx)
    norms = {n: float(g["gnorm_" + n]) for n, _ in net.named_parameters()}
    floor = 1e-4 * max(norms.values())
    for n, p in net.named_parameters():
        got = p.grad.reshape(-1)
        assert abs(float(got.double().norm()) - norms[n]) <= (3e-2 if loose else 2e-3) * norms[n] + floor, n
        head = g["ghead_" + n]
        np.testing.assert_allclose(N(got[:16]), head, rtol=2e-3,
                                   atol=(5e-2 if loose else 5e-3) * float(np.abs(head).max()) + floor, err_msg=n)


@pytest.mark.parametrize("name", ["edgeconv_first", "edgeconv_feat", "edgeconv_c15"])
def test_edgeconv_vs_golden(fsg, device, name):
    from fissure_segmentation_amd.models.dgcnn import EdgeConv
    g = load(name)
    seed, cin, k, Np = int(g["seed"]), int(g["cin"]), int(g["k"]), int(g["N"])
    ec = fill_state_dict(EdgeConv(cin, [int(c) for c in g["couts"]], k, first_layer=bool(g["first"])), seed).to(device).train()
    y, gx = run_model(ec, cloud(seed + 1000, 2, cin, Np), seed + 2000, device)
    np.testing.assert_allclose(N(y), g["y"], **TOL)
    # gradients in norm: a near-tie in the max over k re-routes the gradient of single (point, channel) pairs
    assert np.linalg.norm(N(gx) - g["grad_x"]) <= 5e-3 * np.linalg.norm(g["grad_x"])
    for n, p in ec.named_parameters():
        ref = g["grad_" + n]
        assert np.linalg.norm(N(p.grad) - ref) <= 5e-3 * np.linalg.norm(ref) + 1e-4, n
    for n, b in ec.named_buffers():
        if "running" in n:
            np.testing.assert_allclose(N(b), g["buf_" + n], **TOL)


@pytest.mark.parametrize("name", ["dgcnnseg_dyn", "dgcnnseg_static", "dgcnnseg_c15_eval", "dgcnnseg_stn", "dgcnnseg_img"])
def test_dgcnnseg_vs_golden(fsg, device, name):
    from fissure_segmentation_amd.models.dgcnn import DGCNNSeg
    g = load(name)
    seed, cin = int(g["seed"]), int(g["cin"])
    net = DGCNNSeg(k=8, in_features=cin, num_classes=4, dynamic=bool(g["dynamic"]),
                   spatial_transformer=name.endswith("stn"), image_feat_module=name.endswith("img"))
    assert list(net.state_dict().keys()) == [str(s) for s in g["keys"]]
    fill_state_dict(net, seed).to(device).train(bool(g["train"]))
    y, gx = run_model(net, cloud(seed + 1000, 2, cin, 128), seed + 2000, device)
    check_against_golden(net, g, y, gx, "logits", loose=name.endswith("stn"))


@pytest.mark.parametrize("name", ["ae_fold", "ae_deform_static"])
def test_folding_ae_vs_golden(fsg, device, name):
    from fissure_segmentation_amd.models.folding_net import DGCNNFoldingNet
    g = load(name)
    seed = int(g["seed"])
    net = DGCNNFoldingNet(k=8, n_embedding=64, shape_type="plane", n_input_points=2048, decode_mesh=False,
                          deform=bool(g["deform"]), static=bool(g["static"]))
    assert sorted(net.state_dict().keys()) == sorted(str(s) for s in g["keys"])
    fill_state_dict(net, seed).to(device).train()
    y, gx = run_model(net, cloud(seed + 1000, 2, 3, 2048), seed + 2000, device)
    # the deforming decoder batch-normalises channels that are almost constant over the 2025 grid points (the code
    # vector is broadcast, only 3 of 67 inputs vary): fp32 BN is ill-conditioned there, 3e-4 instead of 1e-4
    # (and its gradient passes through 1/std ~ 1e2 per BatchNorm: `loose`)
    check_against_golden(net, g, y, gx, "recon", loose=bool(g["deform"]),
                         out_tol=dict(rtol=3e-4, atol=3e-4) if g["deform"] else None)


def test_pointnet_config1_on_gpu(fsg, device):
    """BASELINE config 1 is the reference's CPU case (tests/test_host_cpu.py holds the full check); on the GPU
    the logits must agree, the gradient bar is loose because arg-max ties of the 1024-point max-pool route
    gradient differently between ATen's CPU and GPU kernels."""
    from fissure_segmentation_amd.models.point_net import PointNetSeg
    g = load("pointnet_c1")
    net = fill_state_dict(PointNetSeg(3, 4), 501).to(device).train()
    y, gx = run_model(net, cloud(1501, 8, 3, 1024), 2501, device)
    np.testing.assert_allclose(N(y), g["logits"], rtol=2e-4, atol=2e-4)
    ref = g["grad_x"]
    assert np.linalg.norm(N(gx) - ref) <= 5e-2 * np.linalg.norm(ref)


def test_pointtransformer_vs_cpu_restatement(fsg, device):
    """parity UNPINNED at the pointops_cuda boundary: the oracle is the CPU restatement."""
    from fissure_segmentation_amd.models.pointtransformer.seg_model import PointTransformerCompatibility
    torch.manual_seed(0)
    B, C, Np = 2, 6, 2048   # level 5 has 8 points per cloud < nsample 16: the padding rule is exercised
    ref = fill_state_dict(ref_cpu.PointTransformerCompatibility(C, 4), 801).train()
    net = PointTransformerCompatibility(C, 4)
    net.load_state_dict(ref.state_dict())
    net = net.to(device).train()
    x = cloud(1801, B, C, Np)
    xr = torch.from_numpy(x).requires_grad_(True)
    yr = ref(xr)
    gr = np.random.default_rng(2801).standard_normal(tuple(yr.shape)).astype(np.float32)
    yr.backward(torch.from_numpy(gr))
    y, gx = run_model(net, x, 2801, device)
    np.testing.assert_allclose(N(y), yr.detach().numpy(), rtol=5e-4, atol=5e-4)
    rgx = xr.grad.numpy()
    assert np.linalg.norm(N(gx) - rgx) <= 3e-2 * np.linalg.norm(rgx)
    refp = dict(ref.named_parameters())
    scale = max(float(q.grad.double().norm()) for q in refp.values())
    for n, p in net.named_parameters():   # zero-gradient parameters (bias before BN / softmax) are pure noise
        a, b = p.grad.double().cpu().reshape(-1), refp[n].grad.double().reshape(-1)
        assert float((a - b).norm()) <= 3e-2 * float(b.norm()) + 1e-3 * scale, n


@pytest.mark.parametrize("c,ns,sizes,train", [(32, 8, (100, 37), True), (64, 16, (61, 40), True), (128, 16, (50, 9), True),
                                              (256, 16, (33, 17), True), (512, 16, (8, 20), True), (256, 8, (5, 30), True),
                                              (32, 16, (300, 211), True), (64, 16, (64, 23), False), (512, 8, (12, 9), False)])
def test_pt_layer_fused_vs_cpu_restatement(fsg, device, c, ns, sizes, train):
    """Fused PointTransformerLayer (fsg_pt_attn_*: one pass per BatchNorm, MFMA channel contraction, atomics for dk/dv)
    against the CPU restatement of seg_model.py:17-53 in fp64: output, gradient of the input features and of every
    parameter, running statistics.  A segment shorter than nsample exercises the padding rule of the packed kNN."""
    from fissure_segmentation_amd.models.pointtransformer.seg_model import PointTransformerLayer
    xyz, feat, off = packed(900 + c + ns, sizes, c)
    ref = fill_state_dict(ref_cpu.PTLayer(c, c, 8, ns), 811 + c).double()
    lay = PointTransformerLayer(c, c, 8, ns)
    lay.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    lay = lay.to(device)
    ref.train(train), lay.train(train)
    xr = torch.from_numpy(feat).double().requires_grad_(True)
    pr = torch.from_numpy(xyz).double().requires_grad_(True)
    yr = ref([pr, xr, torch.from_numpy(off)])
    gr = np.random.default_rng(5).standard_normal(tuple(yr.shape)).astype(np.float32)
    yr.backward(torch.from_numpy(gr).double())
    x = G(feat, device).requires_grad_(True)
    pg = G(xyz, device).requires_grad_(True)
    y = lay([pg, x, G(off, device)])
    y.backward(G(gr, device))
    assert np.linalg.norm(N(pg.grad) - pr.grad.numpy()) <= 1e-3 * np.linalg.norm(pr.grad.numpy())
    yr_np = yr.detach().numpy()
    assert np.abs(N(y) - yr_np).max() <= 1e-4 * max(1.0, np.abs(yr_np).max())
    rgx = xr.grad.numpy()
    assert np.linalg.norm(N(x.grad) - rgx) <= 1e-3 * np.linalg.norm(rgx)
    refp = dict(ref.named_parameters())
    scale = max(float(q.grad.norm()) for q in refp.values())
    for name, prm in lay.named_parameters():   # biases in front of a train-mode BatchNorm / the softmax have zero gradient
        a, b = prm.grad.double().cpu().reshape(-1), refp[name].grad.reshape(-1)
        assert float((a - b).norm()) <= 1e-3 * float(b.norm()) + 1e-5 * scale, name
    if train:
        for name, buf in lay.named_buffers():
            np.testing.assert_allclose(N(buf.double()), dict(ref.named_buffers())[name].numpy(), rtol=1e-4, atol=1e-6, err_msg=name)


def test_pt_layer_fused_vs_unfused_composition(fsg, device):
    """same layer through the separate grouping / linear / BatchNorm / vec_attn ops (the first HIP path) and fused"""
    from fissure_segmentation_amd.models.pointtransformer.seg_model import PointTransformerLayer
    c, ns = 64, 16
    xyz, feat, off = packed(77, (500, 300, 7), c)
    a = fill_state_dict(PointTransformerLayer(c, c, 8, ns), 5).to(device).train()
    b = fill_state_dict(PointTransformerLayer(c, c, 8, ns), 5).to(device).train()
    b.fused = False
    outs = []
    for lay in (a, b):
        x = G(feat, device).requires_grad_(True)
        y = lay([G(xyz, device), x, G(off, device)])
        y.square().sum().backward()
        outs.append((y, x.grad, lay))
    torch.testing.assert_close(outs[0][0], outs[1][0], rtol=1e-4, atol=1e-4)
    assert float((outs[0][1] - outs[1][1]).norm()) <= 1e-3 * float(outs[1][1].norm())
    pb = dict(b.named_parameters())
    scale = max(float(q.grad.norm()) for q in pb.values())
    for name, prm in a.named_parameters():
        assert float((prm.grad - pb[name].grad).norm()) <= 2e-3 * float(pb[name].grad.norm()) + 1e-4 * scale, name


def test_save_load_roundtrip_and_reinstantiate(fsg, device, tmp_path):
    from fissure_segmentation_amd.models.access_models import get_point_seg_model_class
    cls = get_point_seg_model_class("DGCNN")
    net = cls(in_features=3, num_classes=4, k=8, spatial_transformer=False, dynamic=False).to(device).eval()
    clone = type(net)(**net.config)           # train.py:505
    assert clone.dynamic is False and clone.k == 8
    path = str(tmp_path / "model.pth")
    net.save(path)
    back = cls.load(path, device).to(device).eval()
    x = G(cloud(3, 2, 3, 100), device)
    assert torch.equal(net(x), back(x))
    ck = torch.load(path)
    assert set(ck.keys()) == {"config", "model_state"}


def test_predict_full_pointcloud(fsg, device):
    from fissure_segmentation_amd.models.dgcnn import DGCNNSeg
    net = DGCNNSeg(k=8, in_features=3, num_classes=4).to(device).eval()
    pc = G(cloud(4, 1, 3, 700), device)
    with torch.no_grad():
        out = net.predict_full_pointcloud(pc, sample_points=256, n_runs_min=10)
    assert out.shape == (1, 4, 700)
    torch.testing.assert_close(out.sum(1), torch.ones(1, 700, device=device))


@pytest.mark.parametrize("model", ["dgcnn", "dgcnn_static", "pointnet", "pointtransformer"])
def test_predict_full_pointcloud_batched_equals_sequential(fsg, device, model):
    """eval-mode ensembling: the batched form (all runs of a phase in one forward + fsg_ensemble_accumulate_f32) against
    the reference's sequential loop (point_seg_net.py:21-48) under the same generator state.  3000 points, 8 + 2 runs of
    256: the first phase cannot see every point, so the fill-up phase (duplicated indices included) runs too.
    Tolerance 1e-5 on the class probabilities (batch size changes the GEMM tiling, nothing else)."""
    from fissure_segmentation_amd.models.dgcnn import DGCNNSeg
    from fissure_segmentation_amd.models.point_net import PointNetSeg
    from fissure_segmentation_amd.models.pointtransformer.seg_model import PointTransformerCompatibility
    torch.manual_seed(3)
    if model == "dgcnn":
        net = DGCNNSeg(k=8, in_features=3, num_classes=4)
    elif model == "dgcnn_static":
        net = DGCNNSeg(k=8, in_features=3, num_classes=4, dynamic=False)
    elif model == "pointnet":
        net = PointNetSeg(3, 4)
    else:
        net = PointTransformerCompatibility(3, 4)
    net = net.to(device)
    pc = G(cloud(11, 1, 3, 3000), device)
    net.train()
    with torch.no_grad():                      # running statistics away from their initial values
        for _ in range(2):
            net(pc[..., :512].contiguous())
    net.eval()
    assert net._ensemble_batchable(pc) is False      # grad mode on: sequential form
    with torch.no_grad():
        assert net._ensemble_batchable(pc)
        torch.manual_seed(77)
        got = net.predict_full_pointcloud(pc, sample_points=256, n_runs_min=10)
        net._ensemble_batchable = lambda _pc: False
        torch.manual_seed(77)
        want = net.predict_full_pointcloud(pc, sample_points=256, n_runs_min=10)
    assert got.shape == want.shape == (1, 4, 3000)
    torch.testing.assert_close(got, want, rtol=0, atol=1e-5)
    net.train()
    del net._ensemble_batchable
    assert net._ensemble_batchable(pc) is False      # train mode couples the batch through BatchNorm


def test_predict_full_pointcloud_regression_batched(fsg, device):
    """DGCNNReg.predict_full_pointcloud (mean of the runs' outputs): batched eval-mode form == sequential loop, 1e-5"""
    from fissure_segmentation_amd.models.dgcnn import DGCNNReg
    torch.manual_seed(4)
    net = DGCNNReg(k=8, in_features=3, num_classes=6).to(device)
    pc = G(cloud(12, 2, 3, 1500), device)
    net.train()
    with torch.no_grad():
        net(pc[..., :512].contiguous())
    net.eval()
    with torch.no_grad():
        torch.manual_seed(5)
        got = net.predict_full_pointcloud(pc, sample_points=256, n_runs_min=7)
        net._ensemble_batchable = lambda _pc: False
        torch.manual_seed(5)
        want = net.predict_full_pointcloud(pc, sample_points=256, n_runs_min=7)
    assert got.shape == want.shape == (2, 6, 1)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)


def test_knn_on_interleaved_cloud_views(fsg, device):
    """Clouds handed over as strided views (predict_full_pointcloud builds its batch as a permuted view: cloud stride 256,
    channel stride 512 for N = 256): the operand loads must stay inside each cloud's own extent -- the view of the last
    cloud ends exactly at the end of the storage -- and the result must match the contiguous copy."""
    R, C, S, k = 2, 3, 256, 8
    store = torch.from_numpy(cloud(41, 1, C, R * S)).to(device)                 # (1, 3, 512)
    view = store.view(1, C, R, S).permute(2, 0, 1, 3).reshape(R, C, S)          # strides (256, 512, 1): no copy
    assert view.stride() == (S, R * S, 1) and view.data_ptr() == store.data_ptr()
    got = fsg.functional.knn_graph(view, k, c_knn=3, fix_diag=True)
    want = fsg.functional.knn_graph(view.contiguous(), k, c_knn=3, fix_diag=True)
    assert torch.equal(got, want)
    idx_o, _ = c_api.knn_dense(N(view.contiguous()), k, fix_diag=True)
    assert np.array_equal(N(got), idx_o)


@pytest.mark.parametrize("Np,m,start", [(2000, 256, 0), (2000, 256, 1234), (777, 100, 776), (50, 50, 0), (30, 40, 0)])
def test_farthest_point_sampling_drop_in(fsg, device, Np, m, start):
    """utils.general_utils.farthest_point_sampling (one fsg_fps_f32 launch) against the reference's pure-torch loop
    (dseg_ae_regularization.py:30-43, restated in oracle/ref_cpu.py) from the same start point: identical index lists"""
    from fissure_segmentation_amd.utils.general_utils import farthest_point_sampling
    from oracle import ref_cpu
    pts = torch.from_numpy(cloud(Np + m, 1, 3, Np)).transpose(1, 2).contiguous()      # (1, Np, 3)
    got_p, got_i = farthest_point_sampling(pts.to(device), m, start=start)
    if Np <= m:
        assert got_p.shape == (1, Np, 3) and torch.equal(got_i, torch.arange(Np))
        return
    want_p, want_i = ref_cpu.farthest_point_sampling(pts, m, start)
    assert torch.equal(got_i.cpu(), want_i)
    assert torch.equal(got_p.cpu(), want_p)


def test_point_augmentation_against_oracle_and_scipy(fsg, device):
    """augmentations.point_augmentation / transform_points (fsg_sample_transform_f32) against the CPU restatement
    (oracle/ref_cpu.py) with the same random draws, and the rotation matrices against scipy's exponential map.
    Row-vector convention, rotate -> scale -> translate; 1e-6 absolute (coordinates are O(1))."""
    from scipy.spatial.transform import Rotation
    from fissure_segmentation_amd import augmentations as A
    from oracle import ref_cpu
    pc = torch.from_numpy(cloud(21, 5, 3, 1000))
    torch.manual_seed(9)
    out, tf = A.point_augmentation(pc.to(device))
    torch.manual_seed(9)
    log_rot, tr, sc = A.random_transform_parameters(5, device)
    want = ref_cpu.augment_points(pc, log_rot.cpu(), tr.cpu(), sc.cpu())
    torch.testing.assert_close(out.cpu(), want, rtol=0, atol=1e-6)
    R = A.so3_exp_map(log_rot).cpu().numpy()
    np.testing.assert_allclose(R, Rotation.from_rotvec(log_rot.cpu().numpy()).as_matrix(), atol=1e-6)
    ang = log_rot.norm(dim=1).cpu()
    torch.testing.assert_close(ang, torch.full((5,), 0.1 * np.pi), rtol=1e-5, atol=0)          # |angle| = 0.1 pi
    assert (tr.abs() <= 0.1).all() and (sc <= 1).all() and (sc >= 0.9).all()
    # the 4x4 matrices compose and invert like the reference's Transform3d (data.py:553-582 relies on this)
    back = tf.inverse().transform_points(out.transpose(1, 2)).transpose(1, 2)
    torch.testing.assert_close(back.cpu(), pc, rtol=0, atol=2e-6)
    ident = tf.compose(tf.inverse()).get_matrix().cpu()
    torch.testing.assert_close(ident, torch.eye(4).expand(5, 4, 4), rtol=0, atol=1e-6)
    centred = A.transform_points_with_centering(pc.to(device), tf).cpu()
    c = pc.mean(2, keepdim=True)
    torch.testing.assert_close(centred, ref_cpu.augment_points(pc - c, log_rot.cpu(), tr.cpu(), sc.cpu()) + c, rtol=0, atol=1e-6)
    with pytest.raises(RuntimeError, match="GPU"):
        A.point_augmentation(pc)                          # no CPU fallback


@pytest.mark.parametrize("C,binary", [(3, False), (7, True)])
def test_sample_and_augment_is_the_dataset_item(fsg, device, C, binary):
    """augmentations.sample_and_augment == PointDataset.__getitem__ (data.py:435-460, restated in oracle/ref_cpu.py) item by
    item, with the same draws: augmented coordinates, untouched feature rows, the same column subset for the labels"""
    from fissure_segmentation_amd import augmentations as A
    from oracle import ref_cpu
    B, Nf, S = 4, 3000, 1024
    x = torch.from_numpy(cloud(31, B, C, Nf))
    lbl = torch.from_numpy(np.random.default_rng(3).integers(0, 4, (B, Nf)))
    torch.manual_seed(12)
    xs, ls, tf = A.sample_and_augment(x.to(device), lbl.to(device), S, augment=True, binary=binary)
    torch.manual_seed(12)
    log_rot, tr, sc = A.random_transform_parameters(B, device)
    sample = A.random_subsets(B, Nf, S, device).cpu()
    assert all(len(set(r.tolist())) == S for r in sample) and int(sample.min()) >= 0 and int(sample.max()) < Nf
    assert xs.shape == (B, C, S) and ls.shape == (B, S) and len(tf) == B
    for b in range(B):
        wx, wl = ref_cpu.dataset_item(x[b], lbl[b], sample[b], log_rot[b:b + 1].cpu(), tr[b:b + 1].cpu(), sc[b:b + 1].cpu(),
                                      binary=binary)
        torch.testing.assert_close(xs[b].cpu(), wx, rtol=0, atol=1e-6)
        assert torch.equal(xs[b, 3:].cpu(), wx[3:])       # feature rows are copied bit for bit
        assert torch.equal(ls[b].cpu(), wl)
    # no augmentation, no sub-sampling: a column permutation of the input
    xs2, ls2, tf2 = A.sample_and_augment(x.to(device), None, None, augment=False)
    assert tf2 is None and ls2 is None and xs2.shape == x.shape
    assert torch.equal(xs2.sort(dim=2).values.cpu(), x.sort(dim=2).values)


def test_sample_transform_error_paths(fsg, device):
    import ctypes
    x = torch.zeros(2, 2, 10, device=device)
    out = torch.zeros(2, 2, 10, device=device)
    aff = torch.zeros(2, 12, device=device)
    Pp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    with pytest.raises(RuntimeError, match="three coordinate rows"):
        fsg._lib.call("fsg_sample_transform_f32", Pp(x), 2, 2, 10, None, 10, Pp(aff), Pp(out), None)
    with pytest.raises(RuntimeError, match="S must equal N"):
        fsg._lib.call("fsg_sample_transform_f32", Pp(x), 2, 2, 10, None, 5, None, Pp(out), None)
    fsg._lib.call("fsg_sample_transform_f32", Pp(x), 0, 2, 10, None, 10, None, Pp(out), None)     # empty batch: no-op


def test_ensemble_accumulate_kernel(fsg, device):
    """fsg_ensemble_accumulate_f32 against the loop it replaces, in numpy: run order per point, duplicates inside a run
    (highest slot wins), indices outside the cloud ignored, B = 2 clouds sharing the subsets, 5 classes"""
    import ctypes
    rng = np.random.default_rng(5)
    R, B, cls, S, P_ = 7, 2, 5, 300, 1000
    logits = rng.standard_normal((R, B, cls, S)).astype(np.float32) * 3
    pts = np.stack([rng.permutation(P_)[:S] for _ in range(R)]).astype(np.int64)
    pts[2, 10:20] = pts[2, 0]          # duplicates inside run 2
    pts[4, 5] = P_ + 3                 # outside the cloud
    pts[4, 6] = -1
    acc0 = rng.standard_normal((B, cls, P_)).astype(np.float32)
    want = acc0.copy()
    for r in range(R):
        e = np.exp(logits[r] - logits[r].max(1, keepdims=True))
        sm = (e / e.sum(1, keepdims=True)).astype(np.float32)
        last = {}
        for s_, p in enumerate(pts[r]):
            if 0 <= p < P_:
                last[int(p)] = s_
        for p, s_ in last.items():
            want[:, :, p] += sm[:, :, s_]
    lt, pt, acc = G(logits, device), torch.from_numpy(pts).to(device), G(acc0, device)
    ws = torch.empty(fsg._lib.lib.fsg_ensemble_accumulate_workspace_bytes(R, P_) // 4, dtype=torch.int32, device=device)
    Pp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    fsg._lib.call("fsg_ensemble_accumulate_f32", Pp(lt), R, B, cls, S, Pp(pt), P_, Pp(acc), Pp(ws), st)
    np.testing.assert_allclose(N(acc), want, rtol=1e-6, atol=1e-6)
    fsg._lib.call("fsg_ensemble_accumulate_f32", Pp(lt), 0, B, cls, S, Pp(pt), P_, Pp(acc), Pp(ws), st)   # empty: no-op
    with pytest.raises(RuntimeError, match="cls"):
        fsg._lib.call("fsg_ensemble_accumulate_f32", Pp(lt), R, B, 33, S, Pp(pt), P_, Pp(acc), Pp(ws), st)
    with pytest.raises(RuntimeError, match="NULL"):
        fsg._lib.call("fsg_ensemble_accumulate_f32", Pp(lt), R, B, cls, S, Pp(pt), P_, Pp(acc), None, st)


# --------------------------------------------------------------------------- fused EdgeConv vs unfused composition
@pytest.mark.parametrize("B,C,Np,k,Co,train", [(2, 64, 300, 20, 64, True), (1, 3, 77, 7, 128, True),
                                               (3, 15, 513, 40, 64, False), (2, 128, 256, 8, 256, True)])
def test_edgeconv1_fused_vs_unfused(fsg, device, B, C, Np, k, Co, train):
    """csrc/edgeconv.hip against [fsg_edge_gather -> Conv2d -> BatchNorm2d -> LeakyReLU -> max] (itself pinned to the
    golden EdgeConv vectors above); negative BN scales exercise the min-selection branch."""
    from fissure_segmentation_amd.norm import BatchNorm2d
    torch.manual_seed(C + Co)
    conv = torch.nn.Conv2d(2 * C, Co, 1, bias=False).to(device)
    bns = [BatchNorm2d(Co).to(device) for _ in range(2)]
    with torch.no_grad():
        w = torch.rand(Co, device=device) + 0.5
        w[::3] *= -1
        bias, rm, rv = torch.randn(Co, device=device), torch.randn(Co, device=device), torch.rand(Co, device=device) + 0.5
        for bn in bns:
            bn.weight.copy_(w)
            bn.bias.copy_(bias)
            bn.running_mean.copy_(rm)
            bn.running_var.copy_(rv)
            bn.train(train)
    x = G(cloud(77, B, C, Np), device)
    idx = fsg.functional.knn_graph(x, k)
    gr = torch.randn(B, Co, Np, device=device)
    outs = []
    for fused, bn in zip((True, False), bns):
        xt = x.clone().requires_grad_(True)
        conv.weight.grad = None
        if fused:
            y = fsg.functional.edgeconv1(xt, idx, conv.weight, bn, 0.2)
        else:
            y = torch.nn.functional.leaky_relu(bn(conv(fsg.functional.edge_features(xt, idx))), 0.2).max(-1)[0]
        y.backward(gr)
        outs.append((y.detach(), xt.grad, conv.weight.grad.clone(), bn.weight.grad, bn.bias.grad, bn.running_mean.clone(),
                     bn.running_var.clone(), bn.num_batches_tracked.clone()))
    names = ["out", "grad_x", "grad_w", "grad_gamma", "grad_beta", "running_mean", "running_var", "batches"]
    for n, a, b in zip(names, *outs):
        a, b = a.double(), b.double()
        tol = 1e-4 if n in ("out", "running_mean", "running_var", "batches") else 2e-3
        assert float((a - b).abs().max()) <= tol * (1.0 + float(b.abs().max())), (n, float((a - b).abs().max()))


@pytest.mark.parametrize("B,C,Np,k,C2,train", [(2, 3, 300, 20, 64, True), (1, 15, 77, 7, 64, True), (2, 3, 130, 40, 128, True),
                                               (3, 6, 513, 16, 64, False), (8, 3, 2048, 20, 64, True), (2, 3, 100, 3, 64, True),
                                               (2, 3, 257, 30, 64, True), (1, 3, 90, 64, 64, True)])
def test_edgeconv2_fused_vs_unfused(fsg, device, B, C, Np, k, C2, train):
    """csrc/edgeconv2.hip (two-layer fused EdgeConv, MFMA) against the unfused composition
    [fsg_edge_gather -> (Conv2d -> BatchNorm2d -> LeakyReLU) x 2 -> max].  The C2 = 64 cases run the split-image kernels:
    k = 20 / 7 / 16 -> 64-row tiles of 3 / 9 / 4 points (9 > 4: the per-point table of the backward is filled by its loop
    form), k = 3 -> the 16-point cap of a tile, k = 30 -> 128-row tiles (4 points), k = 64 -> one point per 64-row tile; ragged
    last tiles everywhere.  C2 = 128: the fp32-MFMA kernels."""
    from fissure_segmentation_amd.norm import BatchNorm2d
    torch.manual_seed(C + C2)
    conv1 = torch.nn.Conv2d(2 * C, 64, 1, bias=False).to(device)
    conv2 = torch.nn.Conv2d(64, C2, 1, bias=False).to(device)
    sets = []
    params = []
    for width in (64, C2):
        w = torch.rand(width, device=device) + 0.5
        w[::3] *= -1
        params.append((w, torch.randn(width, device=device), torch.randn(width, device=device) * 0.3,
                       torch.rand(width, device=device) + 0.5))
    for _ in range(2):
        bns = []
        for width, (w, bias, rm, rv) in zip((64, C2), params):
            bn = BatchNorm2d(width).to(device)
            with torch.no_grad():
                bn.weight.copy_(w); bn.bias.copy_(bias); bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
            bn.train(train)
            bns.append(bn)
        sets.append(bns)
    x = G(cloud(78, B, C, Np), device)
    idx = fsg.functional.knn_graph(x, k, c_knn=3)
    gr = torch.randn(B, C2, Np, device=device)
    outs = []
    lrelu = torch.nn.functional.leaky_relu
    for fused, (bn1, bn2) in zip((True, False), sets):
        xt = x.clone().requires_grad_(True)
        conv1.weight.grad = conv2.weight.grad = None
        if fused:
            y = fsg.functional.edgeconv2(xt, idx, conv1.weight, bn1, conv2.weight, bn2, 0.2)
        else:
            e = lrelu(bn1(conv1(fsg.functional.edge_features(xt, idx))), 0.2)
            y = lrelu(bn2(conv2(e)), 0.2).max(-1)[0]
        y.backward(gr)
        outs.append((y.detach(), xt.grad, conv1.weight.grad.clone(), conv2.weight.grad.clone(), bn1.weight.grad,
                     bn1.bias.grad, bn2.weight.grad, bn2.bias.grad, bn1.running_mean.clone(), bn1.running_var.clone(),
                     bn2.running_mean.clone(), bn2.running_var.clone()))
    names = ["out", "grad_x", "grad_w1", "grad_w2", "grad_gamma1", "grad_beta1", "grad_gamma2", "grad_beta2", "rm1", "rv1",
             "rm2", "rv2"]
    for n, a, b in zip(names, *outs):
        a, b = a.double(), b.double()
        if n == "out" or n.startswith("r"):
            assert float((a - b).abs().max()) <= 1e-4 * (1.0 + float(b.abs().max())), (n, float((a - b).abs().max()))
        else:  # gradients in norm (arg-max near-ties re-route single entries)
            assert float((a - b).norm()) <= 5e-3 * float(b.norm()) + 1e-5, (n, float((a - b).norm()), float(b.norm()))


@pytest.mark.parametrize("M,C,train,slope", [(1000, 64, True, 0.2), (16384, 1024, True, 0.2), (777, 256, False, 0.2),
                                             (300, 128, True, 0.0), (129, 64, True, 1.0)])
def test_bn_act_fused_vs_torch(fsg, device, M, C, train, slope):
    from fissure_segmentation_amd.norm import BatchNorm1d
    torch.manual_seed(M + C)
    bns = [BatchNorm1d(C).to(device) for _ in range(2)]
    w, bias = torch.rand(C, device=device) + 0.5, torch.randn(C, device=device)
    w[::4] *= -1
    rm, rv = torch.randn(C, device=device), torch.rand(C, device=device) + 0.5
    for bn in bns:
        with torch.no_grad():
            bn.weight.copy_(w); bn.bias.copy_(bias); bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
        bn.train(train)
    y0 = torch.randn(M, C, device=device) * 3 + 5      # mean >> spread on purpose
    g = torch.randn(M, C, device=device)
    res = []
    for fused, bn in zip((True, False), bns):
        y = y0.clone().requires_grad_(True)
        out = fsg.functional.bn_act(y, bn, slope) if fused else torch.nn.functional.leaky_relu(bn(y), slope)
        out.backward(g)
        res.append((out.detach(), y.grad, bn.weight.grad, bn.bias.grad, bn.running_mean.clone(), bn.running_var.clone()))
    for n, a, b in zip(["out", "grad_y", "grad_gamma", "grad_beta", "rm", "rv"], *res):
        a, b = a.double(), b.double()
        assert float((a - b).abs().max()) <= 2e-4 * (1.0 + float(b.abs().max())), (n, float((a - b).abs().max()))


@pytest.mark.parametrize("B,Np,C,train", [(8, 2048, 1024, True), (3, 130, 64, True), (2, 257, 128, False)])
def test_bn_act_max_fused_vs_torch(fsg, device, B, Np, C, train):
    from fissure_segmentation_amd.norm import BatchNorm1d
    torch.manual_seed(Np + C)
    bns = [BatchNorm1d(C).to(device) for _ in range(2)]
    w, bias = torch.rand(C, device=device) + 0.5, torch.randn(C, device=device)
    w[::4] *= -1
    rm, rv = torch.randn(C, device=device), torch.rand(C, device=device) + 0.5
    for bn in bns:
        with torch.no_grad():
            bn.weight.copy_(w); bn.bias.copy_(bias); bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
        bn.train(train)
    y0 = torch.randn(B, Np, C, device=device) * 2 + 1
    g = torch.randn(B, C, device=device)
    res = []
    for fused, bn in zip((True, False), bns):
        y = y0.clone().requires_grad_(True)
        if fused:
            out = fsg.functional.bn_act_max(y, bn, 0.2)
        else:
            out = torch.nn.functional.leaky_relu(bn(y.view(B * Np, C)), 0.2).view(B, Np, C).max(dim=1)[0]
        out.backward(g)
        res.append((out.detach(), y.grad, bn.weight.grad, bn.bias.grad, bn.running_mean.clone(), bn.running_var.clone()))
    for n, a, b in zip(["out", "grad_y", "grad_gamma", "grad_beta", "rm", "rv"], *res):
        a, b = a.double(), b.double()
        assert float((a - b).abs().max()) <= 2e-4 * (1.0 + float(b.abs().max())), (n, float((a - b).abs().max()))


def test_reverse_graph_is_the_transpose(fsg, device):
    x = G(cloud(3, 2, 3, 500), device)
    idx = fsg.functional.knn_graph(x, 12)
    rowptr, col = fsg.functional.reverse_graph(idx)
    rowptr, col, idxc = N(rowptr), N(col), N(idx)
    for b in range(2):
        assert rowptr[b, 0] == 0 and rowptr[b, -1] == 500 * 12
        edges = set()
        for j in range(500):
            for e in col[b, rowptr[b, j]:rowptr[b, j + 1]]:
                i, s = int(e) >> 6, int(e) & 63
                assert idxc[b, i, s] == j
                edges.add((i, s))
        assert len(edges) == 500 * 12


def test_hipgraph_replay_matches_eager_training(fsg, device):
    """bench.py replays the training step as a hipGraph: three replayed steps must land on the same weights and loss as
    three eagerly launched steps from the same initial state."""
    import copy
    import torch.nn.functional as F
    from fissure_segmentation_amd.models.dgcnn import DGCNNSeg
    torch.manual_seed(0)
    base = DGCNNSeg(k=8, in_features=3, num_classes=4).to(device).train()
    x = G(cloud(11, 2, 3, 256), device)
    y = torch.randint(0, 4, (2, 256), device=device)

    def one_step(net, opt):
        for p in net.parameters():
            p.grad = None
        loss = F.cross_entropy(net(x), y)
        loss.backward()
        opt.step()
        return loss

    def reset(net, opt):   # back to `base` IN PLACE (a captured graph keeps pointing at these very tensors)
        with torch.no_grad():
            for (_, a), (_, b) in zip(net.state_dict().items(), base.state_dict().items()):
                a.copy_(b)
            for st in opt.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()

    runs = []
    for use_graph in (False, True):
        net = copy.deepcopy(base)
        opt = torch.optim.Adam(net.parameters(), lr=1e-3, capturable=True)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            one_step(net, opt)      # creates the optimizer state, sets kernel attributes, warms the allocator
        torch.cuda.current_stream().wait_stream(side)
        losses = []
        if use_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_loss = one_step(net, opt)
            reset(net, opt)
            for _ in range(3):
                graph.replay()
                losses.append(float(static_loss))
        else:
            reset(net, opt)
            for _ in range(3):
                losses.append(float(one_step(net, opt)))
        runs.append((losses, [p.detach().clone() for p in net.parameters()]))
    (l0, w0), (l1, w1) = runs
    assert l0[0] > l0[-1]                                # training moves
    np.testing.assert_allclose(l1, l0, rtol=2e-4)
    # weights: parameters with a mathematically zero gradient (e.g. the BatchNorm bias in front of the global max-pool,
    # whose shift the next train-mode BatchNorm removes) receive rounding-noise gradients that Adam turns into +-lr steps
    # of random sign, so the bar is 3 steps x lr; the loss trajectory above is the sharp check
    for a, b in zip(w0, w1):
        torch.testing.assert_close(a, b, rtol=5e-3, atol=3.5e-3)


@pytest.mark.parametrize("bf16", [False, True])
def test_hipgraph_replay_matches_eager_pointtransformer(fsg, device, bf16):
    """The PointTransformer step as a hipGraph (what bench.py --workload c3 replays): its backward leaves the split sums of the
    small Linears' weight gradients to ONE launch queued as an end-of-pass callback of the autograd engine -- that launch, the
    packed q/k/v copy and the single zero fill of the backward scratch must all be inside the captured graph.  Two replays (the
    second on new input values copied into the static buffer) against the eager pass at the same weights: same loss (1e-5), and
    the gradient vector within 1e-4 (fp32; 5e-2 with bf16 operands) in relative L2 or five times the difference between two
    EAGER passes, whichever is larger
    (the attention backward scatters with fp32 atomics: not bit-reproducible, and the net amplifies the last bits)."""
    from fissure_segmentation_amd.models.pointtransformer.seg_model import PointTransformerCompatibility
    F_hip = fsg.functional
    torch.manual_seed(321)
    net = PointTransformerCompatibility(6, 4).to(device).train()      # (torch's initialisation: at fill_state_dict weights the net
    # amplifies the last-bit differences of the atomics a thousandfold, run to run)
    params = list(net.parameters())
    xs = [G(cloud(77 + i, 2, 6, 512), device) for i in range(2)]
    y = torch.randint(0, 4, (2, 512), device=device)
    x_static = xs[0].clone()

    def fwd_bwd(x):
        for p in params:
            p.grad = None
        loss = torch.nn.functional.cross_entropy(net(x), y)
        loss.backward()
        return loss

    old = F_hip.set_bf16_linear(bf16)
    try:
        with F_hip.mfma_operands("bf16" if bf16 else "f32"):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fwd_bwd(x_static)                       # kernel attributes, memoised constants, allocator warm-up
            torch.cuda.current_stream().wait_stream(side)
            bn_state = {k: v.clone() for k, v in net.state_dict().items() if "running" in k or "num_batches" in k}
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_loss = fwd_bwd(x_static)
                static_grads = [p.grad for p in params]
            for i, x in enumerate(xs):
                x_static.copy_(x)
                graph.replay()
                got = [g.detach().clone() for g in static_grads]
                got_loss = float(static_loss)
                net.load_state_dict({**net.state_dict(), **bn_state})           # the eager passes start from the same buffers
                fwd_bwd(x)
                again = [p.grad.detach().clone() for p in params]                 # run-to-run noise of the eager pass itself
                net.load_state_dict({**net.state_dict(), **bn_state})
                want_loss = float(fwd_bwd(x))
                assert abs(got_loss - want_loss) <= 1e-5 * abs(want_loss), (i, got_loss, want_loss)
                # the whole gradient vector against the eager one, next to the run-to-run difference of the eager pass itself (most
                # parameters of this net differ by more than 1e-3 of their size between two eager passes: the atomics' order,
                # amplified); a weight gradient left unreduced or unflushed would put O(1) here
                flat = lambda gs: torch.cat([t.reshape(-1) for t in gs]).double()  # noqa: E731
                ge, g2, gr = flat([p.grad for p in params]), flat(again), flat(got)
                noise = float((g2 - ge).norm() / ge.norm())
                err = float((gr - ge).norm() / ge.norm())
                print("\nPT replay vs eager, input", i, "bf16" if bf16 else "f32", ": rel. L2", err, "eager vs eager", noise)
                # measured: fp32 3e-6 (eager vs eager 1-3e-6); bf16 operands 0.7-1.7e-2 (eager vs eager 0.15-0.7e-2: a last-bit
                # difference flips bf16 roundings)
                assert err <= max(5e-2 if bf16 else 1e-4, 5.0 * noise), (i, err, noise)
    finally:
        F_hip.set_bf16_linear(old)


def test_graphed_train_step_helper(fsg, device):
    """fissure_segmentation_amd.graph.GraphedTrainStep: the captured step (DGCNN-seg, CE + generalised Dice, FlatAdam)
    must follow the same loss trajectory as the eager loop, also when fresh batches are copied into its static buffers."""
    import copy
    from fissure_segmentation_amd.graph import GraphedTrainStep
    from fissure_segmentation_amd.losses.nnu_loss import NNULoss
    from fissure_segmentation_amd.models.dgcnn import DGCNNSeg
    from fissure_segmentation_amd.optim import FlatAdam
    torch.manual_seed(0)
    base = DGCNNSeg(k=8, in_features=3, num_classes=4).to(device).train()
    crit = NNULoss(torch.tensor([0.5, 1.0, 1.5, 2.0])).to(device)
    batches = [(G(cloud(20 + i, 2, 3, 256), device), torch.randint(0, 4, (2, 256), device=device)) for i in range(4)]

    eager = copy.deepcopy(base)
    opt_e = FlatAdam(eager.parameters(), lr=1e-3, capturable=True)
    graphed = copy.deepcopy(base)
    opt_g = FlatAdam(graphed.parameters(), lr=1e-3, capturable=True)
    step = GraphedTrainStep(graphed, crit, opt_g, *batches[0], warmup=2)
    assert step.captured
    for _ in range(2):                                # the helper took 2 warm-up steps on batch 0 (capturing executes nothing)
        opt_e.zero_grad()
        crit(eager(batches[0][0]), batches[0][1])[0].backward()
        opt_e.step()
    le, lg = [], []
    for xb, yb in batches[1:] + batches[1:]:
        opt_e.zero_grad()
        loss = crit(eager(xb), yb)[0]
        loss.backward()
        opt_e.step()
        le.append(float(loss))
        lg.append(float(step(xb, yb)))
    # the first replays must reproduce the eager losses; later ones drift apart the way two eager runs do (the in-edge order
    # of the reverse graph, hence the summation order of the EdgeConv gradients, differs from run to run, and Adam turns
    # rounding noise on zero-gradient parameters into +-lr steps)
    np.testing.assert_allclose(lg[:3], le[:3], rtol=5e-4)
    np.testing.assert_allclose(lg, le, rtol=2e-2)
    assert le[-1] < le[0]


def test_dgcnn_backward_is_bitwise_reproducible(fsg, device):
    """No float atomics on the DGCNN path and a sorted reverse graph: two backward passes from the same state give
    bit-identical gradients (BASELINE config 2 shape)."""
    from fissure_segmentation_amd.losses.nnu_loss import NNULoss
    from fissure_segmentation_amd.models.dgcnn import DGCNNSeg
    torch.manual_seed(0)
    net = DGCNNSeg(k=20, in_features=3, num_classes=4).to(device).train()
    crit = NNULoss(torch.tensor([0.4, 1.2, 1.2, 1.2])).to(device)
    x = G(cloud(77, 8, 3, 2048), device)
    y = torch.randint(0, 4, (8, 2048), device=device)
    state = {k: v.clone() for k, v in net.state_dict().items()}
    grads = []
    for _ in range(2):
        net.load_state_dict(state)
        net.zero_grad(set_to_none=True)
        crit(net(x), y)[0].backward()
        grads.append([p.grad.clone() for p in net.parameters()])
    for a, b in zip(*grads):
        assert torch.equal(a, b)


def test_chamfer_backward_is_reproducible_and_matches_autograd(fsg, device):
    """ordered (reverse-graph) Chamfer backward: bit-identical between runs, equal to torch autograd of the same loss"""
    from fissure_segmentation_amd.losses.chamfer_loss import ChamferLoss
    rng = np.random.default_rng(9)
    a = rng.uniform(-1, 1, (3, 700, 3)).astype(np.float32)
    b = rng.uniform(-1, 1, (3, 450, 3)).astype(np.float32)
    b[:, 5] = b[:, 4]   # duplicate targets: ties go to the lower index, the other one receives no gradient
    grads = []
    fsg.functional.set_deterministic(True)
    try:
        for _ in range(2):
            at, bt = G(a, device).requires_grad_(True), G(b, device).requires_grad_(True)
            ChamferLoss()(at, bt).backward()
            grads.append((at.grad.clone(), bt.grad.clone()))
    finally:
        fsg.functional.set_deterministic(False)
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])
    at, bt = G(a, device).requires_grad_(True), G(b, device).requires_grad_(True)
    ChamferLoss()(at, bt).backward()          # default: atomics for the scattered half
    torch.testing.assert_close(at.grad, grads[0][0], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(bt.grad, grads[0][1], rtol=1e-5, atol=1e-7)
    ar, br = torch.from_numpy(a).double().requires_grad_(True), torch.from_numpy(b).double().requires_grad_(True)
    d = ((ar[:, :, None, :] - br[:, None, :, :]) ** 2).sum(-1)
    (d.min(2).values.mean(1).mean() + d.min(1).values.mean(1).mean()).backward()
    np.testing.assert_allclose(N(grads[0][0]), ar.grad.numpy(), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(N(grads[0][1]), br.grad.numpy(), rtol=1e-4, atol=1e-7)


def test_new_entry_points_reject_bad_arguments(fsg, device):
    """error behaviour of the C ABI behind the Python wrappers: bad shapes raise, nothing is launched"""
    F = fsg.functional
    x = torch.randn(10, 48, device=device)
    with pytest.raises(ValueError):                                   # plane width outside {32,...,512}
        F.pt_attn(torch.randn(10, 3, device=device), torch.zeros(10, 8, dtype=torch.int32, device=device),
                  torch.randn(10, 144, device=device), None, None)
    with pytest.raises(ValueError):                                   # one class only
        F.nnu_loss(torch.randn(2, 1, 16, device=device), torch.zeros(2, 16, dtype=torch.int64, device=device))
    with pytest.raises(RuntimeError):                                 # CPU tensors: no fallback
        F.nnu_loss(torch.randn(2, 4, 16), torch.zeros(2, 16, dtype=torch.int64))
    import ctypes
    rc = fsg._lib.lib.fsg_bn_rows_fwd_f32(ctypes.c_void_p(x.data_ptr()), None, ctypes.c_void_p(x.data_ptr()),
                                          ctypes.c_void_p(x.data_ptr()), None, None, 10, 48, 0, 0.1, 1e-5, 1,
                                          ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(x.data_ptr()),
                                          ctypes.c_void_p(x.data_ptr()), None, None)
    assert rc != 0 and b"C in {32,64,128,256,512}" in fsg._lib.lib.fsg_last_error()
    rc = fsg._lib.lib.fsg_gemm_small_f32(ctypes.c_void_p(x.data_ptr()), 1, 1, ctypes.c_void_p(x.data_ptr()), 1, 1, None,
                                         ctypes.c_void_p(x.data_ptr()), 1, 4, 4, 0, None, None)
    assert rc != 0 and b"bad shape" in fsg._lib.lib.fsg_last_error()


@pytest.mark.parametrize("B,C,Np,k", [(8, 3, 2048, 20), (8, 64, 2048, 20), (4, 3, 8192, 40), (4, 64, 8192, 40), (8, 128, 4096, 20)])
def test_knn_properties_at_full_size(fsg, device, B, C, Np, k):
    """Size-independent properties at the BASELINE sizes (the oracle comparison runs on smaller clouds): every row is
    sorted ascending by (distance, index), holds k distinct in-range indices, starts with the query itself at distance 0
    (self loop, fix_diag), its distances agree with a direct fp64 evaluation, and no point outside the list is closer
    than the k-th entry (checked on a sample of rows against the full distance row)."""
    x = cloud(7000 + Np + C, B, C, Np)
    xt = G(x, device)
    idx, dist = fsg.functional.knn_graph(xt, k, return_dist=True)
    idx_n, dist_n = N(idx), N(dist)
    assert idx_n.min() >= 0 and idx_n.max() < Np
    assert np.array_equal(idx_n[:, :, 0], np.broadcast_to(np.arange(Np), (B, Np))) and np.all(dist_n[:, :, 0] == 0)
    d0, d1 = dist_n[:, :, :-1], dist_n[:, :, 1:]
    assert np.all((d0 < d1) | ((d0 == d1) & (idx_n[:, :, :-1] < idx_n[:, :, 1:])))
    assert np.all(np.sort(idx_n, axis=2)[:, :, 1:] != np.sort(idx_n, axis=2)[:, :, :-1])          # distinct
    rng = np.random.default_rng(0)
    x64 = x.astype(np.float64)
    for b, i in zip(rng.integers(0, B, 24), rng.integers(0, Np, 24)):
        full = ((x64[b] - x64[b][:, i:i + 1]) ** 2).sum(0)
        full[i] = 0.0
        got = dist_n[b, i].astype(np.float64)
        scale = max(1.0, float(np.abs(x64[b]).max()) ** 2 * C)
        assert np.abs(got - full[idx_n[b, i]]).max() <= 1e-5 * scale
        outside = np.delete(full, idx_n[b, i])
        assert outside.min() >= got[-1] - 1e-5 * scale


def test_fps_and_segment_knn_properties(fsg, device):
    """packed clouds at the PointTransformer sizes: samples are distinct and stay inside their segment; neighbour lists
    are sorted, in-segment and start with the query point itself"""
    sizes = [2048] * 8
    xyz, _, off = packed(31, sizes)
    new_off = np.cumsum([s // 4 for s in sizes]).astype(np.int32)
    fi = N(fsg.functional.fps(G(xyz, device), G(off, device), G(new_off, device), int(new_off[-1])))
    starts = np.concatenate([[0], off[:-1]])
    qs = np.concatenate([[0], new_off[:-1]])
    for s in range(len(sizes)):
        seg = fi[qs[s]:new_off[s]]
        assert seg[0] == starts[s] and seg.min() >= starts[s] and seg.max() < off[s] and len(np.unique(seg)) == len(seg)
    idx, d2 = fsg.functional.knn_segment(16, G(xyz, device), G(xyz, device), G(off, device), G(off, device))
    idx_n, d2_n = N(idx), N(d2)
    seg_of = np.repeat(np.arange(len(sizes)), sizes)
    assert np.array_equal(idx_n[:, 0], np.arange(len(xyz))) and np.all(d2_n[:, 0] == 0)
    assert np.all(seg_of[idx_n] == seg_of[:, None])
    assert np.all((d2_n[:, :-1] < d2_n[:, 1:]) | ((d2_n[:, :-1] == d2_n[:, 1:]) & (idx_n[:, :-1] < idx_n[:, 1:])))


@pytest.mark.gpu
@pytest.mark.parametrize("wd", [0.0, 1e-2])
def test_flat_adam_matches_torch_adam(wd):
    """optim.FlatAdam on the GPU (csrc/adam.hip, fsg_adam_flat_f32) == torch.optim.Adam over the separate tensors
    (model_trainer.py:57), step by step; odd sizes exercise the non-float4 tail.  fp32 tolerance 1e-6 relative to the
    update size (same rule, different association of the bias corrections)."""
    from fissure_segmentation_amd.optim import FlatAdam
    torch.manual_seed(0)
    shapes = [(7, 3), (5,), (33, 17), (1,), (64, 64, 1)]   # 4691 elements: 4691 % 4 == 3
    a = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    oa = torch.optim.Adam(a, lr=1e-2, weight_decay=wd)
    ob = FlatAdam(b, lr=1e-2, weight_decay=wd)
    for step in range(12):
        gs = [torch.randn_like(p) * (0.1 + step) for p in a]
        for p, q, g in zip(a, b, gs):
            p.grad, q.grad = g.clone(), g.clone()
        oa.step(), ob.step()
        for p, q in zip(a, b):
            torch.testing.assert_close(q, p, rtol=1e-5, atol=2e-7)
    assert float(ob.state_dict()["state"][0]["step"]) == 12.0
    # state_dict round trip into a fresh optimizer continues identically
    c = [torch.nn.Parameter(q.detach().clone()) for q in b]
    oc = FlatAdam(c, lr=1e-2, weight_decay=wd)
    oc.load_state_dict(ob.state_dict())
    gs = [torch.randn_like(p) for p in a]
    for q, r, g in zip(b, c, gs):
        q.grad, r.grad = g.clone(), g.clone()
    ob.step(), oc.step()
    for q, r in zip(b, c):
        assert torch.equal(q, r)


@pytest.mark.gpu
def test_flat_adam_graph_replay_and_device_lr():
    """the update replayed inside a hipGraph advances its own device step count, and a tensor learning rate is re-read at
    every replay (a scheduler can change it without re-capturing)"""
    from fissure_segmentation_amd.optim import FlatAdam
    torch.manual_seed(1)
    w0 = torch.randn(5000, device="cuda")
    ref_p = torch.nn.Parameter(w0.clone())
    ref = torch.optim.Adam([ref_p], lr=1e-2)
    p = torch.nn.Parameter(w0.clone())
    lr = torch.tensor([1e-2], device="cuda")
    opt = FlatAdam([p], lr=lr)
    g = torch.randn(5000, device="cuda")
    p.grad = g.clone()
    opt.gather_grads()
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(graph):
            opt.step_flat()
    torch.cuda.current_stream().wait_stream(s)
    for step in range(6):
        if step == 3:
            lr.fill_(5e-3)
            ref.param_groups[0]["lr"] = 5e-3
        ref_p.grad = g.clone()
        ref.step()
        graph.replay()
        torch.testing.assert_close(p, ref_p, rtol=1e-5, atol=2e-7)
    assert float(opt.state_dict()["state"][0]["step"]) == 6.0


@pytest.mark.gpu
def test_flat_adam_error_paths():
    from fissure_segmentation_amd import _lib
    import ctypes
    t = torch.zeros(64, device="cuda")
    st = torch.zeros(2, device="cuda")
    P = lambda x: ctypes.c_void_p(x.data_ptr())  # noqa: E731
    with pytest.raises(RuntimeError, match="NULL"):
        _lib.call("fsg_adam_flat_f32", P(t), None, P(t), P(t), P(st), 64, 1e-3, None, 0.9, 0.999, 1e-8, 0.0, None)
    with pytest.raises(RuntimeError, match="aligned"):
        _lib.call("fsg_adam_flat_f32", ctypes.c_void_p(t.data_ptr() + 4), P(t), P(t), P(t), P(st), 32, 1e-3, None, 0.9,
                  0.999, 1e-8, 0.0, None)
    with pytest.raises(RuntimeError, match="hyper"):
        _lib.call("fsg_adam_flat_f32", P(t), P(t), P(t), P(t), P(st), 64, 1e-3, None, 1.0, 0.999, 1e-8, 0.0, None)


@pytest.mark.gpu
def test_chamfer_backward_collapsed_targets(fsg, device):
    """Chamfer backward when thousands of points share a handful of nearest neighbours (a collapsed reconstruction): the
    default path sums the lanes of a wave that hit the same target before its atomics -- same gradient as fp64 autograd.
    4096 points against 12 targets (and the 12 against the 4096), rtol 1e-4."""
    from fissure_segmentation_amd.losses.chamfer_loss import ChamferLoss
    rng = np.random.default_rng(21)
    a = rng.uniform(-1, 1, (2, 4096, 3)).astype(np.float32)
    b = rng.uniform(-1, 1, (2, 12, 3)).astype(np.float32)
    at, bt = G(a, device).requires_grad_(True), G(b, device).requires_grad_(True)
    ChamferLoss()(at, bt).backward()
    ar, br = torch.from_numpy(a).double().requires_grad_(True), torch.from_numpy(b).double().requires_grad_(True)
    d = ((ar[:, :, None, :] - br[:, None, :, :]) ** 2).sum(-1)
    (d.min(2).values.mean(1).mean() + d.min(1).values.mean(1).mean()).backward()
    np.testing.assert_allclose(N(at.grad), ar.grad.numpy(), rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(N(bt.grad), br.grad.numpy(), rtol=1e-4, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("cp,relu", [(2, True), (3, True), (3, False)])
def test_fold_layer1_matches_the_unfused_composition(fsg, device, cp, relu):
    """functional.fold_layer1 (fsg_fold_layer1_f32: first layer of a folding MLP, folding_net.py:205-221) against
    [relu](Conv1d over cat([code repeated, pts])) written with torch ops, values and all three gradients; the weight
    operand is a column slice of the conv weight (row stride E + cp).  fp32, rtol 1e-4."""
    torch.manual_seed(cp)
    B, m, E, Co = 3, 301, 40, 64
    conv = torch.nn.Conv1d(E + cp, Co, 1).to(device)
    code = torch.randn(B, E, device=device, requires_grad=True)
    pts = torch.randn(B, m, cp, device=device, requires_grad=True)
    g = torch.randn(B, m, Co, device=device)
    w = conv.weight.view(Co, E + cp)
    ref = conv(torch.cat([code[:, :, None].expand(B, E, m), pts.transpose(1, 2)], dim=1)).transpose(1, 2)
    ref = torch.relu(ref) if relu else ref
    ref.backward(g)
    want = [t.grad.clone() for t in (code, pts, conv.weight, conv.bias)]
    for t in (code, pts, conv.weight, conv.bias):
        t.grad = None
    w_code, w_pts = fsg.functional.split_cols(w, E)
    per_cloud = torch.nn.functional.linear(code, w_code, conv.bias)
    out = fsg.functional.fold_layer1(pts, w_pts, per_cloud, relu=relu)
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-5)
    out.backward(g)
    for got, exp in zip((code.grad, pts.grad, conv.weight.grad, conv.bias.grad), want):
        torch.testing.assert_close(got, exp, rtol=1e-4, atol=1e-4)
    with pytest.raises(RuntimeError, match="cp"):
        fsg.functional.fold_layer1(torch.randn(B, m, 4, device=device), torch.randn(Co, 4, device=device), per_cloud.detach())


@pytest.mark.gpu
def test_linear_pm_relu_matches_linear_then_relu(fsg, device):
    """functional.linear_pm_relu (bias + ReLU in the GEMM epilogue) == relu(linear): values bit for bit, gradients 1e-5"""
    torch.manual_seed(0)
    lin = torch.nn.Linear(96, 160).to(device)
    x = torch.randn(4, 500, 96, device=device, requires_grad=True)
    g = torch.randn(4, 500, 160, device=device)
    ref = torch.relu(lin(x))
    ref.backward(g)
    want = [t.grad.clone() for t in (x, lin.weight, lin.bias)]
    for t in (x, lin.weight, lin.bias):
        t.grad = None
    out = fsg.functional.linear_pm_relu(x, lin.weight, lin.bias)
    assert torch.equal(out, ref)
    out.backward(g)
    for got, exp in zip((x.grad, lin.weight.grad, lin.bias.grad), want):
        torch.testing.assert_close(got, exp, rtol=1e-5, atol=1e-4)


# ---------------------------------------------------------------------------------------------------------------------
# PointTransformer path against fixtures produced by the REFERENCE's own seg_model.py / pointops.py
# (oracle/make_golden_pt.py: the two pointops_cuda calls served by the C oracle, everything else is reference code).

def _check_packed_grads(mod, g, rtol, floor_rel):
    """parameter gradients vs the fixture: full tensors where stored, (norm, first 16) otherwise.  Parameters in front
    of a train-mode BatchNorm / the softmax have a mathematically zero gradient (noise on both sides): `floor_rel` of the
    largest gradient norm is the absolute floor."""
    scale = max(float(g["gnorm_" + n]) for n, _ in mod.named_parameters())
    for n, p in mod.named_parameters():
        got = N(p.grad).reshape(-1).astype(np.float64)
        ref_norm = float(g["gnorm_" + n])
        if "grad_" + n in g.files:
            assert np.linalg.norm(got - g["grad_" + n].reshape(-1)) <= rtol * ref_norm + floor_rel * scale, n
        else:
            assert abs(np.linalg.norm(got) - ref_norm) <= rtol * ref_norm + floor_rel * scale, n
            assert np.linalg.norm(got[:16] - g["ghead_" + n]) <= 4 * rtol * np.linalg.norm(g["ghead_" + n]) + \
                rtol * ref_norm / np.sqrt(got.size) * 4 + floor_rel * scale, n
    for n, b in mod.named_buffers():
        if "running" in n:
            if "buf_" + n in g.files:
                np.testing.assert_allclose(N(b), g["buf_" + n], rtol=1e-4, atol=1e-5, err_msg=n)
            else:
                assert abs(float(b.double().norm()) - float(g["bnorm_" + n])) <= 1e-4 * float(g["bnorm_" + n]) + 1e-6, n


@pytest.mark.parametrize("name", ["pt_layer_c32", "pt_layer_c64", "pt_layer_c128", "pt_layer_c256", "pt_layer_c512",
                                  "pt_layer_c64_eval"])
def test_pt_layer_vs_reference_golden(fsg, device, name):
    """fused PointTransformerLayer (fsg_pt_attn_*) against seg_model.py:17-53 run by the reference itself: output 1e-4,
    gradients of features / coordinates / all 14 parameters 1e-3 in norm, running statistics 1e-4."""
    from fissure_segmentation_amd.models.pointtransformer.seg_model import PointTransformerLayer
    g = load(name)
    c, ns, sizes, train = int(g["c"]), int(g["ns"]), tuple(int(s) for s in g["sizes"]), bool(g["train"])
    lay = fill_state_dict(PointTransformerLayer(c, c, 8, ns), 811 + c).to(device).train(train)
    xyz, feat, off = packed(900 + c + ns, sizes, c)
    p, x = G(xyz, device).requires_grad_(True), G(feat, device).requires_grad_(True)
    y = lay([p, x, G(off, device)])
    y.backward(G(np.random.default_rng(5).standard_normal(tuple(y.shape)).astype(np.float32), device))
    assert np.abs(N(y) - g["y"]).max() <= 1e-4 * max(1.0, np.abs(g["y"]).max())
    assert np.linalg.norm(N(x.grad) - g["grad_x"]) <= 1e-3 * np.linalg.norm(g["grad_x"])
    assert np.linalg.norm(N(p.grad) - g["grad_p"]) <= 1e-3 * np.linalg.norm(g["grad_p"])
    _check_packed_grads(lay, g, 1e-3, 1e-5)


def test_pt_block_and_transitions_vs_reference_golden(fsg, device):
    """PointTransformerBlock (:121-142), TransitionDown stride 1 / 4 (:56-84), TransitionUp head / two-level (:87-118),
    interpolation and queryandgroup (pointops.py:198-215, 100-123) against the reference's own outputs."""
    from fissure_segmentation_amd.models.pointtransformer import pointops, seg_model as sm

    def rel(a, ref):
        return np.linalg.norm(N(a) - ref) / np.linalg.norm(ref)

    g = load("pt_block_c64")
    blk = fill_state_dict(sm.PointTransformerBlock(64, 64, 8, 16), 821).to(device).train()
    xyz, feat, off = packed(1821, (90, 11, 60), 64)
    x = G(feat, device).requires_grad_(True)
    _, y, _ = blk([G(xyz, device), x, G(off, device)])
    y.backward(G(np.random.default_rng(6).standard_normal(tuple(y.shape)).astype(np.float32), device))
    np.testing.assert_allclose(N(y), g["y"], **TOL)
    assert rel(x.grad, g["grad_x"]) <= 1e-3
    _check_packed_grads(blk, g, 1e-3, 1e-5)

    for name in ("pt_td_s1", "pt_td_s4"):
        g = load(name)
        seed, sizes = int(g["seed"]), tuple(int(s) for s in g["sizes"])
        td = fill_state_dict(sm.TransitionDown(int(g["cin"]), int(g["cout"]), int(g["stride"]), int(g["ns"])), seed)
        td = td.to(device).train()
        xyz, feat, off = packed(seed + 1000, sizes, int(g["cin"]))
        x = G(feat, device).requires_grad_(True)
        n_p, y, n_o = td([G(xyz, device), x, G(off, device)])
        y.backward(G(np.random.default_rng(seed + 2000).standard_normal(tuple(y.shape)).astype(np.float32), device))
        assert np.array_equal(N(n_o), g["new_o"]) and n_o.dtype == torch.int32
        assert np.array_equal(N(n_p), g["new_p"])          # FPS picked the same rows
        np.testing.assert_allclose(N(y), g["y"], **TOL)
        assert rel(x.grad, g["grad_x"]) <= 1e-3, name
        _check_packed_grads(td, g, 1e-3, 1e-5)

    g = load("pt_tu_head")
    tu = fill_state_dict(sm.TransitionUp(64, None), 841).to(device).train()
    xyz, feat, off = packed(1841, (8, 8, 5), 64)
    x = G(feat, device).requires_grad_(True)
    y = tu([G(xyz, device), x, G(off, device)])
    y.backward(G(np.random.default_rng(2841).standard_normal(tuple(y.shape)).astype(np.float32), device))
    np.testing.assert_allclose(N(y), g["y"], **TOL)
    assert rel(x.grad, g["grad_x"]) <= 1e-3
    _check_packed_grads(tu, g, 1e-3, 1e-5)

    g = load("pt_tu")
    tu = fill_state_dict(sm.TransitionUp(64, 32), 842).to(device).train()
    xyz1, feat1, off1 = packed(1842, (120, 33, 64), 32)
    xyz2, feat2, off2 = packed(1843, (30, 2, 16), 64)    # a coarse segment with fewer than 3 points: padding rule
    x1, x2 = G(feat1, device).requires_grad_(True), G(feat2, device).requires_grad_(True)
    y = tu([G(xyz1, device), x1, G(off1, device)], [G(xyz2, device), x2, G(off2, device)])
    y.backward(G(np.random.default_rng(2842).standard_normal(tuple(y.shape)).astype(np.float32), device))
    np.testing.assert_allclose(N(y), g["y"], **TOL)
    assert rel(x1.grad, g["grad_x1"]) <= 1e-3 and rel(x2.grad, g["grad_x2"]) <= 1e-3
    _check_packed_grads(tu, g, 1e-3, 1e-5)

    g = load("pt_interp")
    f2 = G(feat2, device).requires_grad_(True)
    y = pointops.interpolation(G(xyz2, device), G(xyz1, device), f2, G(off2, device), G(off1, device))
    y.backward(G(np.random.default_rng(2850).standard_normal(tuple(y.shape)).astype(np.float32), device))
    np.testing.assert_allclose(N(y), g["y"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(N(f2.grad), g["grad_feat"], rtol=1e-4, atol=1e-5)

    g = load("pt_group")
    q1 = pointops.queryandgroup(8, G(xyz1, device), G(xyz1, device), G(feat1, device), None, G(off1, device),
                                G(off1, device), use_xyz=True)
    q2 = pointops.queryandgroup(4, G(xyz1, device), G(xyz2, device), G(feat1, device), None, G(off1, device),
                                G(off2, device), use_xyz=False)
    assert np.array_equal(N(q1), g["self_xyz"]) and np.array_equal(N(q2), g["cross"])   # pure gather / subtract: exact


def test_pointtransformer_model_vs_reference_golden(fsg, device):
    """PointTransformerCompatibility(6,4), 2 x 2048 points (BASELINE config 3 cloud size), train mode, against the
    reference's own forward/backward.  Tolerance: the net is ~60 train-mode BatchNorms deep and its level 5 normalises
    over 16 rows; re-ordering fp32 sums alone moves isolated logits by 1.3e-4 (measured between the reference and a
    restatement that only differed in `.contiguous()` calls), so the whole-model bar is 5e-4 on logits, 1e-2 in norm on
    gradients, with the layer-level fixtures above holding the 1e-4 bar."""
    from fissure_segmentation_amd.models.pointtransformer.seg_model import PointTransformerCompatibility
    g = load("pt_compat_c6")
    net = fill_state_dict(PointTransformerCompatibility(6, 4), 801)
    assert [str(s) for s in g["keys"]] == list(net.state_dict().keys())
    net = net.to(device).train()
    y, gx = run_model(net, cloud(1801, 2, 6, 2048), 2801, device)
    np.testing.assert_allclose(N(y), g["logits"], rtol=5e-4, atol=5e-4)
    assert np.linalg.norm(N(gx) - g["grad_x"]) <= 1e-2 * np.linalg.norm(g["grad_x"])
    _check_packed_grads(net, g, 1e-2, 1e-3)

    g = load("pt_compat_c3_eval")
    net = fill_state_dict(PointTransformerCompatibility(3, 4), 802).to(device).eval()
    with torch.no_grad():
        y = net(G(cloud(1802, 2, 3, 1024), device))
    np.testing.assert_allclose(N(y), g["logits"], rtol=2e-4, atol=2e-4)


class _ReplayRandperm:
    def __init__(self, g, device):
        self.rows = [g[f"perm{i}"] for i in range(int(g["n_perm"]))]
        self.i, self.device = 0, device

    def __call__(self, n, *a, **kw):
        r = self.rows[self.i]
        assert len(r) == n, "predict_full_pointcloud drew a different sequence of permutations than the reference"
        self.i += 1
        return torch.from_numpy(r.astype(np.int64)).to(self.device)


@pytest.mark.parametrize("batched", [True, False])
def test_predict_full_pointcloud_vs_reference_golden(fsg, device, monkeypatch, batched):
    """models/point_seg_net.py:21-48 as run by the reference on its own DGCNNSeg (eval mode): the recorded randperm rows
    are replayed, so the class probabilities must agree -- batched HIP form and sequential form."""
    from fissure_segmentation_amd.models.dgcnn import DGCNNSeg
    g = load("predict_full_s851")
    net = fill_state_dict(DGCNNSeg(k=8, in_features=3, num_classes=4), 851).to(device).eval()
    replay = _ReplayRandperm(g, device)
    monkeypatch.setattr(torch, "randperm", replay)
    if not batched:
        net._ensemble_batchable = lambda _pc: False
    with torch.no_grad(), pytest.warns(UserWarning):
        out = net.predict_full_pointcloud(G(cloud(1851, 1, 3, 1500), device), sample_points=256, n_runs_min=10)
    assert replay.i == len(replay.rows)
    np.testing.assert_allclose(N(out), g["probs"], rtol=1e-4, atol=1e-5)


def test_farthest_point_sampling_vs_reference_golden(fsg, device):
    """dseg_ae_regularization.py:30-43 as run by the reference (its random start is ind[0] of the fixture)"""
    from fissure_segmentation_amd.utils.general_utils import farthest_point_sampling
    g = load("fps_torch")
    for i in range(int(g["n_cases"])):
        seed, n, m = (int(v) for v in g[f"case{i}"])
        pts = np.random.default_rng(seed).uniform(-1, 1, (1, n, 3)).astype(np.float32)
        sub, ind = farthest_point_sampling(G(pts, device), m, start=int(g[f"ind{i}"][0]))
        assert np.array_equal(N(ind), g[f"ind{i}"]), i
        assert np.array_equal(N(sub), g[f"pts{i}"]), i


# ---------------------------------------------------------------------------------------------------------------------
# Model-level parity at the BASELINE shapes against the CPU oracle (oracle/ref_cpu.py, pinned to the reference by
# tests/test_oracle_golden.py).  The oracle materialises (B,N,N) and (B,2C,N,k) like the reference does.

class GraphTape:
    """A dynamic-graph net is a discontinuous function of its input: a k-th neighbour whose distance ties with the
    (k+1)-th within fp32 rounding of the FEATURES flips between two correct implementations, and the flipped point's
    max-pooled features then differ by O(0.1).  Model-level parity is therefore stated in two halves:
      (1) every graph the HIP net builds equals, bit for bit, the C oracle's kNN of the very tensor the kernel was given;
      (2) with those graphs handed to the oracle model (instead of the ones it would build from its own, 1e-6-different
          features) logits / gradients / running statistics agree to the floating-point bar.
    The fraction of rows where the oracle's own graph differs from the replayed one is measured and bounded."""

    def __init__(self, fsg, monkeypatch):
        self.calls, self.fsg, self.mp = [], fsg, monkeypatch
        real = fsg.functional.knn_graph

        def recording(x, k, c_knn=None, fix_diag=True, drop_first=False, return_dist=False, **kw):
            out = real(x, k, c_knn=c_knn, fix_diag=fix_diag, drop_first=drop_first, return_dist=return_dist, **kw)
            idx = out[0] if (return_dist or kw.get("pq_weight") is not None) else out
            self.calls.append(dict(x=N(x), k=k, c_knn=c_knn, fix_diag=fix_diag, drop_first=drop_first, idx=N(idx)))
            return out
        monkeypatch.setattr(fsg.functional, "knn_graph", recording)

    def check_exact_and_replay(self, max_flipped_rows=2e-3):
        """half (1), then patch the oracle's graph builders to replay the tape in call order"""
        for c in self.calls:
            want, _ = c_api.knn_dense(c["x"], c["k"], c_knn=c["c_knn"], fix_diag=c["fix_diag"], drop_first=c["drop_first"])
            assert np.array_equal(c["idx"], want), "graph build differs from the C oracle on the kernel's own input"
        self.flipped, self.pos = [], 0

        def replay(own):
            c = self.calls[self.pos % len(self.calls)]
            first_pass = len(self.flipped) < len(self.calls)
            self.pos += 1
            if first_pass:      # the oracle's OWN graph (a brute-force C build: 7 s at 8192 points) is only needed for this statistic,
                own_idx = own()         # once per graph -- the fp64 and the flipped oracle runs replay without building theirs
                assert c["idx"].shape == tuple(own_idx.shape)
                rows = (np.sort(c["idx"], -1) != np.sort(own_idx.numpy(), -1)).any(-1).mean()
                self.flipped.append(float(rows))
                assert rows <= max_flipped_rows, f"{rows:.2%} of the rows have another neighbour set than the oracle's own graph"
            return torch.from_numpy(c["idx"].astype(np.int64))

        self.mp.setattr(ref_cpu, "knn", lambda x, k, self_loop=False, return_dist=False:
                        replay(lambda: ref_cpu._knn_c(x, k, drop_first=not self_loop)))
        self.mp.setattr(ref_cpu, "knn_opensrc", lambda x, k: replay(lambda: ref_cpu._knn_c(x, k, fix_diag=False)))


def _rel(a, b, mask=None, den=None):
    """||a - b|| / ||b|| over the entries where `mask` (entries to leave out) is False"""
    a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
    d = np.abs(a - b)
    if mask is not None:
        d = d[~np.asarray(mask).reshape(-1)]
    return float(np.linalg.norm(d) / max(np.linalg.norm(b) if den is None else den, 1e-300))


class _FlipAct(torch.nn.Module):
    """LeakyReLU / ReLU whose derivative is FLIPPED on the elements within eps of the kink (eps per call, in call order)"""

    def __init__(self, slope, eps_list, counter):
        super().__init__()
        self.slope, self.eps_list, self.call, self.counter = slope, eps_list, 0, counter

    def forward(self, u):
        eps = self.eps_list[self.call % len(self.eps_list)]
        self.call += 1
        near = u.abs() < eps
        self.counter[0] += int(near.sum())
        self.counter[1] += u.numel()
        plain = torch.where(u > 0, u, u * self.slope)
        flipped = torch.where(u > 0, u * self.slope, u)
        return torch.where(near, flipped, plain)


class FlipOracle:
    """The flip noise of a net, computed from the oracle instead of trimmed away.

    The backward pass is discontinuous in the forward values where a pre-activation sits within fp32 noise of the
    LeakyReLU / ReLU kink: two correct fp32 evaluations take different slopes there, and ONE such element moves an entry of the
    BatchNorm-bias gradient in front of it by ~0.8 |g| of one row.  Whether an fp32 run flips a given element is a coin toss, so
    the oracle's own fp32-vs-fp64 error is a poor yardstick (it is ~1e-6 when it happens not to flip and ~1e-3 when it does).
    Instead: (1) hooks on every activation of the fp32 and the fp64 oracle runs measure, per call, the fp32 noise NEAR THE KINK,
    delta0 = max |u32 - u64| over |u64| < 1e-2, and open the flip zone eps = 4 delta0; (2) a third oracle run in fp64 takes the
    OTHER slope on exactly the elements inside their zone; (3) e_flip = |g_flipped - g64| / |g64| per gradient tensor is what
    flips can do to it.  The HIP gradient is then held to  e_hip <= max(floor, 10 e_cpu32, 2 e_flip)  on ALL entries: nothing is
    trimmed or masked.  The number of elements inside a flip zone is printed and bounded (< 0.1 % of the activations).
    (Max-pool near-ties re-route one of N*k contributions of a weight-gradient row -- 1e-4 of its norm, below every bar here.)"""

    ACTS = (torch.nn.LeakyReLU, torch.nn.ReLU)

    def __init__(self, ref32, ref64):
        self.u32, self.eps, self.ref64 = {}, {}, ref64
        self.names = [n for n, m in ref64.named_modules() if isinstance(m, self.ACTS)]
        mods32, mods64 = dict(ref32.named_modules()), dict(ref64.named_modules())
        self.handles = []
        for n in self.names:
            self.handles.append(mods32[n].register_forward_pre_hook(
                lambda mod, inp, key=n: self.u32.setdefault(key, []).append(inp[0].detach().float().clone())))
            self.handles.append(mods64[n].register_forward_pre_hook(lambda mod, inp, key=n: self._measure(key, inp[0].detach())))

    def _measure(self, key, u64):
        calls = self.eps.setdefault(key, [])
        u32 = self.u32[key][len(calls)]
        band = u64.abs() < 1e-2
        d0 = float((u32.double() - u64)[band].abs().max()) if bool(band.any()) else 0.0
        calls.append(4.0 * d0 + 1e-300)
        self.u32[key][len(calls) - 1] = None          # the fp32 copy is no longer needed

    def flipped_model(self):
        import copy
        for h in self.handles:
            h.remove()
        mod = copy.deepcopy(self.ref64)
        self.counter = [0, 0]
        for n in self.names:
            parent = mod
            *path, leaf = n.split(".")
            for p_ in path:
                parent = getattr(parent, p_)
            old = getattr(parent, leaf)
            slope = old.negative_slope if isinstance(old, torch.nn.LeakyReLU) else 0.0
            setattr(parent, leaf, _FlipAct(slope, self.eps[n], self.counter))
        return mod


def _model_vs_oracle(net, ref, x, gseed, device, out_tol, g_floor, tape=None, loss_fn=None, ref_loss_fn=None):
    """Outputs: HIP vs the fp32 oracle at `out_tol` (the north_star bar).
    Gradients: LeakyReLU / ReLU kinks make the backward pass discontinuous in the forward values, so two fp32 evaluations of the
    SAME algorithm differ by more than rounding.  The bar is computed from the oracle (FlipOracle above): the HIP gradient must
    be as close to the fp64 one as  max(g_floor, 10 x the oracle's own fp32 error, 2 x the change the oracle's gradient suffers
    when every pre-activation within fp32 noise of a kink takes the other slope)  -- on all entries, nothing trimmed; and where
    that flip change is SMALLER than the oracle's own fp32 error, as close as max(g_floor, 3 x that error) (the ratchet)."""
    import copy
    xt = G(x, device).requires_grad_(True)
    y = net(xt)
    gr = np.random.default_rng(gseed).standard_normal(tuple(y.shape)).astype(np.float32)
    if loss_fn is None:
        y.backward(G(gr, device))
    else:
        loss = loss_fn(y, xt)
        loss.backward()
    if tape is not None:
        tape.check_exact_and_replay()
    runs = {}
    ref64 = copy.deepcopy(ref).double()
    flips = FlipOracle(ref, ref64)

    def run(mod):
        if tape is not None:
            tape.pos = 0                              # every oracle run replays the HIP graphs from the first one
        xr = torch.from_numpy(x).to(next(mod.parameters()).dtype).requires_grad_(True)
        yr = mod(xr)
        if loss_fn is None:
            yr.backward(torch.from_numpy(gr).to(yr.dtype))
            lr = None
        else:
            lr = ref_loss_fn(yr, xr)
            lr.backward()
        return yr.detach().numpy(), xr.grad.numpy(), {n: p.grad.numpy() for n, p in mod.named_parameters()}, lr
    runs["f32"] = run(ref)
    runs["f64"] = run(ref64)
    runs["flip"] = run(flips.flipped_model())
    y32, gx32, gp32, l32 = runs["f32"]
    y64, gx64, gp64, l64 = runs["f64"]
    _, gxf, gpf, _ = runs["flip"]
    np.testing.assert_allclose(N(y), y32, rtol=out_tol, atol=out_tol)
    if loss_fn is not None:
        assert abs(float(loss.detach()) - float(l32)) <= 1e-4 * abs(float(l32))
    report = {"out_max_abs": float(np.abs(N(y) - y32).max())}
    near, total = flips.counter
    report["flip_zone"] = {"activations_inside": near, "of": total, "fraction": near / max(total, 1)}
    assert near <= 1e-3 * total, report["flip_zone"]
    # RATCHET (round 4): the flip bar is only available where flips CAN explain more than the oracle's own fp32 error.  A tensor
    # whose flip bound is below the oracle's fp32 error is held to 3 x that error (a kernel regression to a few 1e-2 in one
    # weight gradient no longer hides behind a 2 e_flip that belongs to other tensors' kinks); the report counts the tensors
    # that actually needed the flip bar.
    def passes(e_hip, e_cpu, e_flip):
        if e_flip < e_cpu:
            return e_hip <= max(g_floor, 3 * e_cpu), False
        return e_hip <= max(g_floor, 10 * e_cpu, 2 * e_flip), e_hip > max(g_floor, 10 * e_cpu)
    needed_flip = []
    if xt.grad is not None:
        e_hip, e_cpu, e_flip = _rel(N(xt.grad), gx64), _rel(gx32, gx64), _rel(gxf, gx64)
        report["grad_x"] = (e_hip, e_cpu, e_flip)
        ok, used = passes(e_hip, e_cpu, e_flip)
        assert ok, ("grad_x", e_hip, e_cpu, e_flip)
        if used:
            needed_flip.append("grad_x")
    scale = max(float(np.linalg.norm(v)) for v in gp64.values())
    worst, bad = ("", 0.0, 0.0, 0.0), []
    for n, p in net.named_parameters():
        got = N(p.grad).astype(np.float64)
        den = max(float(np.linalg.norm(gp64[n])), 1e-3 * scale)   # mathematically-zero gradients: noise on all sides
        e_hip, e_cpu, e_flip = _rel(got, gp64[n], None, den), _rel(gp32[n], gp64[n], None, den), _rel(gpf[n], gp64[n], None, den)
        if e_hip > worst[1]:
            worst = (n, e_hip, e_cpu, e_flip)
        ok, used = passes(e_hip, e_cpu, e_flip)
        if not ok:
            bad.append((n, e_hip, e_cpu, e_flip))
        if used:
            needed_flip.append(n)
    report["worst_param (e_hip, e_cpu32, e_flip)"] = worst
    report["tensors_that_needed_the_flip_bar"] = (len(needed_flip), needed_flip[:6])
    if tape is not None:
        report["rows_with_other_neighbours_in_own_graph"] = tape.flipped
    print("\nPARITY", type(net).__name__, tuple(x.shape), report)
    assert not bad, bad
    for n, b in net.named_buffers():
        if "running" in n:
            np.testing.assert_allclose(N(b), dict(ref.named_buffers())[n].numpy(), rtol=1e-4, atol=1e-5, err_msg=n)
    return report


@pytest.mark.parametrize("B,Np,k,dynamic", [(8, 2048, 20, True), (1, 8192, 40, True), (2, 2048, 40, False)])
def test_dgcnnseg_full_size_vs_oracle(fsg, device, monkeypatch, B, Np, k, dynamic):
    """DGCNNSeg at BASELINE config 2 (8 x 2048, k=20), at the config-4 cloud size (8192 points, k=40; one cloud keeps
    the oracle's 8192^2 matrices small) and with the static k=40 graph of the reference's experiment scripts
    (bash_scripts/run_dgcnn_seg_experiments.sh:17): logits 1e-4, gradients in norm, running statistics."""
    from fissure_segmentation_amd.models.dgcnn import DGCNNSeg
    ref = fill_state_dict(ref_cpu.DGCNNSeg(k=k, in_features=3, num_classes=4, dynamic=dynamic), 7).train()
    net = DGCNNSeg(k=k, in_features=3, num_classes=4, dynamic=dynamic)
    net.load_state_dict(ref.state_dict())
    _model_vs_oracle(net.to(device).train(), ref, cloud(4000 + Np + k, B, 3, Np), 4001, device, 1e-4, 1e-3,
                     tape=GraphTape(fsg, monkeypatch))


def test_knn_dense_8192_64ch_k40_vs_c_oracle(fsg, device):
    """the config-4 feature-space graph build (64 channels, 8192 points, k=40): indices AND distance bits"""
    x = cloud(4100, 1, 64, 8192)
    idx, dist = fsg.functional.knn_graph(G(x, device), 40, return_dist=True)
    idx_o, dist_o = c_api.knn_dense(x, 40, fix_diag=True)
    assert np.array_equal(N(idx), idx_o)
    assert np.array_equal(N(dist).view(np.int32), dist_o.view(np.int32))


def test_folding_ae_chamfer_config5_shape_vs_oracle(fsg, device, monkeypatch):
    """BASELINE config 5 geometry: DGCNNFoldingNet(n_input_points=4096, decode_mesh=True) -- the sqrt(n) x sqrt(n) grid
    of shapes/shape_constructor.py:8-23 -- + ChamferLoss(recon, input), forward and backward, 2 clouds of 4096 points
    (the oracle's per-layer edge tensors are 160 MB per cloud)."""
    from fissure_segmentation_amd.losses.chamfer_loss import ChamferLoss
    from fissure_segmentation_amd.models.folding_net import DGCNNFoldingNet
    ref = fill_state_dict(ref_cpu.DGCNNFoldingNet(k=20, n_embedding=512, n_input_points=4096, decode_mesh=True), 9).train()
    net = DGCNNFoldingNet(k=20, n_embedding=512, shape_type="plane", n_input_points=4096, decode_mesh=True)
    net.load_state_dict(ref.state_dict())
    net = net.to(device).train()
    rep = _model_vs_oracle(net, ref, cloud(4200, 2, 3, 4096), 0, device, 1e-4, 1e-3, tape=GraphTape(fsg, monkeypatch),
                           loss_fn=lambda y, x: ChamferLoss()(y, x.detach()),
                           ref_loss_fn=lambda y, x: ref_cpu.ChamferLoss()(y, x.detach()))
    assert rep["out_max_abs"] <= 1e-4


def test_pointtransformer_config3_shape_vs_oracle(fsg, device):
    """BASELINE config 3: PointTransformerCompatibility on 8 clouds of 2048 points, forward + backward, against the
    oracle (bit-identical to the reference on the committed whole-model fixture).  Whole-model tolerance as in
    test_pointtransformer_model_vs_reference_golden."""
    from fissure_segmentation_amd.models.pointtransformer.seg_model import PointTransformerCompatibility
    ref = fill_state_dict(ref_cpu.PointTransformerCompatibility(6, 4), 803).train()
    net = PointTransformerCompatibility(6, 4)
    net.load_state_dict(ref.state_dict())
    _model_vs_oracle(net.to(device).train(), ref, cloud(4300, 8, 6, 2048), 4301, device, 5e-4, 1e-3)


# ---------------------------------------------------------------------------------------------------------------------
# The reference's mixed-precision step (model_trainer.py:75-76,154-195): autocast -> model -> loss -> scaled backward ->
# scaler.step(optimizer) -> scaler.update -> zero_grad.

def _forward_step(model, crit, opt, scaler, x, y, amp):
    """the literal sequence of ModelTrainer.forward_step(train=True)"""
    with torch.autocast("cuda", enabled=amp):
        output = model(x)
        loss = crit(output, y)
    if isinstance(loss, tuple):
        loss, components = loss
    scaler.scale(loss).backward()
    scaler.step(opt)
    scaler.update()
    opt.zero_grad()
    return loss.detach(), output.detach()


@pytest.mark.parametrize("model,optimizer", [("dgcnn", "adam"), ("dgcnn", "flat"), ("pointtransformer", "adam"),
                                             ("dgcnn_stn_static", "adam"), ("pointnet", "adam")])
def test_reference_amp_forward_step(fsg, device, model, optimizer):
    """The HIP models inside the reference's autocast + GradScaler step: no dtype error anywhere, outputs fp32, no step
    skipped for an overflow, and -- because every HIP stage leaves the autocast region and the loss scale is a power of
    two -- the loss and the (unscaled) gradients of a step equal those of the plain fp32 step (1e-4 in norm: the
    PointTransformer backward scatters with fp32 atomics, so even two fp32 runs differ in the last bits; parameters after
    several Adam steps are not compared because Adam's g / (|g| + eps) turns that noise into +-lr on zero-gradient
    parameters)."""
    from fissure_segmentation_amd.losses.access_losses import get_loss_fn
    from fissure_segmentation_amd.models.access_models import get_point_seg_model_class
    from fissure_segmentation_amd.optim import FlatAdam

    def build():
        if model == "dgcnn":
            net = get_point_seg_model_class("DGCNN")(in_features=3, num_classes=4, k=8)
        elif model == "dgcnn_stn_static":
            net = get_point_seg_model_class("DGCNN")(in_features=3, num_classes=4, k=8, spatial_transformer=True, dynamic=False)
        elif model == "pointnet":
            net = get_point_seg_model_class("PointNet")(in_features=3, num_classes=4, k=8)
        else:
            net = get_point_seg_model_class("PointTransformer")(in_features=3, num_classes=4)
        net = fill_state_dict(net, 31).to(device).train()
        crit = get_loss_fn("nnunet", torch.tensor([1.0, 2.0, 0.5, 1.5], device=device))
        opt = (FlatAdam(net.parameters(), lr=1e-3) if optimizer == "flat" else torch.optim.Adam(net.parameters(), lr=1e-3))
        return net, crit, opt

    def batch(step):
        return (G(cloud(700 + step, 2, 3, 512), device), G(np.random.default_rng(800 + step).integers(0, 4, (2, 512)), device))

    grads = {}
    for amp in (True, False):          # one step by hand: scaled backward, unscale, compare the gradients
        net, crit, opt = build()
        scaler = torch.amp.GradScaler("cuda", enabled=amp)
        x, y = batch(0)
        with torch.autocast("cuda", enabled=amp):
            out = net(x)
            loss, parts = crit(out, y)
        assert out.dtype == torch.float32 and loss.dtype == torch.float32
        scaler.scale(loss).backward()
        scaler.unscale_(opt)          # walks opt.param_groups: the model's own parameters for FlatAdam too
        grads[amp] = (float(loss), [p.grad.detach().double() for p in net.parameters()])
    assert abs(grads[True][0] - grads[False][0]) <= 1e-6 * abs(grads[False][0])
    ref_scale = max(float(g.norm()) for g in grads[False][1])
    for a, b in zip(grads[True][1], grads[False][1]):
        assert float((a - b).norm()) <= 1e-4 * float(b.norm()) + 1e-6 * ref_scale
    # the literal sequence for three steps
    net, crit, opt = build()
    scaler = torch.amp.GradScaler("cuda")
    for step in range(3):
        loss, out = _forward_step(net, crit, opt, scaler, *batch(step), True)
        assert out.dtype == torch.float32 and bool(torch.isfinite(loss))
    assert scaler.get_scale() == 65536.0                      # no step was skipped for an overflow
    assert all(bool(torch.isfinite(p).all()) for p in net.parameters())


def test_pc_ae_step_and_half_inputs_under_autocast(fsg, device):
    """ModelTrainer disables autocast for ChamferLoss (model_trainer.py:75), but a caller may not: encoder, decoder and
    Chamfer under autocast, and a half-precision input cloud, still run in fp32 and match the plain run."""
    from fissure_segmentation_amd.losses.chamfer_loss import ChamferLoss
    from fissure_segmentation_amd.models.folding_net import DGCNNFoldingNet
    net = fill_state_dict(DGCNNFoldingNet(k=8, n_embedding=64, shape_type="plane", n_input_points=1024, decode_mesh=True), 3)
    net = net.to(device).train()
    x = G(cloud(41, 2, 3, 1024), device).half().float()       # values exactly representable in fp16
    ref = ChamferLoss()(net(x), x)
    ref.backward()
    g_ref = [p.grad.clone() for p in net.parameters()]
    net.zero_grad()
    with torch.autocast("cuda"):
        out = net(x.half())
        loss = ChamferLoss()(out, x.half())
    assert out.dtype == torch.float32 and loss.dtype == torch.float32
    loss.backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * abs(float(ref))
    for p, g in zip(net.parameters(), g_ref):
        assert float((p.grad - g).norm()) <= 1e-4 * float(g.norm()) + 1e-7


def test_flat_adam_is_a_drop_in_for_the_trainer(fsg, device):
    """(1) GradScaler.step(FlatAdam) must unscale / inf-check the real gradients and skip the step on
    overflow; (2) parameters without a gradient stay untouched; (3) state_dict moves to torch.optim.Adam and back."""
    from fissure_segmentation_amd.optim import FlatAdam
    torch.manual_seed(0)

    def make():
        torch.manual_seed(1)
        return torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4)).to(device)
    a, b = make(), make()
    oa, ob = torch.optim.Adam(a.parameters(), lr=1e-2), FlatAdam(b.parameters(), lr=1e-2)
    sa, sb = torch.amp.GradScaler("cuda", init_scale=1024.0), torch.amp.GradScaler("cuda", init_scale=1024.0)
    for step in range(4):
        x = torch.randn(32, 8, device=device)
        for net, opt, sc in ((a, oa, sa), (b, ob, sb)):
            loss = net(x).square().mean()
            sc.scale(loss).backward()
            if step == 2:                                        # an overflow: both must skip the step and halve the scale
                next(net.parameters()).grad[0, 0] = float("inf")
            sc.step(opt)
            sc.update()
            opt.zero_grad()
    assert sa.get_scale() == sb.get_scale() == 512.0
    for p, q in zip(a.parameters(), b.parameters()):
        torch.testing.assert_close(p, q, rtol=1e-5, atol=1e-6)
    # (3) FlatAdam -> torch.optim.Adam -> FlatAdam
    c = make()
    c.load_state_dict(b.state_dict())
    oc = torch.optim.Adam(c.parameters(), lr=1e-2)
    oc.load_state_dict(ob.state_dict())
    d = make()
    d.load_state_dict(b.state_dict())
    od = FlatAdam(d.parameters(), lr=1e-2)
    od.load_state_dict(oc.state_dict())
    x = torch.randn(32, 8, device=device)
    for net, opt in ((b, ob), (c, oc), (d, od)):
        net(x).square().mean().backward()
        opt.step()
    for p, q, r in zip(b.parameters(), c.parameters(), d.parameters()):
        torch.testing.assert_close(p, q, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(p, r, rtol=1e-6, atol=1e-7)
    # (2) a parameter without a gradient keeps its value and its moments
    f = make()
    of = FlatAdam(f.parameters(), lr=1e-2, weight_decay=1e-2)
    for step in range(2):
        x = torch.randn(32, 8, device=device)
        of.zero_grad()
        before = [p.detach().clone() for p in f.parameters()]
        m_before = (of.exp_avg.clone(), of.exp_avg_sq.clone())
        h = f[0](x) if step != 1 else torch.randn(32, 16, device=device)     # step 1: the first layer gets no gradient
        f[2](torch.relu(h)).square().mean().backward()
        of.step()
        if step == 1:
            n0 = f[0].weight.numel() + f[0].bias.numel()
            assert torch.equal(f[0].weight, before[0]) and torch.equal(f[0].bias, before[1])
            assert torch.equal(of.exp_avg[:n0], m_before[0][:n0]) and torch.equal(of.exp_avg_sq[:n0], m_before[1][:n0])
            assert not torch.equal(f[2].weight, before[2])


# ---------------------------------------------------------------------------------------------------------------------
# The bench line of every workload (the driver runs bench.py; a workload whose line breaks is unmeasured).

@pytest.mark.parametrize("workload,dtype,roof", [("c2", "f32", True), ("c3", "bf16", True), ("c3", "f32", True), ("c4", "bf16", True),
                                                 ("c5", "f32", True), ("c2s", "f32", False)])
def test_bench_line_contract(workload, dtype, roof):
    """`python bench.py --workload W` (3 replayed steps, no CPU leg) prints ONE JSON line with the contract's keys, the workload's
    stated arithmetic type as the default dtype, a finite positive rate, whole-job throughput = clouds x points x steps / time,
    and -- for the workloads with a priced dominant kernel -- a roofline object whose fraction lies in (0, 1]."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--workload", workload, "--steps", "3", "--warmup", "1", "--min-seconds", "0",
           "--no-cpu-baseline"]
    if (workload, dtype) == ("c3", "f32"):
        cmd += ["--dtype", "f32"]
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["dtype"] == dtype and d["n_gpus"] == 1 and d["steps"] == 3 and d["unit"] == "points/s" and d["data"] == "synthetic"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    cfg = d["config"]
    assert np.isfinite(d["value"]) and d["value"] > 0
    rate = cfg["clouds_per_gpu"] * cfg["points_per_cloud"] / (d["ms_per_step"] * 1e-3)
    assert abs(rate - d["value"]) <= 2e-2 * d["value"]            # (ms_per_step is rounded to 1 us)
    if roof:
        rl = d["roofline"]
        assert rl is not None and rl["bound"] in ("hbm", "mfma", "l2") and 0.0 < rl["frac"] <= 1.0 and rl["peak"] > 0


# ---------------------------------------------------------------------------------------------------------------------
# N > 1 on the one-GPU box: two ranks share the card (gloo between them) and run the REAL bench.py launch path.

def test_bench_two_rank_rehearsal_averages_the_shard_gradients(fsg, device, tmp_path, monkeypatch):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), both ranks on device 0,
    gloo instead of RCCL: fwd/bwd hipGraph -> in-place all-reduce of FlatAdam's flat gradient buffer -> optimizer
    hipGraph.  After the timed steps bench.py writes rank 0's parameters, every rank's own gradient of one more step and
    their average after the collective (SURVEY 8e parity check).  Three links, each exact or with a computed bar:
      (1) the collective: the average is the mean of the ranks' own gradients;
      (2) the launch path: every rank's gradient out of the replayed graph is what the eager modules compute from the same
          parameters and the same shard (same kernels, deterministic: equal);
      (3) the modules at those TRAINED parameters against the oracle, as test_dgcnnseg_full_size_vs_oracle does at initial ones:
          logits 1e-4, gradients within the oracle's own noise bars (FlipOracle), the oracle replaying the HIP graphs.
    (Until round 3 link (2)+(3) was one comparison with an oracle that builds its own dynamic graphs, bounded at 2e-2 in norm.
    Measured at one set of dumped parameters: two kernel generations that are each within 2.5e-4 of fp64 per layer differ by
    5.9e-2 in the whole-model gradient there, and the oracle differs from them by 2.2e-2 and 5.5e-2 -- a handful of rows that
    pick another k-th neighbour re-route max-pools.  That comparison measured the chaos of the dynamic graph, not the path.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    prefix = str(tmp_path / "ddp")
    r = subprocess.run([os.path.join(root, "tools", "ddp_rehearsal.sh"), prefix, "--steps", "3", "--warmup", "1", "--min-seconds", "0"],
                       cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, open(prefix + ".err").read()[-3000:]
    line = json.loads(open(prefix + ".json").read().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["launch"] == "hipGraph replay"
    assert "in-place all-reduce" in line["config"]["grad_sync"]
    ck = np.load(prefix + "_check.npz")
    B, Np, k, world = int(ck["B"]), int(ck["N"]), int(ck["k"]), int(ck["world"])
    assert world == 2 and str(ck["grad_sync"]) == "flat"
    from bench import synthetic_batch
    from fissure_segmentation_amd.losses.nnu_loss import NNULoss
    from fissure_segmentation_amd.models.dgcnn import DGCNNSeg

    def load(module):
        off = 0
        with torch.no_grad():
            for _, p in module.named_parameters():
                p.copy_(torch.from_numpy(ck["params"][off:off + p.numel()]).view(p.shape))
                off += p.numel()
        assert off == ck["params"].size
        return module
    got = ck["avg_grad"]
    # (1) the collective: to fp32 rounding of one addition and one scaling
    locs = [np.load(prefix + f"_check.npz.rank{r}.npy") for r in range(world)]
    mean_loc = (locs[0].astype(np.float64) + locs[1].astype(np.float64)) / 2
    assert not np.array_equal(locs[0], locs[1])
    assert np.abs(got - mean_loc).max() <= 1e-6 * np.abs(mean_loc).max()
    # (2) the replayed graphs against the eager modules, per rank
    net = load(DGCNNSeg(k=k, in_features=3, num_classes=4)).to(device).train()
    crit = NNULoss(torch.tensor([0.4, 1.2, 1.2, 1.2])).to(device)
    for rank in range(world):
        x, y = synthetic_batch(B, Np, 4, 1234 + rank, device)
        net.zero_grad()
        crit(net(x), y)[0].backward()
        g = N(torch.cat([p.grad.reshape(-1) for p in net.parameters()]))
        err = np.linalg.norm(g - locs[rank]) / np.linalg.norm(locs[rank])
        print(f"\nDDP rehearsal: rank {rank} gradient out of the replayed graph vs eager modules, rel L2 = {err:.3g}")
        assert err <= 1e-6
    # (3) the modules at the dumped parameters against the oracle (rank 0's shard)
    x, y = synthetic_batch(B, Np, 4, 1234, "cpu")
    ref = load(ref_cpu.DGCNNSeg(k=k, in_features=3, num_classes=4)).train()
    net = load(DGCNNSeg(k=k, in_features=3, num_classes=4)).to(device).train()
    w_cls = torch.tensor([0.4, 1.2, 1.2, 1.2])
    y_dev = y.to(device)
    _model_vs_oracle(net, ref, x.numpy(), 0, device, 1e-4, 1e-3, tape=GraphTape(fsg, monkeypatch),
                     loss_fn=lambda out, _x: crit(out, y_dev)[0],
                     ref_loss_fn=lambda out, _x: ref_cpu.nnu_loss(out, y, w_cls.to(out.dtype))[0])


# ---------------------------------------------------------------------------------------------------------------------
# bf16 operand mode (BASELINE configs 3-5): *_bf16 entry points, tolerances stated against the fp32 path / oracle.

def test_edge_gather_bf16(fsg, device):
    """fsg_edge_gather_{fwd,bwd}_bf16: bf16 storage, difference formed in fp32 and rounded once (so it equals torch's
    bf16 arithmetic on the gathered operands exactly); the gradient is accumulated in fp32."""
    x = cloud(55, 2, 16, 300)
    idx = c_api.knn_dense(x, 12)[0]
    xb = G(x, device).bfloat16().requires_grad_(True)
    it = G(idx, device)
    e = fsg.functional.edge_features(xb, it)
    assert e.dtype == torch.bfloat16 and e.shape == (2, 32, 300, 12)
    xf = xb.detach().float()
    nb = torch.gather(xf.unsqueeze(-1).expand(-1, -1, -1, 12), 2, it.long().unsqueeze(1).expand(-1, 16, -1, -1))
    want = torch.cat([(nb - xf.unsqueeze(-1)).bfloat16(), xb.detach().unsqueeze(-1).expand(-1, -1, -1, 12)], 1)
    assert torch.equal(e, want)
    g = torch.randn_like(e)
    e.backward(g)
    xr = xf.clone().requires_grad_(True)
    fsg.functional.edge_features(xr, it).backward(g.float())
    assert xb.grad.dtype == torch.bfloat16
    assert float((xb.grad.float() - xr.grad).abs().max()) <= 1e-2 * float(xr.grad.abs().max())    # one bf16 rounding


def _find_node(fn, name, seen=None):
    seen = set() if seen is None else seen
    if fn is None or fn in seen:
        return None
    seen.add(fn)
    if type(fn).__name__ == name:
        return fn
    for nxt, _ in fn.next_functions:
        r = _find_node(nxt, name, seen)
        if r is not None:
            return r
    return None


@pytest.mark.parametrize("B,C,Np,k,C2", [(2, 3, 300, 20, 64), (1, 15, 130, 40, 128), (1, 3, 8192, 40, 64)])
def test_edgeconv2_bf16_vs_f32(fsg, device, B, C, Np, k, C2):
    """fsg_edgeconv2_{fwd,bwd}_bf16 against the fp32 kernels.  Forward, same inputs: operands carry 8 mantissa bits
    (relative 2^-9 each), products accumulate in fp32 over 64 channels -> outputs within 1e-2 of their scale.  Backward:
    a 1e-2 change of y2 re-routes the max over the k edges in many places, so the two BACKWARD entry points are compared on
    the SAME saved forward state (the fp32 forward's): gradients within 2e-2 in norm."""
    from fissure_segmentation_amd.models.dgcnn import EdgeConv
    F_hip = fsg.functional
    x = cloud(60 + C, B, C, Np)
    ec = fill_state_dict(EdgeConv(C, [64, C2], k, first_layer=True), 61).to(device).train()
    xt = G(x, device).requires_grad_(True)
    y32 = ec(xt)
    with F_hip.mfma_operands("bf16"), torch.no_grad():
        y16 = ec(G(x, device))
    assert not torch.equal(y32, y16)                       # the bf16 entry point really ran
    assert float((y16 - y32).abs().max()) <= 1e-2 * float(y32.abs().max())
    node = _find_node(y32.grad_fn, "_EdgeConv2Backward")
    assert node is not None and node.bf16 is False
    gr = G(np.random.default_rng(62).standard_normal(tuple(y32.shape)).astype(np.float32), device)
    params = [xt] + list(ec.parameters())
    g32 = torch.autograd.grad(y32, params, gr, retain_graph=True)
    node.bf16 = True                                       # same graph, same saved tensors, bf16 backward entry point
    g16 = torch.autograd.grad(y32, params, gr)
    scale = max(float(q.norm()) for q in g32)
    assert any(not torch.equal(a, b) for a, b in zip(g16, g32))
    for a, b in zip(g16, g32):
        assert float((a - b).norm()) <= 2e-2 * float(b.norm()) + 1e-3 * scale


@pytest.mark.parametrize("how", ["switch", "autocast"])
def test_dgcnnseg_bf16_mode_vs_fp32_oracle(fsg, device, monkeypatch, how):
    """DGCNNSeg in bf16 operand mode -- switched on explicitly or by an ambient torch.autocast(bfloat16) -- against the
    fp32 oracle with the HIP graphs replayed (the graphs themselves are still bit-exact fp32 builds of the kernel's own
    input).  Stated tolerance: mean |logit error| <= 2.5e-2 (measured 1.5e-2) and max <= 0.15 (measured 0.094) on logits of scale ~1 (a 1e-2 perturbation of
    the first EdgeConv's features passes three train-mode BatchNorms and re-routes max-pools on its way to the logits);
    the input gradient keeps its direction (cosine >= 0.9) -- the backward kernels themselves are compared at 2e-2 on a
    fixed forward state in test_edgeconv2_bf16_vs_f32."""
    from fissure_segmentation_amd.models.dgcnn import DGCNNSeg
    ref = fill_state_dict(ref_cpu.DGCNNSeg(k=20, in_features=3, num_classes=4), 7).train()
    net = DGCNNSeg(k=20, in_features=3, num_classes=4)
    net.load_state_dict(ref.state_dict())
    net = net.to(device).train()
    x = cloud(77, 2, 3, 1024)
    tape = GraphTape(fsg, monkeypatch)
    xt = G(x, device).requires_grad_(True)
    ctx = fsg.functional.mfma_operands("bf16") if how == "switch" else torch.autocast("cuda", dtype=torch.bfloat16)
    with ctx:
        y = net(xt)
    assert y.dtype == torch.float32
    gr = np.random.default_rng(78).standard_normal(tuple(y.shape)).astype(np.float32)
    y.backward(G(gr, device))
    tape.check_exact_and_replay(max_flipped_rows=0.2)     # bf16 features: the oracle's own graphs differ more often
    xr = torch.from_numpy(x).requires_grad_(True)
    yr = ref(xr)
    yr.backward(torch.from_numpy(gr))
    d = np.abs(N(y) - yr.detach().numpy())
    ga, gb = N(xt.grad).reshape(-1).astype(np.float64), xr.grad.numpy().reshape(-1).astype(np.float64)
    cos = float(ga @ gb / (np.linalg.norm(ga) * np.linalg.norm(gb)))
    print("\nBF16", how, "logit error mean", float(d.mean()), "max", float(d.max()), "grad_x cosine", cos)
    assert 1e-6 < float(d.max()) <= 0.15 and float(d.mean()) <= 2.5e-2 and cos >= 0.9


def _bf16_vs_ref_stats(y, yr, gx, gxr):
    d = np.abs(y - yr)
    ga, gb = gx.reshape(-1).astype(np.float64), gxr.reshape(-1).astype(np.float64)
    return float(d.mean()), float(d.max()), float(ga @ gb / (np.linalg.norm(ga) * np.linalg.norm(gb)))


def test_dgcnnseg_config4_bf16_vs_fp32_oracle(fsg, device, monkeypatch):
    """BASELINE config 4 AS STATED -- DGCNN-seg, 8192 points, k = 40, bf16 MFMA operands (the `bench.py --workload c4` default)
    -- on one cloud (the oracle's 8192^2 matrices and (1,128,8192,40) edge tensors stay small) against the fp32 oracle with the
    HIP graphs replayed.  Every graph the bf16 net builds is still an exact fp32 build of the kernel's own input (bit-exact vs
    the C oracle).  Stated tolerance, as at the small shape (test_dgcnnseg_bf16_mode_vs_fp32_oracle): mean |logit error| <=
    2.5e-2, max <= 0.2 on logits of scale ~1, input-gradient cosine >= 0.9; running statistics of the first EdgeConv within
    2e-2 of their scale.  Reference shape: bash_scripts/run_dgcnn_seg_experiments.sh:17, models/dgcnn.py:212-243."""
    from fissure_segmentation_amd.models.dgcnn import DGCNNSeg
    ref = fill_state_dict(ref_cpu.DGCNNSeg(k=40, in_features=3, num_classes=4), 7).train()
    net = DGCNNSeg(k=40, in_features=3, num_classes=4)
    net.load_state_dict(ref.state_dict())
    net = net.to(device).train()
    x = cloud(4000 + 8192 + 40, 1, 3, 8192)
    tape = GraphTape(fsg, monkeypatch)
    xt = G(x, device).requires_grad_(True)
    with fsg.functional.mfma_operands("bf16"):
        y = net(xt)
    assert y.dtype == torch.float32 and bool(torch.isfinite(y).all())
    node = _find_node(y.grad_fn, "_EdgeConv2Backward")
    assert node is not None and node.bf16 is True                 # the bf16 entry points ran, forward and backward
    gr = np.random.default_rng(4002).standard_normal(tuple(y.shape)).astype(np.float32)
    y.backward(G(gr, device))
    assert len(tape.calls) == 3 and tape.calls[1]["x"].shape == (1, 64, 8192)
    tape.check_exact_and_replay(max_flipped_rows=0.6)             # bf16 features and 40-long lists: a row of the oracle's own graph often differs in its last places (measured 34 %)
    xr = torch.from_numpy(x).requires_grad_(True)
    yr = ref(xr)
    yr.backward(torch.from_numpy(gr))
    mean, mx, cos = _bf16_vs_ref_stats(N(y), yr.detach().numpy(), N(xt.grad), xr.grad.numpy())
    print("\nBF16 config 4 (1 x 8192, k=40): logit error mean", mean, "max", mx, "grad_x cosine", cos,
          "rows with other neighbours in the oracle's own graphs", tape.flipped)
    assert 1e-6 < mx <= 0.2 and mean <= 2.5e-2 and cos >= 0.9
    for n, b in net.named_buffers():
        if n.startswith("ec1.") and "running" in n:
            want = dict(ref.named_buffers())[n].numpy()
            assert float(np.abs(N(b) - want).max()) <= 2e-2 * max(1.0, float(np.abs(want).max())), n


def test_dgcnnseg_config4_full_batch_bf16_properties(fsg, device, monkeypatch):
    """The per-GPU batch of BASELINE config 4 (4 clouds x 8192 points, k = 40, bf16 operands) is beyond the oracle's reach in
    test time, so it goes through size-independent properties: (1) finite logits and gradients for every parameter; (2) the
    bf16 step is bit-reproducible run to run (forward logits and every gradient); (3) against the fp32 HIP path -- itself
    pinned to the oracle at 1 x 8192 -- ON THE SAME GRAPHS (the bf16 run's graphs are replayed into the fp32 run: with its own
    graphs a dynamic-graph net answers a 1e-2 feature perturbation with other neighbours, measured mean |logit error| 0.07)
    logits within the stated bf16 tolerance (mean 3.5e-2, max 0.25) and gradient direction kept (cosine >= 0.9); (4) in eval mode (no cross-cloud BatchNorm coupling)
    the batch is independent clouds: cloud 2 of the batch equals the same cloud run alone (1e-5); (5) the
    hipGraph-replayed bf16 step equals the eager one bit for bit."""
    from fissure_segmentation_amd.models.dgcnn import DGCNNSeg
    F_hip = fsg.functional
    net = fill_state_dict(DGCNNSeg(k=40, in_features=3, num_classes=4), 7).to(device).train()
    x = G(cloud(4400, 4, 3, 8192), device)
    gr = G(np.random.default_rng(4401).standard_normal((4, 4, 8192)).astype(np.float32), device)

    def run(mode):
        for p in net.parameters():
            p.grad = None
        with F_hip.mfma_operands(mode):
            y = net(x)
        y.backward(gr)
        return y.detach().clone(), [p.grad.detach().clone() for p in net.parameters()]
    real = F_hip.knn_graph
    tape = []

    def recording(*a, **kw):
        out = real(*a, **kw)
        tape.append(out)
        return out
    monkeypatch.setattr(F_hip, "knn_graph", recording)
    y16, g16 = run("bf16")
    monkeypatch.setattr(F_hip, "knn_graph", real)
    assert len(tape) == 3 and tape[1].shape == (4, 8192, 40)
    assert bool(torch.isfinite(y16).all()) and all(bool(torch.isfinite(g).all()) for g in g16)
    y16b, g16b = run("bf16")
    assert torch.equal(y16, y16b) and all(torch.equal(a, b) for a, b in zip(g16, g16b))
    it = iter(tape)
    monkeypatch.setattr(F_hip, "knn_graph", lambda *a, **kw: next(it))
    y32, g32 = run("f32")
    monkeypatch.setattr(F_hip, "knn_graph", real)
    assert not torch.equal(y16, y32)
    d = (y16 - y32).abs()
    fa, fb = torch.cat([g.reshape(-1) for g in g16]).double(), torch.cat([g.reshape(-1) for g in g32]).double()
    cos = float(fa @ fb / (fa.norm() * fb.norm()))
    print("\nBF16 config 4 (4 x 8192, k=40) vs the fp32 HIP path on the same graphs: logit error mean", float(d.mean()), "max",
          float(d.max()), "parameter-gradient cosine", cos)
    assert float(d.mean()) <= 3.5e-2 and float(d.max()) <= 0.25 and cos >= 0.9      # measured 0.024 .. 0.029 / 0.15 .. 0.16 / 0.92 .. 0.94
    net.eval()
    with torch.no_grad(), F_hip.mfma_operands("bf16"):
        yb = net(x)
        y1 = net(x[2:3].contiguous())
    # (bf16 operand mode runs the head on vendor GEMMs, which may pick another kernel for another M: same values up to fp32
    # summation order, not the same bits)
    torch.testing.assert_close(yb[2:3], y1, rtol=1e-5, atol=1e-5)
    net.train()
    # hipGraph replay of the bf16 step == eager
    static_x = x.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), F_hip.mfma_operands("bf16"):
        for _ in range(2):
            for p in net.parameters():
                p.grad = None
            net(static_x).backward(gr)
    torch.cuda.current_stream().wait_stream(side)
    stats = {n: b.clone() for n, b in net.named_buffers()}
    g = torch.cuda.CUDAGraph()
    for p in net.parameters():
        p.grad = None
    with F_hip.mfma_operands("bf16"), torch.cuda.graph(g):
        yg = net(static_x)
        yg.backward(gr)
    for n, b in net.named_buffers():
        b.copy_(stats[n])
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(yg, y16)
    for p, want in zip(net.parameters(), g16):
        assert torch.equal(p.grad, want)


@pytest.mark.parametrize("B,C,Np,k,coords", [(2, 5, 300, 12, False), (1, 64, 1024, 20, False), (2, 15, 257, 8, True)])
def test_knn_gather_fused_entry_point(fsg, device, B, C, Np, k, coords):
    """fsg_knn_gather_fused_ws_f32 = the reference's create_neighbor_features with a dynamic graph (models/dgcnn.py:15-36):
    graph bit-exact vs the C oracle, edge tensor exact vs the oracle's edge features on that graph"""
    x = cloud(3000 + C, B, C, Np)
    edge, idx = fsg.functional.knn_edge_features(G(x, device), k, knn_only_over_coords=coords)
    idx_o, _ = c_api.knn_dense(x, k, c_knn=3 if coords else None, fix_diag=True)
    assert np.array_equal(N(idx), idx_o)
    assert np.array_equal(N(edge), c_api.edge_features(x, idx_o))
    # the entry point with the (B,N) scratch (two-phase kernel for every shape): same graph, same edge tensor
    from fissure_segmentation_amd import _lib
    xg = G(x, device)
    idx2 = torch.empty(B, Np, k, dtype=torch.int32, device=device)
    edge2 = torch.empty(B, 2 * C, Np, k, dtype=torch.float32, device=device)
    xx = torch.empty(B, Np, dtype=torch.float32, device=device)
    _lib.call("fsg_knn_gather_fused_f32", xg.data_ptr(), B, C, Np, k, 3 if coords else C, idx2.data_ptr(), edge2.data_ptr(),
              xx.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert torch.equal(idx2, idx) and torch.equal(edge2, edge)


def test_dgcnnreg_vs_reference_golden(fsg, device):
    """DGCNNReg (models/dgcnn.py:165-209) against the reference's own forward / backward: output 2e-4 (the regression head
    normalises FOUR rows per channel in train mode), gradients in norm"""
    from fissure_segmentation_amd.models.dgcnn import DGCNNReg
    g = load("dgcnnreg")
    net = fill_state_dict(DGCNNReg(k=8, in_features=3, num_classes=6), 871)
    assert [str(s) for s in g["keys"]] == list(net.state_dict().keys())
    net = net.to(device).train()
    y, gx = run_model(net, cloud(1871, 4, 3, 128), 2871, device)
    assert y.shape == (4, 6, 1)
    np.testing.assert_allclose(N(y), g["out"], rtol=2e-4, atol=2e-4)
    assert np.linalg.norm(N(gx) - g["grad_x"]) <= 2e-2 * np.linalg.norm(g["grad_x"])
    _check_packed_grads(net, g, 2e-2, 1e-3)


# ---------------------------------------------------------------------------------------------------------------------
# Point-wise layers on the bf16 matrix pipe (csrc/pointwise.hip): three bf16 pieces per fp32 operand, six MFMA products.

@pytest.mark.parametrize("M,N,K,tile", [(16384, 1024, 192, 1), (16384, 256, 256, 2), (16384, 128, 256, 3), (16384, 192, 448, 4),
                                        (1000, 77, 64, 0), (130, 1280, 32, 1), (64, 32, 96, 3), (4096, 4, 128, 3)])
def test_pw_linear_is_fp32_grade(fsg, device, M, N, K, tile):
    """fsg_pw_linear_f32 against fp64: its error must be of the size of an fp32 GEMM's (here: torch's, the vendor fp32 GEMM) --
    at most 1.5x its maximum and its rms, relative to sum |a||w| per output -- on operands with a wide dynamic range (log-uniform
    magnitudes over five decades, so that all three bf16 pieces carry weight); ragged M / N, strided views, bias."""
    F_hip = fsg.functional
    g = np.random.default_rng(M + N + K)
    a = (g.standard_normal((M, K + 8)) * 10.0 ** g.uniform(-3, 2, (M, K + 8))).astype(np.float32)
    w = (g.standard_normal((N, K + 5)) * 10.0 ** g.uniform(-3, 1, (N, K + 5))).astype(np.float32)
    b = g.standard_normal(N).astype(np.float32)
    at, wt, bt = G(a, device)[:, :K], G(w, device)[:, :K], G(b, device)          # views with padded row strides
    img = F_hip.pw_weight_image(wt)
    y = F_hip.pw_linear(at, img, N, bias=bt, tile=tile)
    ref = a[:, :K].astype(np.float64) @ w[:, :K].astype(np.float64).T + b
    mag = np.abs(a[:, :K]).astype(np.float64) @ np.abs(w[:, :K]).astype(np.float64).T + np.abs(b)
    e_pw = np.abs(N_(y) - ref) / mag
    e_t = np.abs(N_(at @ wt.t() + bt) - ref) / mag
    print("\nPW", (M, N, K), "max err / sum|a||w|: pw %.3g torch fp32 %.3g; rms pw %.3g torch %.3g" % (
        e_pw.max(), e_t.max(), np.sqrt((e_pw ** 2).mean()), np.sqrt((e_t ** 2).mean())))
    # measured at 16384 x 1024 x 192: max 9.6e-7 (vendor fp32 GEMM 1.3e-6), rms 8.4e-8 (1.08e-7); at 4096 x 4 x 128: max 5.5e-7
    # (3.5e-7), rms 7.7e-8 (5.3e-8)
    assert e_pw.max() <= max(1e-6, 2 * e_t.max())
    assert np.sqrt((e_pw ** 2).mean()) <= max(1.2e-7, 2 * np.sqrt((e_t ** 2).mean()))
    # exactness on small integers (every product and partial sum representable): bit-identical to the integer result
    ai = g.integers(-8, 9, (M, K)).astype(np.float32)
    wi = g.integers(-8, 9, (N, K)).astype(np.float32)
    yi = F_hip.pw_linear(G(ai, device), F_hip.pw_weight_image(G(wi, device)), N, tile=tile)
    assert np.array_equal(N_(yi), ai @ wi.T)


def _head_reference_fp64(levels, B, Npts, P, slope, train, eps=1e-5):
    """models/dgcnn.py:123-162 of the reference on point-major rows, in float64 torch ops (autograd gives the gradients):
    global feature (conv, BN, LeakyReLU, max over the points), cat with the repeated global vector, four head blocks"""
    def bn(y, g, b, rm, rv):
        if train:
            mu, var = y.mean(0), y.var(0, unbiased=False)
        else:
            mu, var = rm, rv
        return (y - mu) / torch.sqrt(var + eps) * g + b
    lr = torch.nn.functional.leaky_relu
    yg = lr(bn(levels @ P["Wg"].t(), P["gg"], P["bg"], P["rmg"], P["rvg"]), slope)
    g = yg.view(B, Npts, -1).max(1)[0]
    x = torch.cat([levels, g.repeat_interleave(Npts, 0)], 1)
    y = lr(bn(x @ P["W0"].t(), P["g0"], P["b0"], P["rm0"], P["rv0"]), slope)
    y = lr(bn(y @ P["W1"].t(), P["g1"], P["b1"], P["rm1"], P["rv1"]), slope)
    y = lr(bn(y @ P["W2"].t(), P["g2"], P["b2"], P["rm2"], P["rv2"]), slope)
    return y @ P["W3"].t() + P["b3"]


@pytest.mark.parametrize("B,Npts,train", [(8, 2048, True), (3, 256, True), (2, 512, False), (32, 256, True), (13, 1024, True)])
def test_seg_head_fused_vs_fp64_and_unfused(fsg, device, B, Npts, train):
    """The fused DGCNN-seg head (functional.seg_head: csrc/pointwise.hip) against the same head in float64 torch ops and
    against the round-2 path (vendor GEMMs + fsg_bn_act stages): logits 1e-4 of their scale, gradients of the input rows and
    of all 14 parameters 2e-3 in norm (and no worse than 3x the unfused path's own error), running statistics 1e-4.
    Reference: models/dgcnn.py:123-162,282-323."""
    from fissure_segmentation_amd.models.dgcnn import DGCNNSeg
    F_hip = fsg.functional
    torch.manual_seed(1000 + B + Npts)
    net = fill_state_dict(DGCNNSeg(k=20, in_features=3, num_classes=4), 31).to(device)
    net.train(train)
    rng = np.random.default_rng(100 + B + Npts)
    M = B * Npts
    lv = (0.4 + 0.6 * rng.standard_normal((M, 192))).astype(np.float32)       # what LeakyReLU'd EdgeConv features look like
    lv = np.where(lv < 0, 0.2 * lv, lv).astype(np.float32)
    gr = rng.standard_normal((M, 4)).astype(np.float32)
    names = {"Wg": net.global_feature[0].layers[0].weight, "gg": net.global_feature[0].layers[1].weight,
             "bg": net.global_feature[0].layers[1].bias}
    for i in range(3):
        names[f"W{i}"] = net.segmentation[i].layers[0].weight
        names[f"g{i}"], names[f"b{i}"] = net.segmentation[i].layers[1].weight, net.segmentation[i].layers[1].bias
    names["W3"], names["b3"] = net.segmentation[3].layers[0].weight, net.segmentation[3].layers[0].bias
    bns = {"g": net.global_feature[0].layers[1], "0": net.segmentation[0].layers[1], "1": net.segmentation[1].layers[1],
           "2": net.segmentation[2].layers[1]}
    with torch.no_grad():                                                     # both signs of gamma, non-trivial running stats
        for k, bn in bns.items():
            bn.weight.mul_(torch.where(torch.arange(bn.weight.numel(), device=device) % 5 == 0, -1.0, 1.0))
            bn.running_mean.normal_(0, 0.3)
            bn.running_var.uniform_(0.5, 2.0)
    stats0 = {k: (bn.running_mean.clone(), bn.running_var.clone()) for k, bn in bns.items()}
    P64 = {k: v.detach().double().view(v.shape[0], -1).squeeze(-1).clone().requires_grad_(True) if v.dim() > 1 else
           v.detach().double().clone().requires_grad_(True) for k, v in names.items()}
    for k, (rm, rv) in stats0.items():
        P64["rm" + k], P64["rv" + k] = rm.double(), rv.double()
    x64 = G(lv, device).double().requires_grad_(True)
    y64 = _head_reference_fp64(x64, B, Npts, P64, 0.2, train)
    y64.backward(G(gr, device).double())

    def run(fused):
        old = F_hip.set_fused_head(fused)
        try:
            for k, bn in bns.items():
                bn.running_mean.copy_(stats0[k][0])
                bn.running_var.copy_(stats0[k][1])
            for p in net.parameters():
                p.grad = None
            xt = G(lv, device).requires_grad_(True)
            with F_hip.deferred_bn_counters():
                if fused:
                    assert F_hip.seg_head_supported(xt, B, Npts, *(names[k].view(names[k].shape[0], -1) for k in ("Wg", "W0", "W1", "W2", "W3")))
                    y = F_hip.seg_head(xt, B, Npts, net.global_feature[0], list(net.segmentation))
                else:
                    y = net._head_unfused(xt, B, Npts)
            y.backward(G(gr, device))
            return (y.detach(), xt.grad, {k: v.grad.view(v.shape[0], -1).squeeze(-1) if v.dim() > 1 else v.grad for k, v in names.items()},
                    {k: (bn.running_mean.clone(), bn.running_var.clone()) for k, bn in bns.items()})
        finally:
            F_hip.set_fused_head(old)
    yf, gxf, gpf, stf = run(True)
    yu, gxu, gpu_, stu = run(False)
    scale = float(y64.abs().max())
    e_f, e_u = float((yf.double() - y64).abs().max()) / scale, float((yu.double() - y64).abs().max()) / scale
    print("\nHEAD", (B, Npts, train), "logit error / scale: fused %.3g unfused %.3g" % (e_f, e_u))
    assert e_f <= 1e-4

    def rel(a, b):
        return float((a.double() - b).norm() / b.norm().clamp_min(1e-30))

    # input-row gradient: a max-pool near-tie (two points of a cloud within fp32 rounding of each other in one of the 1024
    # global-feature channels) legitimately routes that channel's gradient to the other row.  EXACTLY those rows are left out:
    # the winner and the runner-up of every (cloud, channel) whose fp64 margin is below the fp32 noise of the activation
    # (8 roundings of its magnitude: a 192-term fp32 product chain + BatchNorm); round 3 dropped the four worst rows instead.
    with torch.no_grad():
        bnp = lambda yv: ((yv - (yv.mean(0) if train else P64["rmg"])) /
                          torch.sqrt((yv.var(0, unbiased=False) if train else P64["rvg"]) + 1e-5) * P64["gg"] + P64["bg"])
        ag = torch.nn.functional.leaky_relu(bnp(x64 @ P64["Wg"].t()), 0.2).view(B, Npts, -1)
        top2 = ag.topk(2, dim=1)
        margin = top2.values[:, 0] - top2.values[:, 1]                                   # (B, 1024)
        noise = 8 * 2.0 ** -24 * top2.values[:, 0].abs().clamp_min(1e-3 * float(ag.abs().max()))
        tie = margin <= noise
        keep = torch.ones(M, dtype=torch.bool, device=ag.device)
        bidx = torch.arange(B, device=ag.device).view(B, 1).expand_as(tie)
        for r in range(2):
            keep[(bidx * Npts + top2.indices[:, r])[tie]] = False
    n_tie_rows = int((~keep).sum())
    assert n_tie_rows <= 16, n_tie_rows                                                   # a handful at most, or the inputs are degenerate

    def rel_rows(a, b):
        d = (a.double() - b).pow(2).sum(1)
        return float((d[keep].sum().sqrt()) / b.norm().clamp_min(1e-30))
    errs = {"x": (rel_rows(gxf, x64.grad), rel_rows(gxu, x64.grad))}
    print("HEAD rows left out as max-pool near-ties (fp64 margin below fp32 noise):", n_tie_rows)
    gmax = max(float(v.grad.norm()) for v in P64.values() if v.grad is not None)
    for k in names:
        if P64[k].grad.norm() < 1e-6 * gmax:
            continue                    # mathematically zero gradients (a bias in front of a train-mode BatchNorm): noise on all sides
        errs[k] = (rel(gpf[k], P64[k].grad), rel(gpu_[k], P64[k].grad))
    print("HEAD gradient errors vs fp64 (fused, unfused):", {k: ("%.2g" % a, "%.2g" % b) for k, (a, b) in errs.items()})
    bad = {k: v for k, v in errs.items() if v[0] > max(2e-3, 3 * v[1])}
    assert not bad, bad
    if train:
        for k in bns:
            torch.testing.assert_close(stf[k][0], stu[k][0], rtol=1e-4, atol=1e-5)
            torch.testing.assert_close(stf[k][1], stu[k][1], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B,C,Np,k", [(8, 3, 2048, 20), (2, 3, 1024, 40)])
def test_ec2s_is_fp32_grade(fsg, device, B, C, Np, k):
    """The split-bf16 kernels of the two-layer EdgeConv (ec2s_fwd / ec2s_bwd, csrc/edgeconv2.hip: three bf16 pieces per fp32
    operand, six v_mfma_f32_32x32x16_bf16 products, fp32 accumulation -- the kernels of the headline path since round 3) must be
    as close to an fp64 composition of the same layers on the same graph as the fp32-MFMA kernels they replaced
    (v_mfma_f32_32x32x2_f32: an exact fp32 fma chain): output and every gradient within 2 x the fp32 kernels' error (with a
    floor of 2e-6 of the output scale / 3e-5 in norm for the gradients, where that error happens to be tiny), and within
    1e-5 / 1e-3 absolutely.
    Reference op: models/dgcnn.py:226-243 (EdgeConv with two shared-MLP layers)."""
    import ctypes
    from fissure_segmentation_amd.norm import BatchNorm2d
    F_hip = fsg.functional
    lib = fsg._lib.lib
    lib.fsg_debug_ec2_use_fp32_mfma.argtypes = [ctypes.c_int]
    lib.fsg_debug_ec2_use_fp32_mfma.restype = None
    torch.manual_seed(5 + Np)
    x = torch.rand(B, C, Np, device=device)
    conv1 = torch.nn.Conv2d(2 * C, 64, 1, bias=False).to(device)
    conv2 = torch.nn.Conv2d(64, 64, 1, bias=False).to(device)
    bn1, bn2 = BatchNorm2d(64).to(device), BatchNorm2d(64).to(device)
    with torch.no_grad():
        for bn in (bn1, bn2):
            bn.weight.copy_((torch.rand(64, device=device) + 0.5) * torch.where(torch.arange(64, device=device) % 7 == 0, -1.0, 1.0))
            bn.bias.copy_(torch.randn(64, device=device) * 0.2)
    idx = F_hip.knn_graph(x, k, c_knn=3)
    gr = torch.randn(B, 64, Np, device=device)
    mods = (conv1, conv2, bn1, bn2)

    def run():
        for m in mods:
            for p in m.parameters():
                p.grad = None
        xt = x.clone().requires_grad_(True)
        with F_hip.deferred_bn_counters():
            y = F_hip.edgeconv2(xt, idx, conv1.weight, bn1, conv2.weight, bn2, 0.2)
        y.backward(gr)
        return dict(out=y.detach().double(), grad_x=xt.grad.double(), grad_w1=conv1.weight.grad.double().view(64, 2 * C),
                    grad_w2=conv2.weight.grad.double().view(64, 64), grad_g1=bn1.weight.grad.double(), grad_b1=bn1.bias.grad.double(),
                    grad_g2=bn2.weight.grad.double(), grad_b2=bn2.bias.grad.double())
    try:
        lib.fsg_debug_ec2_use_fp32_mfma(0)
        split = run()
        lib.fsg_debug_ec2_use_fp32_mfma(1)
        exact = run()
    finally:
        lib.fsg_debug_ec2_use_fp32_mfma(0)
    # fp64 composition (models/dgcnn.py:28-36 edge features, two conv + train-mode BatchNorm + LeakyReLU blocks, max over k)
    xd = x.double().requires_grad_(True)
    w1 = conv1.weight.detach().double().view(64, 2 * C).requires_grad_(True)
    w2 = conv2.weight.detach().double().view(64, 64).requires_grad_(True)
    g1, b1 = bn1.weight.detach().double().requires_grad_(True), bn1.bias.detach().double().requires_grad_(True)
    g2, b2 = bn2.weight.detach().double().requires_grad_(True), bn2.bias.detach().double().requires_grad_(True)
    xp = xd.transpose(1, 2)
    nb = xp.reshape(B * Np, C)[(idx.long() + (torch.arange(B, device=device) * Np).view(B, 1, 1)).reshape(-1)].view(B, Np, k, C)
    ctr = xp.unsqueeze(2).expand(B, Np, k, C)
    e = torch.cat([nb - ctr, ctr], -1)

    def block(t, w, g, b_):
        t = t @ w.t()
        mu, var = t.mean((0, 1, 2)), t.var((0, 1, 2), unbiased=False)
        return torch.nn.functional.leaky_relu((t - mu) / torch.sqrt(var + 1e-5) * g + b_, 0.2)
    yd = block(block(e, w1, g1, b1), w2, g2, b2).max(2)[0].permute(0, 2, 1)
    yd.backward(gr.double())
    want = dict(out=yd.detach(), grad_x=xd.grad, grad_w1=w1.grad, grad_w2=w2.grad, grad_g1=g1.grad, grad_b1=b1.grad,
                grad_g2=g2.grad, grad_b2=b2.grad)
    rep, bad = {}, []
    for n in want:
        sc = float(want[n].abs().max())
        if n == "out":
            es, ee = float((split[n] - want[n]).abs().max()) / sc, float((exact[n] - want[n]).abs().max()) / sc
            lim, floor = 1e-5, 2e-6
        else:
            es, ee = float((split[n] - want[n]).norm() / want[n].norm()), float((exact[n] - want[n]).norm() / want[n].norm())
            lim, floor = 1e-3, 3e-5
        rep[n] = ("%.2e" % es, "%.2e" % ee)
        if es > min(lim, max(2 * ee, floor)):
            bad.append((n, es, ee))
    print("\nEC2S vs fp64 (split-bf16 kernels, fp32-MFMA kernels):", rep)
    assert not bad, bad


def test_seg_head_backward_is_bitwise_reproducible(fsg, device):
    """no float atomics anywhere in the fused head: two runs give identical logits and gradients"""
    from fissure_segmentation_amd.models.dgcnn import DGCNNSeg
    F_hip = fsg.functional
    net = fill_state_dict(DGCNNSeg(k=20, in_features=3, num_classes=4), 33).to(device).train()
    rng = np.random.default_rng(5)
    lv, gr = rng.standard_normal((4 * 1024, 192)).astype(np.float32), rng.standard_normal((4 * 1024, 4)).astype(np.float32)
    outs = []
    for _ in range(2):
        for p in net.parameters():
            p.grad = None
        xt = G(lv, device).requires_grad_(True)
        y = F_hip.seg_head(xt, 4, 1024, net.global_feature[0], list(net.segmentation))
        y.backward(G(gr, device))
        outs.append([y.detach().clone(), xt.grad.clone()] + [p.grad.clone() for p in net.parameters() if p.grad is not None])
    assert len(outs[0]) == 2 + 14
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("M,N,K", [(16384, 96, 32), (4096, 64, 64), (1000, 256, 128), (256, 512, 512), (64, 35, 96)])
def test_pw_bf16_operand_mode_matches_rounded_operands(fsg, device, M, N, K):
    """fsg_pw_linear_bf16 / fsg_pw_tn_bf16: ONE bf16 piece per operand, fp32 accumulation.  The result must be the product of
    the bf16-ROUNDED operands (round-to-nearest-even, as torch's .bfloat16()) up to fp32 accumulation order -- 2e-6 of
    sum |a||w| -- for the forward product, the input gradient (transposed weight view) and the weight gradient."""
    F_hip = fsg.functional
    g = np.random.default_rng(M + N + K)
    a = g.standard_normal((M, K)).astype(np.float32)
    w = (0.2 * g.standard_normal((N, K))).astype(np.float32)
    b = g.standard_normal(N).astype(np.float32)
    gy = g.standard_normal((M, N)).astype(np.float32)
    at, wt, bt, gt = G(a, device), G(w, device), G(b, device), G(gy, device)
    r = lambda t: t.bfloat16().double()
    y = F_hip.pw_linear_bf16(at, wt, bt)
    want = r(at) @ r(wt).t() + bt.double()
    mag = r(at).abs() @ r(wt).abs().t() + bt.abs().double()
    assert float(((y.double() - want).abs() / mag).max()) <= 2e-6
    assert float((y.double() - (at.double() @ wt.double().t() + bt.double())).abs().max()) > 1e-4      # it IS the bf16 product
    if N % 32 == 0:
        dx = F_hip.pw_linear_bf16(gt, wt.t(), None)
        want = r(gt) @ r(wt)
        assert float(((dx.double() - want).abs() / (r(gt).abs() @ r(wt).abs())).max()) <= 2e-6
    dw = F_hip.pw_tn_bf16(gt, at)
    want = r(gt).t() @ r(at)
    assert float(((dw.double() - want).abs() / (r(gt).abs().t() @ r(at).abs())).max()) <= 4e-6
    dw2 = F_hip.pw_tn_bf16(gt, at)
    assert torch.equal(dw, dw2)                                                                         # fixed slice order


@pytest.mark.parametrize("n,m,c,ns", [(4096, 1024, 32, 16), (300, 77, 64, 16), (64, 256, 512, 3), (16384, 4096, 3, 8)])
def test_pointops_fused_glue_vs_torch(fsg, device, n, m, c, ns):
    """The one-launch forms of the PointTransformer's glue against the tensor expressions of the reference they replace
    (models/pointtransformer/pointops.py:100-123 queryandgroup, :198-215 interpolation; seg_model.py:77-83 max-pool), values and
    gradients: fsg_group_xyz_feat_*, fsg_interp_* (k = 3 and up to 8), fsg_rows_max_*."""
    F_hip = fsg.functional
    g = np.random.default_rng(n + m + c)
    xyz, nxyz = G(g.standard_normal((n, 3)).astype(np.float32), device), G(g.standard_normal((m, 3)).astype(np.float32), device)
    feat = G(g.standard_normal((n, c)).astype(np.float32), device)
    idx = G(g.integers(0, n, (m, ns)).astype(np.int32), device)
    # queryandgroup(use_xyz=True)
    f1, f2 = feat.clone().requires_grad_(True), feat.clone().requires_grad_(True)
    got = F_hip.group_xyz_feat(xyz, nxyz, f1, idx)
    want = torch.cat((xyz[idx.long()] - nxyz.unsqueeze(1), f2[idx.long()]), -1)
    assert torch.equal(got, want)
    go = G(g.standard_normal((m, ns, 3 + c)).astype(np.float32), device)
    got.backward(go)
    want.backward(go)
    assert float((f1.grad - f2.grad).abs().max()) <= 1e-5 * max(1.0, float(f2.grad.abs().max()))      # atomics: order of the sums
    # interpolation, k nearest = the first k columns of idx with some squared distances (one exactly zero)
    for k in (3, min(ns, 8)):
        d2 = G(g.uniform(0.0, 2.0, (m, k)).astype(np.float32), device)
        d2[0, 0] = 0.0
        f1, f2 = feat.clone().requires_grad_(True), feat.clone().requires_grad_(True)
        got = F_hip.interpolate(f1, idx[:, :k].contiguous(), d2)
        w = 1.0 / (torch.sqrt(d2) + 1e-8)
        w = w / w.sum(dim=1, keepdim=True)
        want = (f2[idx[:, :k].long()] * w.unsqueeze(-1)).sum(dim=1)
        assert float((got - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max()))
        go = G(g.standard_normal((m, c)).astype(np.float32), device)
        got.backward(go)
        want.backward(go)
        assert float((f1.grad - f2.grad).abs().max()) <= 1e-5 * max(1.0, float(f2.grad.abs().max()))
    # max over the neighbour rows (with exact ties: the gradient goes to ONE row, as torch's does)
    x = g.standard_normal((m, ns, c)).astype(np.float32)
    x[:, 1, :] = x[:, 0, :]
    x1, x2 = G(x, device).requires_grad_(True), G(x, device).requires_grad_(True)
    got, want = F_hip.rows_max(x1), x2.max(dim=1)[0]
    assert torch.equal(got, want)
    go = G(g.standard_normal((m, c)).astype(np.float32), device)
    got.backward(go)
    want.backward(go)
    assert torch.equal(x1.grad.sum(1), x2.grad.sum(1)) and int((x1.grad != 0).sum()) <= m * c
    assert torch.equal((x1.grad != 0).sum(1) <= 1, torch.ones(m, c, dtype=torch.bool, device=device))


@pytest.mark.parametrize("M,N,K", [(16384, 96, 32), (256, 256, 256), (1000, 77, 64), (64, 512, 512), (4096, 64, 67)])
def test_gemm_small_bf16_matches_rounded_operands(fsg, device, M, N, K):
    """fsg_gemm_small_bf16 (the nn.Linear products of the PointTransformer in bf16 mode): operands rounded to bf16 inside the
    kernel, fp32 accumulation.  Forward x W^T + b, input gradient dY W and weight gradient dY^T X with its row-sum by-product must
    be the products of the bf16-ROUNDED operands (torch's .bfloat16(), nearest even) up to fp32 summation order -- 2e-6 of
    sum |a||w| -- and differ from the fp32 products (it IS the bf16 product)."""
    F_hip = fsg.functional
    g = np.random.default_rng(M + N + K)
    at, wt, bt, gt = (G(v.astype(np.float32), device) for v in (g.standard_normal((M, K)), 0.2 * g.standard_normal((N, K)),
                                                                g.standard_normal(N), g.standard_normal((M, N))))
    r = lambda t: t.bfloat16().double()  # noqa: E731

    def check(got, want, mag, lim=2e-6):
        assert float(((got.double() - want).abs() / mag).max()) <= lim
    y = F_hip.gemm_small(at, K, 1, wt, 1, K, bt, M, N, K, bf16=True)
    check(y, r(at) @ r(wt).t() + bt.double(), r(at).abs() @ r(wt).abs().t() + bt.abs().double())
    assert float((y.double() - (at.double() @ wt.double().t() + bt.double())).abs().max()) > 1e-4
    dx = F_hip.gemm_small(gt, N, 1, wt, K, 1, None, M, K, N, bf16=True)
    check(dx, r(gt) @ r(wt), r(gt).abs() @ r(wt).abs())
    dw, db = F_hip.gemm_small(gt, 1, N, at, K, 1, None, N, K, M, rowsum=True, bf16=True)
    check(dw, r(gt).t() @ r(at), r(gt).abs().t() @ r(at).abs(), 4e-6)
    check(db, r(gt).sum(0), r(gt).abs().sum(0), 4e-6)
    dw2, db2 = F_hip.gemm_small(gt, 1, N, at, K, 1, None, N, K, M, rowsum=True, bf16=True)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)                                                # fixed split order


def test_pointtransformer_bf16_mode_vs_fp32_oracle(fsg, device):
    """BASELINE config 3 names bf16 (the reference trains PointTransformer under autocast, model_trainer.py:75-76,157).  With
    functional.mfma_operands("bf16") + set_bf16_linear(True) every nn.Linear of seg_model.py -- q/k/v, linear1/linear3, the
    transitions, the classifier -- runs forward AND backward on bf16 operands with fp32 accumulation (fsg_gemm_small_bf16);
    graphs, BatchNorm statistics, the attention layer's internal c -> c/8 contraction and all stored tensors stay fp32.
    THIS test: fill_state_dict weights, where the architecture amplifies any operand rounding (60 BatchNorms, 18 softmax layers):
    measured against the fp32 oracle on 2 x 2048 points mean |logit error| 0.13, max 1.36 on logits of scale ~1, parameter-
    gradient cosine 0.17 (fp32 mode: 0.9999) -- and the ORACLE ITSELF under the reference's autocast measures 0.041 / 0.50 / 0.76
    (fp16) and 0.16 / 1.7 / -0.10 (bf16) at these weights (tools/pt_autocast_oracle.py): the gradient direction of a bf16 run
    means nothing here, for the reference either.  It pins the behaviour (bounds 0.2 / 2.0 on the logits, the oracle's own bf16
    autocast figures: gross breakage, not closeness) and that the default stays fp32-exact; the parity bar for the mode is held at
    weights a training run visits, in test_pointtransformer_bf16_mode_after_training_vs_fp32_oracle below."""
    from fissure_segmentation_amd.models.pointtransformer.seg_model import PointTransformerCompatibility
    ref = fill_state_dict(ref_cpu.PointTransformerCompatibility(6, 4), 803).train()
    net = PointTransformerCompatibility(6, 4)
    net.load_state_dict(ref.state_dict())
    net = net.to(device).train()
    x = cloud(4310, 2, 6, 2048)
    gr = np.random.default_rng(4311).standard_normal((2, 4, 2048)).astype(np.float32)

    def run(mode, linear):
        for p in net.parameters():
            p.grad = None
        xt = G(x, device)
        old = fsg.functional.set_bf16_linear(linear)
        try:
            with fsg.functional.mfma_operands(mode):
                y = net(xt)
            y.backward(G(gr, device))
        finally:
            fsg.functional.set_bf16_linear(old)
        return y.detach().clone(), torch.cat([p.grad.reshape(-1) for p in net.parameters()]).double()
    y16, g16 = run("bf16", True)
    y32, g32 = run("f32", False)
    y16off, _ = run("bf16", False)                 # the operand mode alone leaves the PointTransformer on fp32 products
    assert y16.dtype == torch.float32 and not torch.equal(y16, y32) and torch.equal(y16off, y32)
    xr = torch.from_numpy(x)
    yr = ref(xr)
    yr.backward(torch.from_numpy(gr))
    gref = torch.cat([p.grad.reshape(-1) for p in ref.parameters()]).double()
    d = (y16.cpu() - yr.detach()).abs()
    cos = float(g16.cpu() @ gref / (g16.norm().cpu() * gref.norm()))
    cos32 = float(g32.cpu() @ gref / (g32.norm().cpu() * gref.norm()))
    print("\nBF16 PointTransformer (2 x 2048): logit error mean", float(d.mean()), "max", float(d.max()), "gradient cosine", cos,
          "(fp32 mode:", cos32, ")")
    assert cos32 >= 0.999
    assert float(d.mean()) <= 0.2 and float(d.max()) <= 2.0 and np.isfinite(cos)


def test_pointtransformer_bf16_mode_after_training_vs_fp32_oracle(fsg, device):
    """The bf16 operand mode of the PointTransformer (see the test above) at weights a training run actually visits: the oracle
    net, torch's default initialisation, 20 Adam steps (lr 1e-3, cross-entropy on random labels, CPU, fp32), then HIP bf16 mode
    against the fp32 oracle at those weights on 2 x 2048 points.
    Why not at fill_state_dict weights: there the architecture itself amplifies any operand rounding -- the ORACLE under
    torch.autocast (the reference's own mixed-precision step, model_trainer.py:75-76,157) measures, against its fp32 run,
        fill_state_dict weights:  fp16 autocast mean |logit error| 0.041, max 0.50, gradient cosine 0.76;  bf16 autocast 0.16 / 1.7 / -0.10
        default initialisation:   fp16 0.013 / 0.11 / 0.88;   bf16 0.064 / 0.62 / 0.13
        after 20 Adam steps:      fp16 0.0010 / 0.006 / 0.996;  bf16 0.0080 / 0.075 / 0.964
    (CPU, same inputs; tools/pt_autocast_oracle.py prints them).  The bar here is the one the round-3 review set: mean |logit
    error| <= 3e-2 and parameter-gradient cosine >= 0.95 against the fp32 oracle."""
    from fissure_segmentation_amd.models.pointtransformer.seg_model import PointTransformerCompatibility
    torch.manual_seed(0)
    ref = ref_cpu.PointTransformerCompatibility(6, 4).train()
    x = cloud(4310, 2, 6, 2048)
    xr = torch.from_numpy(x)
    lab = torch.randint(0, 4, (2, 2048))
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    for _ in range(20):
        opt.zero_grad()
        torch.nn.functional.cross_entropy(ref(xr), lab).backward()
        opt.step()
    net = PointTransformerCompatibility(6, 4)
    net.load_state_dict(ref.state_dict())
    net = net.to(device).train()
    gr = np.random.default_rng(4311).standard_normal((2, 4, 2048)).astype(np.float32)
    for p in ref.parameters():
        p.grad = None
    yr = ref(xr)
    yr.backward(torch.from_numpy(gr))
    gref = torch.cat([p.grad.reshape(-1) for p in ref.parameters()]).double()

    def run(mode, linear):
        for p in net.parameters():
            p.grad = None
        old = fsg.functional.set_bf16_linear(linear)
        try:
            with fsg.functional.mfma_operands(mode):
                y = net(G(x, device))
            y.backward(G(gr, device))
        finally:
            fsg.functional.set_bf16_linear(old)
        g = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).double().cpu()
        d = (y.detach().cpu() - yr.detach()).abs()
        return float(d.mean()), float(d.max()), float(g @ gref / (g.norm() * gref.norm()))
    m16, x16, c16 = run("bf16", True)
    m32, x32, c32 = run("f32", False)
    print("\nBF16 PointTransformer after 20 Adam steps (2 x 2048): logit error mean %.4g max %.4g, gradient cosine %.4f; fp32 mode: "
          "%.3g / %.3g / %.6f" % (m16, x16, c16, m32, x32, c32))
    assert c32 >= 0.9999 and m32 <= 1e-4
    assert m16 <= 3e-2 and c16 >= 0.95


@pytest.mark.parametrize("B,Np,k,two_layer", [(2, 2048, 20, True), (3, 1024, 20, False), (1, 8192, 40, False)])
def test_knn_prepared_by_edgeconv_equals_plain_build(fsg, device, B, Np, k, two_layer):
    """The feature-space graph build prepared by its producer (fsg_edgeconv_apply_f32 with a knn_workspace, then
    fsg_knn_dense_prepared_f32) against the plain build (fsg_knn_dense_ws_f32) of the same features and against the C oracle:
    the block's outputs are bit-identical with and without the prep, the graphs are bit-identical (reference:
    models/dgcnn.py:212-243 EdgeConv -> utils/general_utils.py:315-327 knn of its output)."""
    from fissure_segmentation_amd.models.dgcnn import EdgeConv
    F_hip = fsg.functional
    C = 3 if two_layer else 64
    ec = fill_state_dict(EdgeConv(C, [64, 64] if two_layer else [64], k, first_layer=two_layer), 91).to(device).train()
    x = G(cloud(9100 + Np, B, C, Np), device)
    g0 = F_hip.knn_graph(x, k, c_knn=3 if two_layer else None)
    with torch.no_grad():
        ws = F_hip.knn_prep_workspace(B, Np, 64, device)
        assert ws is not None
        out_a, pm_a = ec(x, g0, both=True)
        out_b, pm_b = ec(x, g0, both=True, knn_ws=ws)
    assert torch.equal(out_a, out_b) and torch.equal(pm_a, pm_b)
    plain = F_hip.knn_graph(out_a, k)
    prepared = F_hip.knn_graph(out_b, k, prepared=(ws, pm_b))
    assert torch.equal(plain, prepared)
    if Np <= 2048:
        want, _ = c_api.knn_dense(N(out_a), k, fix_diag=True)
        assert np.array_equal(N(prepared), want)
    # a cloud with a far outlier in the features: the marked cloud takes the exact slow path in both builds
    if not two_layer and Np == 1024:
        x2 = x.clone()
        x2[0, :, 7] *= 1.0e4
        with torch.no_grad():
            o_a, p_a = ec(x2, g0, both=True)
            o_b, p_b = ec(x2, g0, both=True, knn_ws=ws)
        assert torch.equal(F_hip.knn_graph(o_a, k), F_hip.knn_graph(o_b, k, prepared=(ws, p_b)))
    assert F_hip.knn_prep_workspace(B, 1000, 64, device) is None and F_hip.knn_prep_workspace(B, Np, 32, device) is None
    # round 4: with the NEXT block's [W_rel ; W_ctr - W_rel] weight the same pass also emits that block's per-point rows
    # (fsg_edgeconv_apply_pq_f32): outputs and prep unchanged bit for bit, rows = out_pm W^T (exact fp32 matrix instruction:
    # an fp32 fma chain's error against float64), and a block handed those rows computes what it computes from its own product
    nxt = fill_state_dict(EdgeConv(64, [64], k), 92).to(device).train()
    (w_next,) = EdgeConv.pq_weights([nxt])
    with torch.no_grad():
        out_c, pm_c, pq_next = ec(x, g0, both=True, knn_ws=ws, w_next=w_next)
    assert torch.equal(out_c, out_a) and torch.equal(pm_c, pm_a) and pq_next is not None and tuple(pq_next.shape) == (B, Np, 128)
    assert torch.equal(F_hip.knn_graph(out_c, k, prepared=(ws, pm_c)), plain)
    want_pq = pm_a.double() @ w_next.detach().double().t()
    assert float((pq_next.double() - want_pq).abs().max()) <= 2e-6 * float(want_pq.abs().max())
    xa = out_a.clone().requires_grad_(True)
    xb = out_a.clone().requires_grad_(True)
    with F_hip.deferred_bn_counters():
        ya = nxt(xa, plain, x_pm=xa.transpose(1, 2).contiguous())
    for p_ in nxt.parameters():
        p_.grad = None
    ga = torch.randn_like(ya)
    ya.backward(ga)
    grads_a = [p_.grad.clone() for p_ in nxt.parameters()]
    for p_ in nxt.parameters():
        p_.grad = None
    with F_hip.deferred_bn_counters():
        yb = nxt(xb, plain, x_pm=xb.transpose(1, 2).contiguous(), pq_given=pq_next)
    yb.backward(ga)
    assert float((ya - yb).abs().max()) <= 1e-5 * float(ya.abs().max())
    for a_, p_ in zip(grads_a, nxt.parameters()):
        assert float((a_ - p_.grad).norm()) <= 1e-4 * float(a_.norm()) + 1e-8
    assert float((xa.grad - xb.grad).norm()) <= 1e-4 * float(xa.grad.norm())
