"""CPU tests: the oracle (C primitives + pure-torch restatement) against the golden vectors that
oracle/make_golden.py produced by running the real reference.  This is what pins the oracle."""
import glob
import os

import numpy as np
import pytest
import torch

from golden_util import GOLDEN_DIR, cloud, fill_state_dict, load, seg_loss_case
from knn_check import assert_knn_equal
from oracle import c_api, ref_cpu

TOL = dict(rtol=1e-4, atol=1e-4)  # north_star: features/logits within 1e-4 fp32


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


KNN_FIXTURES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "knn_s*.npz")))


@pytest.mark.parametrize("name", KNN_FIXTURES)
def test_knn_c_oracle_vs_reference(name):
    g = load(name)
    B, C, N, k = int(g["B"]), int(g["C"]), int(g["N"]), int(g["k"])
    x = cloud(int(g["seed"]), B, C, N)
    for sl in (1, 0):
        idx, dist = c_api.knn_dense(x, k, fix_diag=True, drop_first=not sl)
        assert_knn_equal(x, idx, g[f"idx_self{sl}"], k, drop_first=not sl)
        np.testing.assert_allclose(np.sort(dist, -1), np.sort(g[f"dist_self{sl}"], -1), rtol=1e-4, atol=2e-4)
    idx, _ = c_api.knn_dense(x, k, fix_diag=False, drop_first=False)
    assert_knn_equal(x, idx, g["idx_open"], k, fix_diag=False)
    if "idx_coords_self1" in g:
        idx, _ = c_api.knn_dense(x, k, c_knn=3, fix_diag=True)
        assert_knn_equal(x, idx, g["idx_coords_self1"], k, c_knn=3)


@pytest.mark.parametrize("name", KNN_FIXTURES[:3])
def test_knn_torch_restatement_vs_reference(name):
    g = load(name)
    B, C, N, k = int(g["B"]), int(g["C"]), int(g["N"]), int(g["k"])
    x = cloud(int(g["seed"]), B, C, N)
    for sl in (1, 0):
        idx = ref_cpu.knn(T(x), k, self_loop=bool(sl)).numpy()
        assert_knn_equal(x, idx, g[f"idx_self{sl}"], k, drop_first=not sl)
    assert_knn_equal(x, ref_cpu.knn_opensrc(T(x), k).numpy(), g["idx_open"], k, fix_diag=False)


def test_edge_features_exact():
    g = load("edge_feat_s201")
    x = cloud(201, 2, 5, 64)
    idx = g["idx"].astype(np.int32)
    e = c_api.edge_features(x, idx)
    assert np.array_equal(e, g["edge"]) and np.array_equal(e, g["edge_open"])
    assert np.array_equal(ref_cpu.edge_features(T(x), T(idx.astype(np.int64))).numpy(), g["edge"])
    gr = np.random.default_rng(int(g["gseed"])).standard_normal(e.shape).astype(np.float32)
    np.testing.assert_allclose(c_api.edge_features_bwd(gr, idx), g["grad_x"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", ["edgeconv_first", "edgeconv_feat", "edgeconv_c15"])
def test_edgeconv_restatement(name):
    g = load(name)
    seed, cin, k, N = int(g["seed"]), int(g["cin"]), int(g["k"]), int(g["N"])
    ec = fill_state_dict(ref_cpu.EdgeConv(cin, [int(c) for c in g["couts"]], k, first_layer=bool(g["first"])), seed)
    ec.train()
    x = T(cloud(seed + 1000, 2, cin, N)).requires_grad_(True)
    y = ec(x)
    gr = np.random.default_rng(seed + 2000).standard_normal(tuple(y.shape)).astype(np.float32)
    y.backward(T(gr))
    np.testing.assert_allclose(y.detach().numpy(), g["y"], **TOL)
    np.testing.assert_allclose(x.grad.numpy(), g["grad_x"], **TOL)
    for n, p in ec.named_parameters():
        np.testing.assert_allclose(p.grad.numpy(), g["grad_" + n], rtol=1e-3, atol=2e-4)
    for n, b in ec.named_buffers():
        if "running" in n:
            np.testing.assert_allclose(b.numpy(), g["buf_" + n], **TOL)


def check_model(net, g, x, out_key):
    xt = T(x).requires_grad_(True)
    y = net(xt)
    gr = np.random.default_rng(int(g["seed"]) + 2000).standard_normal(tuple(y.shape)).astype(np.float32)
    y.backward(T(gr))
    assert list(net.state_dict().keys()) == [str(s) for s in g["keys"]] or \
        sorted(net.state_dict().keys()) == sorted(str(s) for s in g["keys"])
    np.testing.assert_allclose(y.detach().numpy(), g[out_key], **TOL)
    np.testing.assert_allclose(xt.grad.numpy(), g["grad_x"], rtol=1e-3, atol=1e-4)
    for n, p in net.named_parameters():
        ref_norm = float(g["gnorm_" + n])
        got = p.grad.reshape(-1)
        assert abs(float(got.double().norm()) - ref_norm) <= 1e-3 * ref_norm + 1e-4, n
        np.testing.assert_allclose(got[:16].numpy(), g["ghead_" + n], rtol=2e-3, atol=2e-4, err_msg=n)


@pytest.mark.parametrize("name", ["dgcnnseg_dyn", "dgcnnseg_static", "dgcnnseg_c15_eval", "dgcnnseg_stn",
                                  "dgcnnseg_img"])
def test_dgcnnseg_restatement(name):
    g = load(name)
    seed, cin = int(g["seed"]), int(g["cin"])
    net = ref_cpu.DGCNNSeg(k=8, in_features=cin, num_classes=4, dynamic=bool(g["dynamic"]),
                           spatial_transformer=name.endswith("stn"), image_feat_module=name.endswith("img"))
    fill_state_dict(net, seed).train(bool(g["train"]))
    check_model(net, g, cloud(seed + 1000, 2, cin, 128), "logits")


def test_pointnet_config1():
    g = load("pointnet_c1")
    net = fill_state_dict(ref_cpu.PointNetSeg(3, 4), 501).train()
    check_model(net, g, cloud(1501, 8, 3, 1024), "logits")


@pytest.mark.parametrize("name", ["ae_fold", "ae_deform_static"])
def test_folding_ae_restatement(name):
    g = load(name)
    seed = int(g["seed"])
    net = ref_cpu.DGCNNFoldingNet(k=8, n_embedding=64, n_input_points=2048, decode_mesh=False,
                                  deform=bool(g["deform"]), static=bool(g["static"]))
    fill_state_dict(net, seed).train()
    x = cloud(seed + 1000, 2, 3, 2048)
    np.testing.assert_allclose(net.encoder(T(x)).detach().numpy(), g["code"], **TOL)
    check_model(net, g, x, "recon")


def test_chamfer_restatement():
    g = load("chamfer_s701")
    rng = np.random.default_rng(701)
    a = rng.uniform(-1, 1, (2, 512, 3)).astype(np.float32)
    b = rng.uniform(-1, 1, (2, 384, 3)).astype(np.float32)
    at, bt = T(a).requires_grad_(True), T(b).requires_grad_(True)
    loss = ref_cpu.chamfer(at, bt)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * float(g["loss"])
    np.testing.assert_allclose(at.grad.numpy(), g["grad_a"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(bt.grad.numpy(), g["grad_b"], rtol=1e-4, atol=1e-6)
    # C primitive agrees with the torch restatement, (B,3,N) layout accepted (chamfer_loss.py:10-16)
    d1, _ = c_api.chamfer_nn(a, b)
    d2, _ = c_api.chamfer_nn(b, a)
    assert abs(d1.mean(1).mean() + d2.mean(1).mean() - float(g["loss"])) <= 1e-5 * float(g["loss"])
    assert abs(ref_cpu.chamfer(T(a).transpose(1, 2), T(b).transpose(1, 2)).item() - float(g["loss"])) < 1e-5


def test_nnu_loss_restatement():
    """oracle/ref_cpu.nnu_loss against the reference's NNULoss outputs (losses/nnu_loss.py, losses/dice_loss.py)."""
    g = load("nnu_loss")
    for seed, B, C, N, wt in g["cases"]:
        seed = int(seed)
        lg, lb, w = seg_loss_case(seed, int(B), int(C), int(N), bool(wt), drop_class=(seed == 305))
        x = T(lg).requires_grad_(True)
        crit = ref_cpu.NNULoss(None if w is None else T(w))
        total, parts = crit(x, T(lb))
        total.backward()
        assert abs(total.item() - float(g[f"s{seed}_total"])) < 2e-6
        assert abs(parts["CE"].item() - float(g[f"s{seed}_ce"])) < 2e-6
        assert abs(parts["GDL"].item() - float(g[f"s{seed}_gdl"])) < 2e-6
        gr = g[f"s{seed}_grad"]
        assert np.abs(x.grad.numpy() - gr).max() <= 1e-5 * np.abs(gr).max()
