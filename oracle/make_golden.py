"""Generate the committed golden vectors under tests/golden/ by running the REAL reference.

Runs only in the build container (needs /root/reference); never on the GPU box.  The reference's
arithmetic modules import cleanly once inert placeholders stand in for libraries that are imported
at module top but never touched by the arithmetic (open3d, pytorch3d, thop, ptflops, torchinfo --
SURVEY.md 8c).  Only inputs (as seeds), outputs and gradients are written; no reference source text.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py
"""
import os
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


class _Meta(type):
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Meta(name, (), {"__init__": lambda self, *a, **k: None})


class _Inert(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Meta(name, (), {"__init__": lambda self, *a, **k: None})


def import_reference():
    for name in ["thop", "open3d", "pytorch3d", "pytorch3d.structures", "pytorch3d.transforms",
                 "pytorch3d.loss", "pytorch3d.ops", "ptflops", "torchinfo"]:
        m = _Inert(name)
        m.__path__ = []
        sys.modules[name] = m
    sys.path.insert(0, REF)
    os.chdir(REF)  # shapes/*.npy are opened relative to cwd


def main():
    sys.dont_write_bytecode = True
    import_reference()
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import numpy as np
    import torch
    from golden_util import GOLDEN_DIR, cloud, fill_state_dict

    import models.dgcnn as r_dgcnn
    import models.dgcnn_opensrc as r_open
    import models.folding_net as r_fold
    import models.point_net as r_pn
    import utils.general_utils as r_gu

    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(GOLDEN_DIR, exist_ok=True)

    def save(name, **arrs):
        np.savez_compressed(os.path.join(GOLDEN_DIR, name + ".npz"), **arrs)
        print("wrote", name, len(arrs), "arrays")

    def T(a):
        return torch.from_numpy(np.ascontiguousarray(a))

    # ---- knn (utils/general_utils.py:315-327) and dgcnn_opensrc.knn (:34-40)
    cases = [(101, 2, 3, 256, 20), (102, 2, 3, 256, 16), (103, 2, 3, 256, 40), (104, 2, 64, 256, 20),
             (105, 2, 64, 256, 40), (106, 1, 3, 2048, 20), (107, 1, 15, 512, 20)]
    for seed, B, C, N, k in cases:
        x = cloud(seed, B, C, N)
        out = {"seed": seed, "B": B, "C": C, "N": N, "k": k}
        for sl in (True, False):
            idx, dist = r_gu.knn(T(x), k, self_loop=sl, return_dist=True)
            out[f"idx_self{int(sl)}"] = idx.numpy().astype(np.int16)
            out[f"dist_self{int(sl)}"] = dist.numpy()
        if C > 3:  # first-layer rule: graph over coords only (models/dgcnn.py:26)
            out["idx_coords_self1"] = r_gu.knn(T(x)[:, :3], k, self_loop=True).numpy().astype(np.int16)
        out["idx_open"] = r_open.knn(T(x), k).numpy().astype(np.int16)
        save(f"knn_s{seed}", **out)

    # ---- edge features (models/dgcnn.py:15-36, models/dgcnn_opensrc.py:43-66) + grad
    x = cloud(201, 2, 5, 64)
    idx = r_gu.knn(T(x), 4, self_loop=True)
    xt = T(x).requires_grad_(True)
    e = r_dgcnn.create_neighbor_features(xt, 4, fixed_knn_graph=idx)
    g = np.random.default_rng(2010).standard_normal(tuple(e.shape)).astype(np.float32)
    e.backward(T(g))
    e2 = r_open.get_graph_feature(T(x), k=4, idx=idx)
    save("edge_feat_s201", seed=201, gseed=2010, idx=idx.numpy().astype(np.int16), edge=e.detach().numpy(),
         edge_open=e2.numpy(), grad_x=xt.grad.numpy())

    # ---- EdgeConv, train mode (models/dgcnn.py:212-243)
    for name, seed, cin, couts, k, first, N in [("edgeconv_first", 301, 3, [64, 64], 8, True, 128),
                                                ("edgeconv_feat", 302, 64, [64], 8, False, 128),
                                                ("edgeconv_c15", 303, 15, [64, 64], 20, True, 256)]:
        ec = fill_state_dict(r_dgcnn.EdgeConv(cin, couts, k, first_layer=first), seed)
        ec.train()
        x = cloud(seed + 1000, 2, cin, N)
        xt = T(x).requires_grad_(True)
        y = ec(xt)
        g = np.random.default_rng(seed + 2000).standard_normal(tuple(y.shape)).astype(np.float32)
        y.backward(T(g))
        out = {"seed": seed, "cin": cin, "couts": np.array(couts), "k": k, "first": int(first), "N": N,
               "y": y.detach().numpy(), "grad_x": xt.grad.numpy()}
        for n, p in ec.named_parameters():
            out["grad_" + n] = p.grad.numpy()
        for n, b in ec.named_buffers():
            if "running" in n:
                out["buf_" + n] = b.numpy()
        save(name, **out)

    # ---- DGCNNSeg fwd+bwd (models/dgcnn.py:115-162), dynamic / static / eval
    def grads_digest(model):
        d = {}
        for n, p in model.named_parameters():
            gr = p.grad.reshape(-1)
            d["gnorm_" + n] = np.float64(gr.double().norm().item())
            d["ghead_" + n] = gr[:16].numpy().copy()
        return d

    for name, seed, cin, dyn, train, kw in [
            ("dgcnnseg_dyn", 401, 3, True, True, {}),
            ("dgcnnseg_static", 402, 3, False, True, {}),
            ("dgcnnseg_c15_eval", 403, 15, True, False, {}),
            ("dgcnnseg_stn", 404, 3, True, True, {"spatial_transformer": True}),
            ("dgcnnseg_img", 405, 9, True, True, {"image_feat_module": True})]:
        net = fill_state_dict(r_dgcnn.DGCNNSeg(k=8, in_features=cin, num_classes=4, dynamic=dyn, **kw), seed)
        net.train(train)
        x = cloud(seed + 1000, 2, cin, 128)
        xt = T(x).requires_grad_(True)
        y = net(xt)
        g = np.random.default_rng(seed + 2000).standard_normal(tuple(y.shape)).astype(np.float32)
        y.backward(T(g))
        save(name, seed=seed, cin=cin, dynamic=int(dyn), train=int(train), logits=y.detach().numpy(),
             grad_x=xt.grad.numpy(), keys=np.array(list(net.state_dict().keys())), **grads_digest(net))

    # ---- PointNetSeg, BASELINE config 1 (models/point_net.py:55-100)
    net = fill_state_dict(r_pn.PointNetSeg(3, 4), 501)
    net.train()
    x = cloud(1501, 8, 3, 1024)
    xt = T(x).requires_grad_(True)
    y = net(xt)
    g = np.random.default_rng(2501).standard_normal(tuple(y.shape)).astype(np.float32)
    y.backward(T(g))
    save("pointnet_c1", seed=501, logits=y.detach().numpy(), grad_x=xt.grad.numpy(),
         keys=np.array(list(net.state_dict().keys())), **grads_digest(net))

    # ---- PC-AE (models/folding_net.py): encoder + folding / deforming decoders, point output
    for name, seed, deform, static in [("ae_fold", 601, False, False), ("ae_deform_static", 602, True, True)]:
        net = fill_state_dict(r_fold.DGCNNFoldingNet(k=8, n_embedding=64, shape_type="plane", n_input_points=2048,
                                                     decode_mesh=False, deform=deform, static=static), seed)
        net.train()
        x = cloud(seed + 1000, 2, 3, 2048)
        xt = T(x).requires_grad_(True)
        y, h = net(xt, return_hidden=True)
        g = np.random.default_rng(seed + 2000).standard_normal(tuple(y.shape)).astype(np.float32)
        y.backward(T(g))
        save(name, seed=seed, deform=int(deform), static=int(static), recon=y.detach().numpy(),
             code=h.detach().numpy(), grad_x=xt.grad.numpy(), keys=np.array(list(net.state_dict().keys())),
             **grads_digest(net))

    # ---- Chamfer restated with the reference's own pairwise_dist2 (pytorch3d absent: the
    #      third-party boundary itself stays unpinned)
    rng = np.random.default_rng(701)
    a = rng.uniform(-1, 1, (2, 512, 3)).astype(np.float32)
    b = rng.uniform(-1, 1, (2, 384, 3)).astype(np.float32)
    at, bt = T(a).requires_grad_(True), T(b).requires_grad_(True)
    d = r_gu.pairwise_dist2(at, bt)
    loss = d.min(2).values.mean(1).mean() + d.min(1).values.mean(1).mean()
    loss.backward()
    save("chamfer_s701", seed=701, loss=np.float64(loss.item()), grad_a=at.grad.numpy(), grad_b=bt.grad.numpy())


if __name__ == "__main__":
    main()
