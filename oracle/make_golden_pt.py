"""Golden vectors for the PointTransformer path, `predict_full_pointcloud` and `farthest_point_sampling`,
produced by running the REAL reference code (build container only; never on the GPU box).

    PYTHONDONTWRITEBYTECODE=1 python -O oracle/make_golden_pt.py

What is real and what is substituted:

* `/root/reference/models/pointtransformer/seg_model.py:17-231` and `pointops.py:100-123,198-215` are pure torch
  around exactly two native calls, `pointops_cuda.knnquery_cuda` (`pointops.py:59`) and
  `pointops_cuda.furthestsampling_cuda` (`pointops.py:35`).  `pointops_cuda` (POSTECH-CVLab/point-transformer
  `lib/pointops`, unversioned, absent) is an inert placeholder; the two module attributes `pointops.knnquery` /
  `pointops.furthestsampling` (`pointops.py:39,62`) are replaced by CPU functions on `oracle/fsg_oracle.c`
  (segment kNN ascending by (d2, index), short segments padded with (segment start, 1e10); FPS from the first
  row of a segment).  Everything else -- every BatchNorm view, the softmax axis, the share_planes grouping, the
  TransitionUp head, the interpolation weights -- is the reference's own code.  The third-party kernels
  themselves stay unpinned; their outputs here are integer index lists the tests compare exactly.
* `-O` strips the reference's `assert xyz.device.type == 'cuda'` guards (`pointops.py:106,204`): device checks,
  no arithmetic.
* `predict_full_pointcloud` (`models/point_seg_net.py:21-48`) runs on the reference's DGCNNSeg; the rows
  `torch.randperm` returned are recorded so the test can replay them on any device.
* `farthest_point_sampling` (`dseg_ae_regularization.py:30-43`): the module imports after placeholders for
  SimpleITK, batchgenerators, skimage, cv2, imageio, pyamg, pycpd (none touched by the function).

Only seeds, index lists and numeric outputs are written; no reference source text.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import make_golden as mg  # noqa: E402

EXTRA_PLACEHOLDERS = ["pointops_cuda", "SimpleITK", "batchgenerators", "batchgenerators.transforms",
                      "batchgenerators.transforms.abstract_transforms", "batchgenerators.transforms.spatial_transforms",
                      "skimage", "skimage.color", "skimage.morphology", "skimage.measure", "cv2", "imageio", "pyamg",
                      "pycpd"]

PT_LAYER_CASES = [  # name, planes, nsample, segment sizes (one shorter than nsample), train
    ("pt_layer_c32", 32, 8, (70, 5, 53), True),
    ("pt_layer_c64", 64, 16, (61, 9, 40), True),
    ("pt_layer_c128", 128, 16, (50, 16, 30), True),
    ("pt_layer_c256", 256, 16, (33, 17, 3), True),
    ("pt_layer_c512", 512, 16, (8, 20, 12), True),
    ("pt_layer_c64_eval", 64, 16, (64, 23, 7), False),
]
FULL_LIMIT = 20000  # parameter gradients above this size are stored as (norm, first 16 entries)


def packed(seed, sizes, c):
    """same generator as tests/test_gpu_parity.py::packed"""
    import numpy as np
    rng = np.random.default_rng(seed)
    n = int(sum(sizes))
    xyz = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    feat = rng.standard_normal((n, c)).astype(np.float32)
    off = np.cumsum(np.asarray(sizes)).astype(np.int32)
    return xyz, feat, off


def main():
    if __debug__:
        raise SystemExit("run with python -O (the reference asserts device.type == 'cuda')")
    sys.dont_write_bytecode = True
    mg.import_reference()
    for name in EXTRA_PLACEHOLDERS:
        m = mg._Inert(name)
        m.__path__ = []
        sys.modules[name] = m
    import numpy as np
    import torch
    from golden_util import GOLDEN_DIR, cloud, fill_state_dict
    from oracle import c_api

    import models.pointtransformer.pointops as r_po
    import models.pointtransformer.seg_model as r_pt
    import models.dgcnn as r_dgcnn

    torch.set_num_threads(8)

    def knnquery(nsample, xyz, new_xyz, offset, new_offset):
        new_xyz = xyz if new_xyz is None else new_xyz
        idx, d2 = c_api.knn_segment(xyz.detach().numpy(), new_xyz.detach().numpy(), offset.numpy(),
                                    new_offset.numpy(), nsample)
        return torch.from_numpy(idx), torch.sqrt(torch.from_numpy(d2))   # pointops.py:60

    def furthestsampling(xyz, offset, new_offset):
        return torch.from_numpy(c_api.fps(xyz.detach().numpy(), offset.numpy(), new_offset.numpy()))

    r_po.knnquery = knnquery
    r_po.furthestsampling = furthestsampling

    def save(name, **arrs):
        np.savez_compressed(os.path.join(GOLDEN_DIR, name + ".npz"), **arrs)
        print("wrote", name, len(arrs), "arrays",
              os.path.getsize(os.path.join(GOLDEN_DIR, name + ".npz")) // 1024, "KiB")

    def T(a):
        return torch.from_numpy(np.ascontiguousarray(a))

    def pack_grads(model, limit=FULL_LIMIT):
        d = {}
        for n, p in model.named_parameters():
            gr = p.grad.reshape(-1)
            if gr.numel() <= limit:
                d["grad_" + n] = gr.numpy().copy().reshape(tuple(p.shape))
            d["gnorm_" + n] = np.float64(gr.double().norm().item())
            d["ghead_" + n] = gr[:16].numpy().copy()
        for n, b in model.named_buffers():
            if "running" in n:
                if b.numel() <= limit:
                    d["buf_" + n] = b.numpy().copy()
                d["bnorm_" + n] = np.float64(b.double().norm().item())
        return d

    # ---- PointTransformerLayer (seg_model.py:17-53)
    for name, c, ns, sizes, train in PT_LAYER_CASES:
        lay = fill_state_dict(r_pt.PointTransformerLayer(c, c, 8, ns), 811 + c).train(train)
        xyz, feat, off = packed(900 + c + ns, sizes, c)
        p, x = T(xyz).requires_grad_(True), T(feat).requires_grad_(True)
        y = lay([p, x, T(off)])
        g = np.random.default_rng(5).standard_normal(tuple(y.shape)).astype(np.float32)
        y.backward(T(g))
        save(name, c=c, ns=ns, sizes=np.array(sizes), train=int(train), y=y.detach().numpy(),
             grad_x=x.grad.numpy(), grad_p=p.grad.numpy(), **pack_grads(lay))

    # ---- PointTransformerBlock (seg_model.py:121-142)
    blk = fill_state_dict(r_pt.PointTransformerBlock(64, 64, 8, 16), 821).train()
    xyz, feat, off = packed(1821, (90, 11, 60), 64)
    x = T(feat).requires_grad_(True)
    _, y, _ = blk([T(xyz), x, T(off)])
    g = np.random.default_rng(6).standard_normal(tuple(y.shape)).astype(np.float32)
    y.backward(T(g))
    save("pt_block_c64", sizes=np.array((90, 11, 60)), y=y.detach().numpy(), grad_x=x.grad.numpy(), **pack_grads(blk))

    # ---- TransitionDown (seg_model.py:56-84): stride 1 and stride 4 (FPS + grouped Linear/BN/ReLU/max)
    for name, seed, cin, cout, stride, ns, sizes in [("pt_td_s1", 831, 6, 32, 1, 8, (100, 37)),
                                                     ("pt_td_s4", 832, 32, 64, 4, 16, (200, 40, 131))]:
        td = fill_state_dict(r_pt.TransitionDown(cin, cout, stride, ns), seed).train()
        xyz, feat, off = packed(seed + 1000, sizes, cin)
        x = T(feat).requires_grad_(True)
        n_p, y, n_o = td([T(xyz), x, T(off)])
        g = np.random.default_rng(seed + 2000).standard_normal(tuple(y.shape)).astype(np.float32)
        y.backward(T(g))
        save(name, seed=seed, cin=cin, cout=cout, stride=stride, ns=ns, sizes=np.array(sizes),
             new_p=n_p.detach().numpy(), new_o=n_o.numpy(), y=y.detach().numpy(), grad_x=x.grad.numpy(),
             **pack_grads(td))

    # ---- TransitionUp (seg_model.py:87-118): head branch (per-cloud mean feature) and the two-level branch
    tu = fill_state_dict(r_pt.TransitionUp(64, None), 841).train()
    xyz, feat, off = packed(1841, (8, 8, 5), 64)
    x = T(feat).requires_grad_(True)
    y = tu([T(xyz), x, T(off)])
    g = np.random.default_rng(2841).standard_normal(tuple(y.shape)).astype(np.float32)
    y.backward(T(g))
    save("pt_tu_head", sizes=np.array((8, 8, 5)), y=y.detach().numpy(), grad_x=x.grad.numpy(), **pack_grads(tu))

    tu = fill_state_dict(r_pt.TransitionUp(64, 32), 842).train()
    xyz1, feat1, off1 = packed(1842, (120, 33, 64), 32)   # fine level
    xyz2, feat2, off2 = packed(1843, (30, 2, 16), 64)     # coarse level (a segment with fewer than 3 points)
    x1, x2 = T(feat1).requires_grad_(True), T(feat2).requires_grad_(True)
    y = tu([T(xyz1), x1, T(off1)], [T(xyz2), x2, T(off2)])
    g = np.random.default_rng(2842).standard_normal(tuple(y.shape)).astype(np.float32)
    y.backward(T(g))
    save("pt_tu", sizes1=np.array((120, 33, 64)), sizes2=np.array((30, 2, 16)), y=y.detach().numpy(),
         grad_x1=x1.grad.numpy(), grad_x2=x2.grad.numpy(), **pack_grads(tu))

    # ---- interpolation (pointops.py:198-215)
    f2 = T(feat2).requires_grad_(True)
    y = r_po.interpolation(T(xyz2), T(xyz1), f2, T(off2), T(off1))
    g = np.random.default_rng(2850).standard_normal(tuple(y.shape)).astype(np.float32)
    y.backward(T(g))
    save("pt_interp", y=y.detach().numpy(), grad_feat=f2.grad.numpy())

    # ---- queryandgroup (pointops.py:100-123), both use_xyz settings, cross-set query
    q1 = r_po.queryandgroup(8, T(xyz1), T(xyz1), T(feat1), None, T(off1), T(off1), use_xyz=True)
    q2 = r_po.queryandgroup(4, T(xyz1), T(xyz2), T(feat1), None, T(off1), T(off2), use_xyz=False)
    save("pt_group", self_xyz=q1.numpy(), cross=q2.numpy())

    # ---- whole model (seg_model.py:145-231), BASELINE config 3 cloud size: level 5 has 8 points < nsample 16
    net = fill_state_dict(r_pt.PointTransformerCompatibility(6, 4), 801).train()
    x = cloud(1801, 2, 6, 2048)
    xt = T(x).requires_grad_(True)
    y = net(xt)
    g = np.random.default_rng(2801).standard_normal(tuple(y.shape)).astype(np.float32)
    y.backward(T(g))
    save("pt_compat_c6", seed=801, logits=y.detach().numpy(), grad_x=xt.grad.numpy(),
         keys=np.array(list(net.state_dict().keys())), **pack_grads(net, limit=0))

    net = fill_state_dict(r_pt.PointTransformerCompatibility(3, 4), 802).eval()   # coords only (self.c == 3), eval
    x = cloud(1802, 2, 3, 1024)
    with torch.no_grad():
        y = net(T(x))
    save("pt_compat_c3_eval", seed=802, logits=y.numpy())

    # ---- DGCNNReg (models/dgcnn.py:165-209): regression head on the global feature, (B, out, 1); 4 clouds so that the
    #      head's train-mode BatchNorm sees more than two rows
    net = fill_state_dict(r_dgcnn.DGCNNReg(k=8, in_features=3, num_classes=6), 871).train()
    x = cloud(1871, 4, 3, 128)
    xt = T(x).requires_grad_(True)
    y = net(xt)
    g = np.random.default_rng(2871).standard_normal(tuple(y.shape)).astype(np.float32)
    y.backward(T(g))
    save("dgcnnreg", seed=871, out=y.detach().numpy(), grad_x=xt.grad.numpy(),
         keys=np.array(list(net.state_dict().keys())), **pack_grads(net, limit=0))

    # ---- predict_full_pointcloud (models/point_seg_net.py:21-48) on the reference's DGCNNSeg, eval mode as in
    #      train.py:test; the randperm rows are recorded for replay
    net = fill_state_dict(r_dgcnn.DGCNNSeg(k=8, in_features=3, num_classes=4), 851).eval()
    pc = cloud(1851, 1, 3, 1500)
    rows = []
    real_randperm = torch.randperm

    def recording_randperm(*a, **kw):
        r = real_randperm(*a, **kw)
        rows.append(r.numpy().astype(np.int32))
        return r

    torch.manual_seed(851)
    torch.randperm = recording_randperm
    try:
        with torch.no_grad():
            out = net.predict_full_pointcloud(T(pc), sample_points=256, n_runs_min=10)
    finally:
        torch.randperm = real_randperm
    save("predict_full_s851", seed=851, n_perm=len(rows), probs=out.numpy(),
         **{f"perm{i}": r for i, r in enumerate(rows)})

    # ---- farthest_point_sampling (dseg_ae_regularization.py:30-43); its random start index is recorded
    import dseg_ae_regularization as r_reg
    out = {}
    for i, (seed, n, m) in enumerate([(861, 2000, 256), (862, 777, 100), (863, 50, 50), (864, 30, 40)]):
        pts = np.random.default_rng(seed).uniform(-1, 1, (1, n, 3)).astype(np.float32)
        torch.manual_seed(seed)
        sub, ind = r_reg.farthest_point_sampling(T(pts), m)
        out[f"case{i}"] = np.array([seed, n, m])
        out[f"ind{i}"] = ind.numpy().astype(np.int32)
        out[f"pts{i}"] = sub.numpy()
    save("fps_torch", n_cases=4, **out)


if __name__ == "__main__":
    main()
