"""Pure-PyTorch CPU restatement of the reference's point-cloud dataflow.  TEST INFRASTRUCTURE ONLY.

This is the "reference-equivalent CPU path": it materialises the (B,N,N) distance matrix, the
(B,2C,N,k) edge tensor and every intermediate exactly the way the reference does, using stock ATen
ops on the CPU.  It is pinned against golden vectors generated from the real reference
(oracle/make_golden.py -> tests/golden/*.npz) and is in turn the checker for the HIP path and the
timed ``cpu_baseline`` of bench.py.  Module attribute names follow the reference so that a
``state_dict`` moves freely between reference, oracle and product.

Citations are relative to /root/reference.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import c_api


# ----------------------------------------------------------------------------- primitives
KNN_BACKEND = "torch"  # "torch": bmm + topk exactly like the reference (tie order = ATen's);
#                        "c": oracle/fsg_oracle.c (the build's exact arithmetic + lowest-index tie rule).
# Both are pinned against the reference's golden kNN vectors; model-level GPU parity uses "c" so that a
# near-tie at the k-th neighbour cannot flip between the checker and the HIP path.


def _knn_c(x, k, c_knn=None, fix_diag=True, drop_first=False, return_dist=False):
    idx, dist = c_api.knn_dense(x.detach().numpy(), k, c_knn=c_knn, fix_diag=fix_diag, drop_first=drop_first)
    idx = torch.from_numpy(idx.astype(np.int64))
    return (idx, torch.from_numpy(dist)) if return_dist else idx


def pairwise_dist(x):
    """utils/general_utils.py:43-53 -- x: (B,N,C) -> (B,N,N), diagonal forced to zero."""
    sq = x.pow(2).sum(2, keepdim=True)
    d = sq - 2.0 * torch.bmm(x, x.transpose(1, 2)) + sq.transpose(1, 2)
    ar = torch.arange(x.shape[1])
    d[:, ar, ar] = 0
    return d


def pairwise_dist2(x, y):
    """utils/general_utils.py:56-67."""
    return (x.pow(2).sum(2, keepdim=True) - 2.0 * torch.bmm(x, y.transpose(1, 2))
            + y.pow(2).sum(2, keepdim=True).transpose(1, 2))


def knn(x, k, self_loop=False, return_dist=False):
    """utils/general_utils.py:315-327 -- x: (B,C,N) -> idx (B,N,k) int64 [, dist]."""
    if KNN_BACKEND == "c":
        return _knn_c(x, k, drop_first=not self_loop, return_dist=return_dist)
    skip = 0 if self_loop else 1
    top, idx = pairwise_dist(x.transpose(1, 2)).topk(k + skip, dim=-1, largest=False)
    top, idx = top[..., skip:], idx[..., skip:]
    return (idx, top) if return_dist else idx


def knn_opensrc(x, k):
    """models/dgcnn_opensrc.py:34-40."""
    if KNN_BACKEND == "c":
        return _knn_c(x, k, fix_diag=False)
    inner = -2 * torch.matmul(x.transpose(1, 2), x)
    sq = x.pow(2).sum(1, keepdim=True)
    return (-sq - inner - sq.transpose(1, 2)).topk(k, dim=-1)[1]


def edge_features(x, idx):
    """models/dgcnn.py:31-36 and models/dgcnn_opensrc.py:43-66: cat(x_j - x_i, x_i) -> (B,2C,N,k)."""
    B, C, N = x.shape
    k = idx.shape[-1]
    nb = torch.gather(x, 2, idx.reshape(B, 1, N * k).expand(B, C, N * k)).view(B, C, N, k)
    ctr = x.unsqueeze(-1).expand(B, C, N, k)
    return torch.cat([nb - ctr, ctr], 1)


def chamfer(pred, target):
    """losses/chamfer_loss.py:9-20 with pytorch3d defaults (mean over points, both directions
    summed, mean over batch; train_pc_ae.py:85).  Direct (x - y)^2 form."""
    if pred.shape[1] == 3:
        pred = pred.transpose(1, 2)
    if target.shape[1] == 3:
        target = target.transpose(1, 2)
    assert pred.shape[0] == target.shape[0] and pred.shape[2] == target.shape[2]
    d = (pred.unsqueeze(2) - target.unsqueeze(1)).pow(2).sum(-1)
    return d.min(2).values.mean(1).mean() + d.min(1).values.mean(1).mean()


class ChamferLoss(nn.Module):
    def forward(self, prediction, target):
        return chamfer(prediction, target)


def init_weights(m):
    """utils/model_utils.py:11-15."""
    if isinstance(m, (nn.modules.conv._ConvNd, nn.Linear)):
        nn.init.xavier_normal_(m.weight)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0.0)


# ----------------------------------------------------------------------------- DGCNN
class Block(nn.Module):
    """models/dgcnn.py:282-323 -- 1x1 conv [+ BN + LeakyReLU]; `layers` naming as the reference."""

    def __init__(self, cin, cout, dim=2, last=False, slope=0.2):
        super().__init__()
        conv = {1: nn.Conv1d, 2: nn.Conv2d}[dim]
        bn = {1: nn.BatchNorm1d, 2: nn.BatchNorm2d}[dim]
        mods = [conv(cin, cout, 1, bias=last)]
        if not last:
            mods += [bn(cout), nn.LeakyReLU(slope)]
        self.layers = nn.ModuleList(mods)

    def forward(self, x):
        for m in self.layers:
            x = m(x)
        return x


class EdgeConv(nn.Module):
    """models/dgcnn.py:212-243."""

    def __init__(self, cin, couts, k, first_layer=False):
        super().__init__()
        self.k, self.first_layer = k, first_layer
        chans = [2 * cin] + list(couts)
        self.shared_mlp = nn.ModuleList(Block(a, b) for a, b in zip(chans[:-1], chans[1:]))

    def forward(self, x, graph=None):
        if graph is None:
            graph = knn(x[:, :3] if self.first_layer else x, self.k, self_loop=True)
        e = edge_features(x, graph)
        for blk in self.shared_mlp:
            e = blk(e)
        return e.max(-1)[0]


class SpatialTransformer(nn.Module):
    """models/dgcnn.py:246-279."""

    def __init__(self, k):
        super().__init__()
        self.ec = EdgeConv(3, [64, 128], k)
        self.shared_fc = Block(128, 1024, dim=1)
        self.mlp = nn.Sequential(nn.Linear(1024, 512), nn.BatchNorm1d(512), nn.LeakyReLU(0.2),
                                 nn.Linear(512, 256), nn.BatchNorm1d(256), nn.LeakyReLU(0.2))
        self.transform = nn.Linear(256, 9)

    def forward(self, x, graph=None):
        xyz = x[:, :3].clone()
        t = self.shared_fc(self.ec(xyz, graph)).max(-1)[0]
        t = self.transform(self.mlp(t)).view(-1, 3, 3)
        xyz = torch.bmm(xyz.transpose(1, 2), t).transpose(1, 2)
        return torch.cat([xyz, x[:, 3:]], 1)


class ImageFeatures(nn.Module):
    """models/dgcnn.py:326-343 (ConvBlock default slope 1e-2, dim=1)."""

    def __init__(self, cin=6, couts=(6, 12)):
        super().__init__()
        chans = [cin] + list(couts)
        self.layers = nn.ModuleList(Block(a, b, dim=1, slope=1e-2) for a, b in zip(chans[:-1], chans[1:]))

    def forward(self, x):
        f = x[:, 3:].clone()
        for blk in self.layers:
            f = blk(f)
        return torch.cat([x[:, :3], f], 1)


class DGCNNSeg(nn.Module):
    """models/dgcnn.py:61-162."""

    def __init__(self, k, in_features, num_classes, spatial_transformer=False, dynamic=True,
                 image_feat_module=False):
        super().__init__()
        self.k, self.dynamic, self.num_classes = k, dynamic, num_classes
        self.image_feature_module = ImageFeatures(in_features - 3) if image_feat_module else None
        cin = 15 if image_feat_module else in_features
        self.spatial_transformer = SpatialTransformer(k) if spatial_transformer else None
        self.ec1 = EdgeConv(cin, [64, 64], k, first_layer=True)
        self.ec2 = EdgeConv(64, [64], k)
        self.ec3 = EdgeConv(64, [64], k)
        self.global_feature = nn.Sequential(Block(192, 1024, dim=1), nn.AdaptiveMaxPool1d(1))
        self.segmentation = nn.Sequential(Block(1216, 256, dim=1), Block(256, 256, dim=1),
                                          Block(256, 128, dim=1),
                                          Block(128, num_classes, dim=1, last=True))
        self.apply(init_weights)
        if self.spatial_transformer is not None:
            nn.init.constant_(self.spatial_transformer.transform.weight, 0)
            nn.init.eye_(self.spatial_transformer.transform.bias.view(3, 3))

    def forward(self, x):
        graph = None if self.dynamic else knn(x[:, :3], self.k, self_loop=False)
        if self.image_feature_module is not None:
            x = self.image_feature_module(x)
        if self.spatial_transformer is not None:
            x = self.spatial_transformer(x)
        x1 = self.ec1(x, graph)
        x2 = self.ec2(x1, graph)
        x3 = self.ec3(x2, graph)
        ml = torch.cat([x1, x2, x3], 1)
        g = self.global_feature(ml)
        return self.segmentation(torch.cat([ml, g.expand(-1, -1, ml.shape[-1])], 1))


class DGCNNReg(nn.Module):
    """models/dgcnn.py:165-209: four one-layer EdgeConvs 64/64/128/256, global max feature, regression head (B,out,1)."""

    def __init__(self, k, in_features, num_classes, dynamic=True):
        super().__init__()
        self.k, self.dynamic, self.num_classes = k, dynamic, num_classes
        self.ec1 = EdgeConv(in_features, [64], k, first_layer=True)
        self.ec2 = EdgeConv(64, [64], k)
        self.ec3 = EdgeConv(64, [128], k)
        self.ec4 = EdgeConv(128, [256], k)
        self.global_feature = nn.Sequential(Block(512, 1024, dim=1), nn.AdaptiveMaxPool1d(1))
        self.regression = nn.Sequential(Block(1024, 512, dim=1), Block(512, 256, dim=1),
                                        Block(256, num_classes, dim=1, last=True))
        self.apply(init_weights)

    def forward(self, x):
        graph = None if self.dynamic else knn(x[:, :3], self.k, self_loop=False)
        feats = []
        for ec in (self.ec1, self.ec2, self.ec3, self.ec4):
            x = ec(x, graph)
            feats.append(x)
        return self.regression(self.global_feature(torch.cat(feats, 1)))


# ----------------------------------------------------------------------------- PointNet (config 1)
class MLPBlock(nn.Module):
    """models/point_net.py:11-30 (LeakyReLU default slope 0.01)."""

    def __init__(self, cin, widths):
        super().__init__()
        mods, prev = [], cin
        for w in widths:
            mods += [nn.Conv1d(prev, w, 1, bias=False), nn.BatchNorm1d(w), nn.LeakyReLU()]
            prev = w
        self.layers = nn.ModuleList(mods)

    def forward(self, x):
        for m in self.layers:
            x = m(x)
        return x


class PointNetSeg(nn.Module):
    """models/point_net.py:55-100, default path (no T-Nets)."""

    def __init__(self, in_features, num_classes, **_):
        super().__init__()
        self.local_features = MLPBlock(in_features, [64, 64])
        self.global_features = nn.Sequential(MLPBlock(64, [64, 128, 1024]), nn.AdaptiveMaxPool1d(1))
        self.seg_branch = nn.Sequential(MLPBlock(1088, [256, 128, 64, 64]),
                                        nn.Conv1d(64, num_classes, 1, bias=True))
        self.apply(init_weights)

    def forward(self, x):
        loc = self.local_features(x)
        glob = self.global_features(loc)
        return self.seg_branch(torch.cat([loc, glob.expand(-1, -1, loc.shape[-1])], 1))


# ----------------------------------------------------------------------------- PC-AE (FoldingNet)
def plane_grid_45():
    """shapes/shape_constructor.py:35-40: itertools.product of two 45-step linspaces in [-0.3,0.3]."""
    a = np.linspace(-0.3, 0.3, 45)
    return np.stack(np.meshgrid(a, a, indexing="ij"), -1).reshape(-1, 2)


def plane_grid_mesh(n, lo=-0.3, hi=0.3):
    """shapes/shape_constructor.py:8-14 (vertices only)."""
    s = int(math.sqrt(n))
    gx, gy = torch.meshgrid(torch.linspace(lo, hi, s), torch.linspace(lo, hi, s), indexing="ij")
    return torch.stack([gx.reshape(-1), gy.reshape(-1)], 1)


class ClsEncoder(nn.Module):
    """models/folding_net.py:83-141 (bnX registered twice, as in the reference's state_dict)."""

    def __init__(self, k, n_embedding, static=False):
        super().__init__()
        self.k, self.static = k, static
        widths = [(6, 64), (128, 64), (128, 128), (256, 256)]
        for i, (a, b) in enumerate(widths, 1):
            bn = nn.BatchNorm2d(b)
            setattr(self, f"bn{i}", bn)
            setattr(self, f"conv{i}", nn.Sequential(nn.Conv2d(a, b, 1, bias=False), bn, nn.LeakyReLU(0.2)))
        self.bn5 = nn.BatchNorm1d(n_embedding)
        self.conv5 = nn.Sequential(nn.Conv1d(512, n_embedding, 1, bias=False), self.bn5, nn.LeakyReLU(0.2))

    def forward(self, x):
        graph = knn_opensrc(x[:, :3], self.k) if self.static else None
        feats = []
        for i in range(1, 5):
            idx = graph if graph is not None else knn_opensrc(x, self.k)
            x = getattr(self, f"conv{i}")(edge_features(x, idx)).max(-1)[0]
            feats.append(x)
        return self.conv5(torch.cat(feats, 1)).max(-1)[0].unsqueeze(1)


class FoldingDecoder(nn.Module):
    """models/folding_net.py:186-228, plane shape, decode_mesh=False geometry; `grid` is (m,2)."""

    def __init__(self, n_embedding, grid):
        super().__init__()
        self.register_buffer("grid", grid.float(), persistent=False)

        def fold(cin):
            return nn.Sequential(nn.Conv1d(cin, n_embedding, 1), nn.ReLU(), nn.Conv1d(n_embedding, n_embedding, 1),
                                 nn.ReLU(), nn.Conv1d(n_embedding, 3, 1))
        self.folding1, self.folding2 = fold(n_embedding + 2), fold(n_embedding + 3)

    def forward(self, code):
        m = self.grid.shape[0]
        z = code.transpose(1, 2).expand(-1, -1, m)
        g = self.grid.t().unsqueeze(0).expand(z.shape[0], -1, -1)
        f1 = self.folding1(torch.cat([z, g], 1))
        return self.folding2(torch.cat([z, f1], 1))


class DeformingDecoder(nn.Module):
    """models/folding_net.py:231-288, plane shape, point output."""

    def __init__(self, n_embedding, grid, depth=2):
        super().__init__()
        self.register_buffer("grid", torch.cat([grid.float(), torch.zeros(grid.shape[0], 1)], 1), persistent=False)

        def deform():
            return nn.Sequential(Block(n_embedding + 3, n_embedding, dim=1), Block(n_embedding, n_embedding, dim=1),
                                 Block(n_embedding, 3, dim=1, last=True))
        if depth == 2:
            self.deforming1, self.deforming2 = deform(), deform()
            self.deforming_layers = nn.ModuleList([self.deforming1, self.deforming2])
        else:
            self.deforming_layers = nn.ModuleList(deform() for _ in range(depth))

    def forward(self, code):
        m = self.grid.shape[0]
        z = code.transpose(1, 2).expand(-1, -1, m)
        p = self.grid.t().unsqueeze(0).expand(z.shape[0], -1, -1)
        for layer in self.deforming_layers:
            p = p + layer(torch.cat([z, p], 1))
        return p


class DGCNNFoldingNet(nn.Module):
    """models/folding_net.py:42-63 (point output).  mesh_grid=True takes the vertex grid of
    get_plane_mesh(m) (decode_mesh geometry, needed for N != 2048), else the fixed 45x45 plane."""

    def __init__(self, k, n_embedding, shape_type="plane", n_input_points=1024, decode_mesh=False,
                 deform=False, static=False, dec_depth=2):
        super().__init__()
        assert shape_type == "plane"
        m = int(round(math.sqrt(n_input_points))) ** 2
        grid = plane_grid_mesh(m) if decode_mesh else torch.from_numpy(plane_grid_45())
        self.encoder = ClsEncoder(k, n_embedding, static)
        self.decoder = DeformingDecoder(n_embedding, grid, dec_depth) if deform else FoldingDecoder(n_embedding, grid)

    def forward(self, x):
        return self.decoder(self.encoder(x))


# ----------------------------------------------------------------------------- PointTransformer
def knnquery(nsample, xyz, new_xyz, offset, new_offset):
    """pointops.py:42-62 -> (idx int32 (m,ns), sqrt(dist2))."""
    new_xyz = xyz if new_xyz is None else new_xyz
    idx, d2 = c_api.knn_segment(xyz.detach().numpy(), new_xyz.detach().numpy(), offset.numpy(),
                                new_offset.numpy(), nsample)
    return torch.from_numpy(idx), torch.from_numpy(d2).sqrt()


def furthestsampling(xyz, offset, new_offset):
    """pointops.py:16-39."""
    return torch.from_numpy(c_api.fps(xyz.detach().numpy(), offset.numpy(), new_offset.numpy()))


def queryandgroup(nsample, xyz, new_xyz, feat, idx, offset, new_offset, use_xyz=True):
    """pointops.py:100-123."""
    new_xyz = xyz if new_xyz is None else new_xyz
    if idx is None:
        idx, _ = knnquery(nsample, xyz, new_xyz, offset, new_offset)
    m = new_xyz.shape[0]
    flat = idx.reshape(-1).long()
    g_xyz = xyz[flat].view(m, nsample, 3) - new_xyz.unsqueeze(1)
    g_feat = feat[flat].view(m, nsample, -1)
    return torch.cat([g_xyz, g_feat], -1) if use_xyz else g_feat


def interpolation(xyz, new_xyz, feat, offset, new_offset, k=3):
    """pointops.py:198-215."""
    idx, dist = knnquery(k, xyz, new_xyz, offset, new_offset)
    w = 1.0 / (dist + 1e-8)
    w = w / w.sum(1, keepdim=True)
    out = torch.zeros(new_xyz.shape[0], feat.shape[1])
    for i in range(k):
        out = out + feat[idx[:, i].long()] * w[:, i:i + 1]
    return out


class PTLayer(nn.Module):
    """models/pointtransformer/seg_model.py:17-53."""

    def __init__(self, cin, cout, share_planes=8, nsample=16):
        super().__init__()
        self.mid_planes = self.out_planes = cout
        self.share_planes, self.nsample = share_planes, nsample
        self.linear_q, self.linear_k, self.linear_v = nn.Linear(cin, cout), nn.Linear(cin, cout), nn.Linear(cin, cout)
        self.linear_p = nn.Sequential(nn.Linear(3, 3), nn.BatchNorm1d(3), nn.ReLU(inplace=True), nn.Linear(3, cout))
        self.linear_w = nn.Sequential(nn.BatchNorm1d(cout), nn.ReLU(inplace=True), nn.Linear(cout, cout // share_planes),
                                      nn.BatchNorm1d(cout // share_planes), nn.ReLU(inplace=True),
                                      nn.Linear(cout // share_planes, cout // share_planes))

    def forward(self, pxo):
        p, x, o = pxo
        q, kf, v = self.linear_q(x), self.linear_k(x), self.linear_v(x)
        idx, _ = knnquery(self.nsample, p, p, o, o)
        gk = queryandgroup(self.nsample, p, p, kf, idx, o, o, use_xyz=True)
        gv = queryandgroup(self.nsample, p, p, v, idx, o, o, use_xyz=False)
        pr, gk = gk[:, :, :3], gk[:, :, 3:]
        for i, layer in enumerate(self.linear_p):
            pr = layer(pr.transpose(1, 2).contiguous()).transpose(1, 2).contiguous() if i == 1 else layer(pr)
        w = gk - q.unsqueeze(1) + pr
        for i, layer in enumerate(self.linear_w):
            w = layer(w.transpose(1, 2).contiguous()).transpose(1, 2).contiguous() if i % 3 == 0 else layer(w)
        w = torch.softmax(w, 1)
        n, ns, c = gv.shape
        s = self.share_planes
        return ((gv + pr).view(n, ns, s, c // s) * w.unsqueeze(2)).sum(1).view(n, c)


class TransitionDown(nn.Module):
    """seg_model.py:56-84."""

    def __init__(self, cin, cout, stride=1, nsample=16):
        super().__init__()
        self.stride, self.nsample = stride, nsample
        self.linear = nn.Linear((3 + cin) if stride != 1 else cin, cout, bias=False)
        if stride != 1:
            self.pool = nn.MaxPool1d(nsample)
        self.bn = nn.BatchNorm1d(cout)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, pxo):
        p, x, o = pxo
        if self.stride == 1:
            return [p, self.relu(self.bn(self.linear(x))), o]
        counts = torch.diff(o, prepend=o.new_zeros(1)) // self.stride
        n_o = torch.cumsum(counts, 0).to(o.dtype)
        idx = furthestsampling(p, o, n_o)
        n_p = p[idx.long()]
        g = queryandgroup(self.nsample, p, n_p, x, None, o, n_o, use_xyz=True)
        x = self.relu(self.bn(self.linear(g).transpose(1, 2).contiguous()))
        return [n_p, self.pool(x).squeeze(-1), n_o]


class TransitionUp(nn.Module):
    """seg_model.py:87-118."""

    def __init__(self, cin, cout=None):
        super().__init__()
        if cout is None:
            self.linear1 = nn.Sequential(nn.Linear(2 * cin, cin), nn.BatchNorm1d(cin), nn.ReLU(inplace=True))
            self.linear2 = nn.Sequential(nn.Linear(cin, cin), nn.ReLU(inplace=True))
        else:
            self.linear1 = nn.Sequential(nn.Linear(cout, cout), nn.BatchNorm1d(cout), nn.ReLU(inplace=True))
            self.linear2 = nn.Sequential(nn.Linear(cin, cout), nn.BatchNorm1d(cout), nn.ReLU(inplace=True))

    def forward(self, pxo1, pxo2=None):
        if pxo2 is None:
            _, x, o = pxo1
            parts, st = [], 0
            for en in o.tolist():
                xb = x[st:en]
                parts.append(torch.cat([xb, self.linear2(xb.sum(0, True) / (en - st)).repeat(en - st, 1)], 1))
                st = en
            return self.linear1(torch.cat(parts, 0))
        p1, x1, o1 = pxo1
        p2, x2, o2 = pxo2
        return self.linear1(x1) + interpolation(p2, p1, self.linear2(x2), o2, o1)


class PTBlock(nn.Module):
    """seg_model.py:121-142."""
    expansion = 1

    def __init__(self, cin, planes, share_planes=8, nsample=16):
        super().__init__()
        self.linear1 = nn.Linear(cin, planes, bias=False)
        self.bn1 = nn.BatchNorm1d(planes)
        self.transformer2 = PTLayer(planes, planes, share_planes, nsample)
        self.bn2 = nn.BatchNorm1d(planes)
        self.linear3 = nn.Linear(planes, planes, bias=False)
        self.bn3 = nn.BatchNorm1d(planes)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, pxo):
        p, x, o = pxo
        y = self.relu(self.bn1(self.linear1(x)))
        y = self.relu(self.bn2(self.transformer2([p, y, o])))
        return [p, self.relu(self.bn3(self.linear3(y)) + x), o]


class PointTransformerSeg(nn.Module):
    """seg_model.py:145-207."""

    def __init__(self, blocks=(2, 3, 4, 6, 3), c=6, k=13):
        super().__init__()
        self.c, self.in_planes = c, c
        planes, stride, ns = [32, 64, 128, 256, 512], [1, 4, 4, 4, 4], [8, 16, 16, 16, 16]
        for i in range(5):
            setattr(self, f"enc{i + 1}", self._enc(planes[i], blocks[i], stride[i], ns[i]))
        for i in range(4, -1, -1):
            setattr(self, f"dec{i + 1}", self._dec(planes[i], ns[i], head=(i == 4)))
        self.cls = nn.Sequential(nn.Linear(32, 32), nn.BatchNorm1d(32), nn.ReLU(inplace=True), nn.Linear(32, k))

    def _enc(self, planes, n, stride, ns):
        mods = [TransitionDown(self.in_planes, planes, stride, ns)]
        self.in_planes = planes
        mods += [PTBlock(planes, planes, 8, ns) for _ in range(1, n)]
        return nn.Sequential(*mods)

    def _dec(self, planes, ns, head=False):
        mods = [TransitionUp(self.in_planes, None if head else planes)]
        self.in_planes = planes
        mods.append(PTBlock(planes, planes, 8, ns))
        return nn.Sequential(*mods)

    def forward(self, pxo):
        p0, x0, o0 = pxo
        x0 = p0 if self.c == 3 else torch.cat([p0, x0], 1)
        lv = [self.enc1([p0, x0, o0])]
        for i in range(2, 6):
            lv.append(getattr(self, f"enc{i}")(lv[-1]))
        p5, x5, o5 = lv[4]
        up = self.dec5[1:]([p5, self.dec5[0]([p5, x5, o5]), o5])[1]
        prev = [p5, up, o5]
        for i in range(3, -1, -1):
            p, x, o = lv[i]
            dec = getattr(self, f"dec{i + 1}")
            xi = dec[1:]([p, dec[0]([p, x, o], prev), o])[1]
            prev = [p, xi, o]
        return self.cls(prev[1])


class PointTransformerCompatibility(nn.Module):
    """seg_model.py:215-231."""

    def __init__(self, in_features, num_classes, **_):
        super().__init__()
        self.point_transformer = PointTransformerSeg(c=in_features, k=num_classes)

    def forward(self, x):
        B, C, N = x.shape
        flat = x.transpose(1, 2).reshape(-1, C)
        off = (torch.arange(B, dtype=torch.int32) + 1) * N
        out = self.point_transformer([flat[:, :3].contiguous(), flat[:, 3:].contiguous(), off])
        return out.reshape(B, N, -1).transpose(1, 2)


# ------------------------------------------------------------------------------------------------------------------
# segmentation loss (losses/nnu_loss.py:6-19; losses/dice_loss.py:24-96, 99-152)

def nnu_loss(logits, labels, class_weights=None, smooth=1.0):
    """CE(class_weights) + generalised Dice on softmax probabilities with batch_dice=True, do_bg=True, weights 1/volume.

    logits (B,C,N) float, labels (B,N) int64.  Returns (total, ce, gdl) as 0-d tensors, differentiable in `logits`.
    Restated from the closed form: per class c over all B*N points, tp_c = sum p_c [y=c], fp_c = sum p_c - tp_c,
    fn_c = count_c - tp_c, vol_c = count_c + 1e-6; tp, fp, fn = sum_c (.)/vol_c; dice = (2tp+s)/(2tp+fp+fn+s)."""
    B, C, N = logits.shape
    logp = torch.log_softmax(logits, dim=1)
    w = torch.ones(C, dtype=logits.dtype) if class_weights is None else class_weights.to(logits.dtype)
    picked = logp.gather(1, labels[:, None, :])[:, 0]                       # (B,N)
    wy = w[labels]
    ce = -(wy * picked).sum() / wy.sum()                                    # weighted mean of nn.CrossEntropyLoss
    p = logp.exp()
    onehot = torch.zeros_like(p).scatter_(1, labels[:, None, :], 1.0)
    tp_c = (p * onehot).sum((0, 2))
    sp_c = p.sum((0, 2))
    cnt_c = onehot.sum((0, 2))
    vol = cnt_c + 1e-6
    tp = (tp_c / vol).sum()
    fp = ((sp_c - tp_c) / vol).sum()
    fn = ((cnt_c - tp_c) / vol).sum()
    gdl = -(2 * tp + smooth) / (2 * tp + fp + fn + smooth)
    return ce + gdl, ce, gdl


class NNULoss(nn.Module):
    """losses/nnu_loss.py:6-19.  `w_dice` / `w_ce` are stored and, as in the reference, not applied."""

    def __init__(self, class_weights, w_dice=1, w_ce=1):
        super().__init__()
        self.w_dice, self.w_ce = w_dice, w_ce
        self.class_weights = class_weights

    def forward(self, prediction, target):
        total, ce, gdl = nnu_loss(prediction, target, self.class_weights)
        return total, {"CE": ce, "GDL": gdl}


def predict_full_pointcloud(net, pc, sample_points=1024, n_runs_min=50):
    """models/point_seg_net.py:21-48 for any net with `num_classes`: 4/5 of the runs on random subsets, the rest mix
    still-unseen points with seen ones; same torch.randperm calls in the same order as the reference's loop."""
    import warnings
    n_fill = n_runs_min // 5
    n_first = n_runs_min - n_fill
    n_pts = pc.shape[-1]
    n_cls = net.num_classes if hasattr(net, "num_classes") else net(pc[..., :sample_points]).shape[1]
    acc = torch.zeros(pc.shape[0], n_cls, *pc.shape[2:])
    for _ in range(n_first):
        pts = torch.randperm(n_pts)[:sample_points]
        acc[..., pts] += torch.softmax(net(pc[..., pts]), 1)
    unseen = torch.nonzero(acc.sum(1) == 0)[..., 1]
    if unseen.shape[0] > 0:
        seen = torch.nonzero(acc.sum(1))[..., 1]
        n_mix = sample_points // 2
        pick = torch.randperm(n_fill * n_mix) % len(unseen)
        for r in range(n_fill):
            lo = unseen[pick[r * n_mix:(r + 1) * n_mix]]
            rest = torch.randperm(len(seen))[:sample_points - n_mix]
            pts = torch.cat((lo, rest), 0)
            acc[..., pts] += torch.softmax(net(pc[..., pts]), 1)
        if (acc.sum(1) == 0).any():
            warnings.warn("NOT ALL POINTS HAVE BEEN SEEN")
    return torch.softmax(acc, 1)


def farthest_point_sampling(kpts, num_points, start):
    """dseg_ae_regularization.py:30-43 with the random start passed in (pure torch, CPU)"""
    _, N, _ = kpts.size()
    if N <= num_points:  # :32-35 -- nothing to sample, the cloud comes back whole and in order
        return kpts, torch.arange(N)
    ind = torch.zeros(num_points).long()
    ind[0] = start
    dist = torch.sum((kpts - kpts[:, ind[0], :]) ** 2, dim=2)
    for i in range(1, num_points):
        ind[i] = torch.argmax(dist)
        dist = torch.min(dist, torch.sum((kpts - kpts[:, ind[i], :]) ** 2, dim=2))
    return kpts[:, ind, :], ind


# ---------------------------------------------------------------------------------------------------------------------
# augmentations.py:52-113 + data.py:435-460, restated without pytorch3d (parity unpinned: pytorch3d is not vendored).
# so3_exp_map: Rodrigues' formula as published by pytorch3d (eps 1e-4 on the squared norm); Transform3d: row-vector
# convention, rotate -> scale -> translate.  tests/ cross-check the rotation against scipy's Rotation.from_rotvec.
def so3_exp_map(log_rot, eps=1e-4):
    theta = torch.clamp((log_rot * log_rot).sum(1), eps).sqrt()
    K = torch.zeros(log_rot.shape[0], 3, 3, dtype=log_rot.dtype)
    K[:, 0, 1], K[:, 0, 2] = -log_rot[:, 2], log_rot[:, 1]
    K[:, 1, 0], K[:, 1, 2] = log_rot[:, 2], -log_rot[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -log_rot[:, 1], log_rot[:, 0]
    f1 = (theta.sin() / theta)[:, None, None]
    f2 = ((1 - theta.cos()) / theta ** 2)[:, None, None]
    return torch.eye(3)[None] + f1 * K + f2 * (K @ K)


def augment_points(point_clouds, log_rot, translation, scaling):
    """transform_points(point_clouds (B,3,N), compose_transform(log_rot, translation, scaling)): x -> (x R) * s + t on rows"""
    R = so3_exp_map(log_rot)
    p = point_clouds.transpose(1, 2)                      # (B,N,3) row vectors
    return ((p @ R) * scaling.expand(-1, 3)[:, None, :] + translation[:, None, :]).transpose(1, 2)


def dataset_item(x, labels, sample, log_rot=None, translation=None, scaling=None, binary=False):
    """PointDataset.__getitem__ (data.py:435-460) for one item with the random draws passed in: x (C,N), labels (N)"""
    if log_rot is not None:
        x = torch.cat([augment_points(x[None, :3], log_rot, translation, scaling)[0], x[3:]], dim=0)
    lbl = labels[sample]
    if binary:
        lbl = (lbl != 0).long()
    return x[:, sample], lbl
