"""PointTransformer segmentation net on the HIP path (reference: models/pointtransformer/seg_model.py).
Same module tree / state_dict keys; kNN queries, farthest point sampling, neighbour grouping and the
vector-attention aggregate run in libfsg_hip.so.  Host syncs of the reference (`.item()` loops at
:71-75, :104-112) are gone: segment sizes travel with the offsets on the host."""
import torch
import torch.nn as nn

from ...norm import BatchNorm1d
from ..point_seg_net import PointSegmentationModelBase
from ... import functional as F_hip
from . import pointops


class Linear(nn.Linear):
    """nn.Linear (same parameters / state_dict keys) whose products go through functional.linear_pm: vendor GEMM for
    the large ones, fsg_gemm_small_f32 where the vendor library would run a single workgroup."""

    def forward(self, x):
        if not x.is_cuda:
            return super().forward(x)
        return _lin(self, x)


def _lin(layer, x):
    """nn.Linear applied as a point-major GEMM with the split-K weight gradient (functional.linear_pm)."""
    shape = x.shape
    y = F_hip.linear_pm(x.reshape(-1, shape[-1]), layer.weight, layer.bias)
    return y.view(*shape[:-1], -1)


def _bn_relu(bn, x, relu=True, residual=None):
    """[relu](bn(x) [+ residual]) on packed rows: fused HIP stage when the width allows, plain modules otherwise."""
    if F_hip.bn_rows_supported(x, bn):
        return F_hip.bn_rows(x, bn, relu=relu, residual=residual)
    y = bn(x)
    if residual is not None:
        y = y + residual
    return torch.relu(y) if relu else y


def _lin_bn_relu(seq, x):
    """nn.Sequential(Linear, BatchNorm1d, ReLU) -> fused glue"""
    return _bn_relu(seq[1], seq[0](x))


class PointTransformerLayer(nn.Module):
    """Vector attention over the nsample nearest neighbours (seg_model.py:17-53)."""

    def __init__(self, in_planes, out_planes, share_planes=8, nsample=16):
        super().__init__()
        self.mid_planes = mid_planes = out_planes // 1
        self.out_planes, self.share_planes, self.nsample = out_planes, share_planes, nsample
        self.linear_q = Linear(in_planes, mid_planes)
        self.linear_k = Linear(in_planes, mid_planes)
        self.linear_v = Linear(in_planes, out_planes)
        self.linear_p = nn.Sequential(Linear(3, 3), BatchNorm1d(3), nn.ReLU(inplace=True),
                                      Linear(3, out_planes))
        self.linear_w = nn.Sequential(BatchNorm1d(mid_planes), nn.ReLU(inplace=True),
                                      Linear(mid_planes, mid_planes // share_planes),
                                      BatchNorm1d(mid_planes // share_planes), nn.ReLU(inplace=True),
                                      Linear(out_planes // share_planes, out_planes // share_planes))
        self.softmax = nn.Softmax(dim=1)

    @staticmethod
    def _bn_over_neighbours(bn, t):
        # BatchNorm1d on the (n, channels, nsample) view == statistics over n*nsample rows (:42,:47)
        n, ns, c = t.shape
        return bn(t.reshape(n * ns, c)).view(n, ns, c)

    fused = True  # class-wide switch: False composes the layer from the separate grouping / vec_attn ops (tests)

    def forward(self, pxo):
        p, x, o = pxo
        idx, _ = pointops.knn_squared(self.nsample, p, p, o, o)  # one graph for keys and values
        if self.fused and self.out_planes in F_hip.PT_ATTN_PLANES and self.nsample <= 16 and self.share_planes == 8:
            # q, k, v as ONE GEMM, then everything up to the aggregate in the fused HIP layer (no (n,ns,c) tensor in HBM)
            packed = F_hip.packed_qkv(self)
            if packed is None:      # outside PointTransformerSeg.forward: this layer's own two cats
                packed = (torch.cat([self.linear_q.weight, self.linear_k.weight, self.linear_v.weight], 0),
                          torch.cat([self.linear_q.bias, self.linear_k.bias, self.linear_v.bias], 0))
            return F_hip.pt_attn(p, idx, F_hip.linear_pm(x, *packed), self.linear_p, self.linear_w)
        q, k, v = _lin(self.linear_q, x), _lin(self.linear_k, x), _lin(self.linear_v, x)
        rel = pointops.grouping(p, idx) - p.unsqueeze(1)       # (n, ns, 3)
        gk = pointops.grouping(k, idx)                         # (n, ns, c)
        lin1, bn, act, lin2 = self.linear_p
        pr = _lin(lin2, act(self._bn_over_neighbours(bn, _lin(lin1, rel))))
        w = gk - q.unsqueeze(1) + pr
        bn1, act1, lw1, bn2, act2, lw2 = self.linear_w
        w = _lin(lw1, act1(self._bn_over_neighbours(bn1, w)))
        w = _lin(lw2, act2(self._bn_over_neighbours(bn2, w)))
        w = self.softmax(w)                                    # over the neighbours
        return pointops.aggregation(v, pr, w, idx)             # (n, c)


class TransitionDown(nn.Module):
    """seg_model.py:56-84: FPS to n/stride points, group nsample neighbours, Linear-BN-ReLU-maxpool."""

    def __init__(self, in_planes, out_planes, stride=1, nsample=16):
        super().__init__()
        self.stride, self.nsample = stride, nsample
        if stride != 1:
            self.linear = Linear(3 + in_planes, out_planes, bias=False)
            self.pool = nn.MaxPool1d(nsample)
        else:
            self.linear = Linear(in_planes, out_planes, bias=False)
        self.bn = BatchNorm1d(out_planes)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, pxo):
        p, x, o = pxo
        if self.stride == 1:
            return [p, _bn_relu(self.bn, _lin(self.linear, x)), o]
        ends = pointops.host_offsets(o)
        new_ends, prev, total = [], 0, 0
        for e in ends:
            total += (e - prev) // self.stride
            new_ends.append(total)
            prev = e
        n_o = pointops.with_host_offsets(pointops.device_ints(new_ends, p.device, o.dtype), new_ends)
        idx = pointops.furthestsampling(p, o, n_o)
        n_p = p[idx.long(), :].contiguous()
        g = pointops.queryandgroup(self.nsample, p, n_p, x, None, o, n_o, use_xyz=True)  # (m, ns, 3+c)
        m, ns, c = g.shape
        y = _bn_relu(self.bn, _lin(self.linear, g).reshape(m * ns, -1)).view(m, ns, -1)
        return [n_p, F_hip.rows_max(y) if y.is_cuda else y.max(dim=1)[0], n_o]


class TransitionUp(nn.Module):
    """seg_model.py:87-118."""

    def __init__(self, in_planes, out_planes=None):
        super().__init__()
        if out_planes is None:
            self.linear1 = nn.Sequential(Linear(2 * in_planes, in_planes), BatchNorm1d(in_planes),
                                         nn.ReLU(inplace=True))
            self.linear2 = nn.Sequential(Linear(in_planes, in_planes), nn.ReLU(inplace=True))
        else:
            self.linear1 = nn.Sequential(Linear(out_planes, out_planes), BatchNorm1d(out_planes),
                                         nn.ReLU(inplace=True))
            self.linear2 = nn.Sequential(Linear(in_planes, out_planes), BatchNorm1d(out_planes),
                                         nn.ReLU(inplace=True))

    def forward(self, pxo1, pxo2=None):
        if pxo2 is None:  # head: concat every point with its cloud's mean feature (:101-113)
            _, x, o = pxo1
            ends = pointops.host_offsets(o)
            if x.is_cuda and x.dtype == torch.float32 and ends[-1] <= 4096:
                A, S = pointops.segment_mean_matrices(ends, x.device)      # constants of the batch layout
                return _lin_bn_relu(self.linear1, torch.cat((x, S @ self.linear2(A @ x)), dim=1))
            counts = pointops.device_ints([e - s for s, e in zip([0] + ends[:-1], ends)], x.device, torch.int64)
            seg = pointops.segment_ids(ends, x.device)
            mean = torch.zeros(len(ends), x.shape[1], device=x.device, dtype=x.dtype).index_add_(0, seg, x)
            mean = mean / counts.unsqueeze(1).to(x.dtype)
            return _lin_bn_relu(self.linear1, torch.cat((x, self.linear2(mean)[seg]), dim=1))
        p1, x1, o1 = pxo1
        p2, x2, o2 = pxo2
        return _lin_bn_relu(self.linear1, x1) + pointops.interpolation(p2, p1, _lin_bn_relu(self.linear2, x2), o2, o1)


class PointTransformerBlock(nn.Module):
    """seg_model.py:121-142."""
    expansion = 1

    def __init__(self, in_planes, planes, share_planes=8, nsample=16):
        super().__init__()
        self.linear1 = Linear(in_planes, planes, bias=False)
        self.bn1 = BatchNorm1d(planes)
        self.transformer2 = PointTransformerLayer(planes, planes, share_planes, nsample)
        self.bn2 = BatchNorm1d(planes)
        self.linear3 = Linear(planes, planes * self.expansion, bias=False)
        self.bn3 = BatchNorm1d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, pxo):
        p, x, o = pxo
        y = _bn_relu(self.bn1, _lin(self.linear1, x))
        y = _bn_relu(self.bn2, self.transformer2([p, y, o]))
        return [p, _bn_relu(self.bn3, _lin(self.linear3, y), residual=x), o]


class PointTransformerSeg(nn.Module):
    """U-Net of seg_model.py:145-207: planes 32/64/128/256/512, strides 1/4/4/4/4, nsample 8/16/16/16/16."""

    def __init__(self, block, blocks, c=6, k=13):
        super().__init__()
        self.c = c
        self.in_planes, planes = c, [32, 64, 128, 256, 512]
        share_planes, stride, nsample = 8, [1, 4, 4, 4, 4], [8, 16, 16, 16, 16]
        for lvl in range(5):
            setattr(self, f'enc{lvl + 1}', self._make_enc(block, planes[lvl], blocks[lvl], share_planes,
                                                          stride[lvl], nsample[lvl]))
        for lvl in range(4, -1, -1):
            setattr(self, f'dec{lvl + 1}', self._make_dec(block, planes[lvl], 2, share_planes, nsample[lvl],
                                                          is_head=(lvl == 4)))
        self.cls = nn.Sequential(Linear(planes[0], planes[0]), BatchNorm1d(planes[0]), nn.ReLU(inplace=True),
                                 Linear(planes[0], k))

    def _make_enc(self, block, planes, blocks, share_planes=8, stride=1, nsample=16):
        mods = [TransitionDown(self.in_planes, planes * block.expansion, stride, nsample)]
        self.in_planes = planes * block.expansion
        mods += [block(self.in_planes, self.in_planes, share_planes, nsample=nsample) for _ in range(1, blocks)]
        return nn.Sequential(*mods)

    def _make_dec(self, block, planes, blocks, share_planes=8, nsample=16, is_head=False):
        mods = [TransitionUp(self.in_planes, None if is_head else planes * block.expansion)]
        self.in_planes = planes * block.expansion
        mods += [block(self.in_planes, self.in_planes, share_planes, nsample=nsample) for _ in range(1, blocks)]
        return nn.Sequential(*mods)

    def forward(self, pxo):
        p0, x0, o0 = pxo
        if not p0.is_cuda:
            raise RuntimeError("PointTransformer (HIP path) needs its input on the GPU")
        x0 = p0 if self.c == 3 else torch.cat((p0, x0), 1)
        layers = [m for m in self.modules() if isinstance(m, PointTransformerLayer)]
        # [Wq ; Wk ; Wv] and [bq ; bk ; bv] of ALL layers by one copy launch (two cats per layer before)
        with pointops.knn_cache(), F_hip.qkv_pack(layers), F_hip.zero_arena():
            levels = [self.enc1([p0, x0, o0])]
            for lvl in range(2, 6):
                levels.append(getattr(self, f'enc{lvl}')(levels[-1]))
            p, x, o = levels[4]
            coarse = [p, self.dec5[1:]([p, self.dec5[0]([p, x, o]), o])[1], o]
            for lvl in range(3, -1, -1):
                p, x, o = levels[lvl]
                dec = getattr(self, f'dec{lvl + 1}')
                coarse = [p, dec[1:]([p, dec[0]([p, x, o], coarse), o])[1], o]
        return self.cls[3](_bn_relu(self.cls[1], self.cls[0](coarse[1])))


def pointtransformer_seg_repro(**kwargs):
    return PointTransformerSeg(PointTransformerBlock, [2, 3, 4, 6, 3], **kwargs)


class PointTransformerCompatibility(PointSegmentationModelBase):
    """(B,C,N) <-> packed adaptor of seg_model.py:215-231; the first three channels are coordinates."""

    def __init__(self, in_features, num_classes, **kwargs):
        super().__init__(in_features, num_classes)
        self.point_transformer = pointtransformer_seg_repro(c=in_features, k=num_classes)

    @F_hip.with_deferred_bn_counters
    def forward(self, x):
        bs, n_feat, npts = x.shape
        flat = x.transpose(1, 2).reshape(-1, n_feat)
        coords = flat[:, :3].contiguous()
        feat = flat[:, 3:].contiguous()
        ends = [(i + 1) * npts for i in range(bs)]
        offsets = pointops.with_host_offsets(pointops.device_ints(ends, x.device, torch.int32), ends)
        out = self.point_transformer([coords, feat, offsets])
        return out.reshape(bs, npts, -1).transpose(1, 2)
