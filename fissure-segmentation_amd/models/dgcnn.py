"""DGCNN on the HIP path.  Same classes, constructor arguments, forward signatures and state_dict
keys as the reference's models/dgcnn.py; the graph build, the neighbour gather and the edge tensor
come from libfsg_hip.so (fsg_knn_dense_f32, fsg_edge_gather_*_f32)."""
import torch
from torch import nn

from .. import functional as F_hip
from ..norm import BatchNorm1d, BatchNorm2d
from ..utils.general_utils import knn
from ..utils.model_utils import init_weights
from .point_seg_net import PointSegmentationModelBase


def create_neighbor_features(x, k, fixed_knn_graph=None, knn_only_over_coords=False):
    """(B,C,N) -> (B,2C,N,k) = cat(x_j - x_i, x_i)   (reference: models/dgcnn.py:15-36).
    Dynamic graphs include the point itself (self_loop=True, :26-27); the first layer builds the graph
    over the coordinate channels only, passed to the kernel as a strided view (no copy)."""
    if fixed_knn_graph is None:
        graph = F_hip.knn_graph(x, k, c_knn=3 if knn_only_over_coords else None, fix_diag=True)
    else:
        graph = fixed_knn_graph
    return F_hip.edge_features(x, graph)


class ConvBlock(nn.Module):
    """1x1 (or k x k) conv -> BatchNorm -> LeakyReLU, module names as models/dgcnn.py:282-315."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=True, dim=2,
                 negative_slope=1e-2, bn=True, activation=True):
        super().__init__()
        try:
            conv, norm = {1: (nn.Conv1d, BatchNorm1d), 2: (nn.Conv2d, BatchNorm2d),
                          3: (nn.Conv3d, nn.BatchNorm3d)}[dim]
        except KeyError:
            raise ValueError(f'There is no Conv layer for dimensionality {dim}.')
        mods = [conv(in_channels, out_channels, kernel_size, stride=stride, bias=not bn,
                     padding=(kernel_size // 2) if padding else 0)]
        if bn:
            mods.append(norm(out_channels))
        if activation:
            mods.append(nn.LeakyReLU(negative_slope))
        self.layers = nn.ModuleList(mods)

    def forward(self, x):
        for layer in self.layers:
            if (isinstance(layer, nn.modules.conv._ConvNd) and x.is_cuda and all(k == 1 for k in layer.kernel_size)
                    and all(v == 1 for v in layer.stride) and all(v == 0 for v in layer.padding)
                    and all(v == 1 for v in layer.dilation) and layer.groups == 1):
                # a plain 1x1 conv (unit stride, no padding / dilation / groups) is a GEMM over the channel axis: call it
                # as one (rocBLAS) instead of going through MIOpen's convolution solver search; anything else is the conv
                w = layer.weight.reshape(layer.out_channels, layer.in_channels)
                y = torch.matmul(w, x.flatten(2))
                if layer.bias is not None:
                    y = y + layer.bias.view(1, -1, 1)
                x = y.view(x.shape[0], layer.out_channels, *x.shape[2:])
            else:
                x = layer(x)
        return x


class SharedFullyConnected(ConvBlock):
    """models/dgcnn.py:318-323."""

    def __init__(self, in_features, out_features, dim=2, last_layer=False):
        super().__init__(in_features, out_features, dim=dim, kernel_size=1, padding=False,
                         bn=not last_layer, activation=not last_layer, negative_slope=0.2)


class EdgeConv(nn.Module):
    """kNN graph -> edge features -> shared MLP -> max over neighbours (models/dgcnn.py:212-243)."""

    def __init__(self, in_features, out_features_list, k, first_layer=False):
        super().__init__()
        self.k = k
        self.first_layer = first_layer
        widths = [2 * in_features, *out_features_list]
        self.shared_mlp = nn.ModuleList(SharedFullyConnected(a, b) for a, b in zip(widths[:-1], widths[1:]))

    @property
    def fused(self):
        """True when this block runs on one of the fused HIP EdgeConv paths (its first conv then takes the P/Q weight)"""
        mlp = self.shared_mlp
        if len(mlp) == 1 and len(mlp[0].layers) == 3:
            return F_hip.edgeconv1_supported(mlp[0].layers[0].out_channels, self.k)
        return len(mlp) == 2 and all(len(m.layers) == 3 for m in mlp) and \
            F_hip.edgeconv2_supported(mlp[0].layers[0].out_channels, mlp[1].layers[0].out_channels, self.k)

    @staticmethod
    def pq_weights(blocks):
        """P/Q weights of the first conv of several fused EdgeConv blocks in ONE launch (None for unfused blocks)"""
        fused = [b for b in blocks if b.fused]
        ws = F_hip.edge_weights_many([b.shared_mlp[0].layers[0].weight for b in fused]) if fused else []
        it = iter(ws)
        return [next(it) if b.fused else None for b in blocks]

    def forward(self, x, fixed_knn_graph=None, x_pm=None, both=False, w_cat=None, knn_ws=None, w_next=None, pq_given=None):
        """x (B,C,N) -> (B,Cout,N) like the reference; `both=True` additionally returns the point-major copy
        (B,N,Cout) that the point-wise head consumes (`both="twice"`: that copy twice, for two consumers), `x_pm` is an
        optional point-major copy of the input, `knn_ws` a workspace in which the last pass of a fused block prepares the
        NEXT layer's graph build over this block's output (functional.knn_prep_workspace); `w_next` (with knn_ws): the P/Q weight
        of the NEXT fused block -- its [P | Q] rows are then emitted by this block's last pass and returned as one more output
        (None if the pass cannot), to be handed to the next block as `pq_given` (no GEMM launch for its first conv)."""
        if len(self.shared_mlp) == 1 and len(self.shared_mlp[0].layers) == 3 and \
                F_hip.edgeconv1_supported(self.shared_mlp[0].layers[0].out_channels, self.k):
            # fused path: no (B,2C,N,k) / (B,Cout,N,k) tensor is ever written (csrc/edgeconv.hip)
            graph = fixed_knn_graph
            if graph is None:
                graph = F_hip.knn_graph(x, self.k, c_knn=3 if self.first_layer else None, fix_diag=True)
            conv, bn, act = self.shared_mlp[0].layers
            return F_hip.edgeconv1(x, graph, conv.weight, bn, act.negative_slope, x_pm=x_pm, both=both, w_cat=w_cat,
                                   knn_ws=knn_ws, w_next=w_next, pq_given=pq_given)
        if len(self.shared_mlp) == 2 and all(len(m.layers) == 3 for m in self.shared_mlp) and \
                F_hip.edgeconv2_supported(self.shared_mlp[0].layers[0].out_channels,
                                          self.shared_mlp[1].layers[0].out_channels, self.k):
            graph = fixed_knn_graph
            if graph is None:
                graph = F_hip.knn_graph(x, self.k, c_knn=3 if self.first_layer else None, fix_diag=True)
            (conv1, bn1, act), (conv2, bn2, _) = self.shared_mlp[0].layers, self.shared_mlp[1].layers
            return F_hip.edgeconv2(x, graph, conv1.weight, bn1, conv2.weight, bn2, act.negative_slope, x_pm=x_pm,
                                   both=both, w_cat=w_cat, knn_ws=knn_ws, w_next=w_next, pq_given=pq_given)
        e = create_neighbor_features(x, self.k, fixed_knn_graph, knn_only_over_coords=self.first_layer)
        for layer in self.shared_mlp:
            e = layer(e)
        out = e.max(dim=-1)[0]
        if not both:
            return out
        pm = out.transpose(1, 2).contiguous()
        return (out, pm, pm) if both == "twice" else (out, pm)


def pointwise_block(x_pm, block):
    """A SharedFullyConnected(dim=1) block applied to point-major rows (M, Cin): the 1x1 Conv1d is one GEMM over all
    B*N points, BatchNorm1d sees the same B*N samples per channel as on the (B,C,N) layout."""
    conv = block.layers[0]
    y = F_hip.linear_pm(x_pm, conv.weight.view(conv.out_channels, conv.in_channels), conv.bias)
    return _norm_act(y, list(block.layers)[1:])


def _norm_act(y, tail):
    """[BatchNorm, LeakyReLU] tail of a ConvBlock on point-major rows: one fused HIP stage when the width allows."""
    if len(tail) == 2 and isinstance(tail[1], nn.LeakyReLU) and F_hip.bn_act_supported(y, tail[0]):
        return F_hip.bn_act(y, tail[0], tail[1].negative_slope)
    for layer in tail:
        y = layer(y)
    return y


class SpatialTransformer(nn.Module):
    """models/dgcnn.py:246-279."""

    def __init__(self, k):
        super().__init__()
        self.in_features = 3
        self.ec = EdgeConv(3, [64, 128], k)
        self.shared_fc = SharedFullyConnected(128, 1024, dim=1)
        self.mlp = nn.Sequential(nn.Linear(1024, 512), BatchNorm1d(512), nn.LeakyReLU(0.2),
                                 nn.Linear(512, 256), BatchNorm1d(256), nn.LeakyReLU(0.2))
        self.transform = nn.Linear(256, 9)

    def forward(self, x, fixed_knn_graph=None):
        xyz = x[:, :3].contiguous()
        t = self.shared_fc(self.ec(xyz, fixed_knn_graph)).max(dim=-1)[0]
        t = self.transform(self.mlp(t)).view(-1, 3, 3)
        xyz = torch.bmm(xyz.transpose(1, 2), t).transpose(1, 2)
        return torch.cat([xyz, x[:, 3:]], dim=1)

    def init_weights(self):
        self.apply(init_weights)
        nn.init.zeros_(self.transform.weight)
        nn.init.eye_(self.transform.bias.view(3, 3))


class ImageFeatures(nn.Module):
    """models/dgcnn.py:326-343: point-wise MLP on the non-coordinate channels."""

    def __init__(self, in_channels=6, out_channels=(6, 12), kernel_size=1):
        super().__init__()
        chans = [in_channels, *out_channels]
        self.layers = nn.ModuleList(ConvBlock(a, b, kernel_size=kernel_size, dim=1)
                                    for a, b in zip(chans[:-1], chans[1:]))

    def forward(self, x):
        f = x[:, 3:]
        for layer in self.layers:
            f = layer(f)
        return torch.cat([x[:, :3], f], dim=1)


class DGCNNBase(PointSegmentationModelBase):
    """models/dgcnn.py:61-112."""

    def __init__(self, k, in_features, num_classes, spatial_transformer=False, dynamic=True,
                 image_feat_module=False):
        super().__init__(in_features, num_classes, k=k, spatial_transformer=spatial_transformer,
                         dynamic=dynamic, image_feat_module=image_feat_module)
        self.k = k
        self.dynamic = dynamic
        self.knn_graph = None
        if image_feat_module:
            if in_features < 4:
                raise ValueError('Number of In-Features for DGCNN too low if you want to use the image feature '
                                 'module! Need at 3, as the first 3 are assumed to be the point coordinates.')
            self.image_feature_module = ImageFeatures(in_channels=in_features - 3, out_channels=(6, 12))
            self.in_features = 3 + 12
        else:
            self.image_feature_module = None
        self.spatial_transformer = SpatialTransformer(k) if spatial_transformer else None
        self.output_activation = nn.Identity()

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("DGCNN (HIP path) needs its input on the GPU")
        if not self.dynamic:  # static graph over the coordinates, self excluded (:95-96)
            self.knn_graph = F_hip.knn_graph(x, self.k, c_knn=3, fix_diag=True, drop_first=True)
        if self.image_feature_module is not None:
            x = self.image_feature_module(x)
        if self.spatial_transformer is not None:
            x = self.spatial_transformer(x)
        return x

    def init_weights(self):
        self.apply(init_weights)
        if self.spatial_transformer is not None:
            self.spatial_transformer.init_weights()


class DGCNNSeg(DGCNNBase):
    """models/dgcnn.py:115-162."""

    def __init__(self, k, in_features, num_classes, spatial_transformer=False, dynamic=True,
                 image_feat_module=False):
        super().__init__(k, in_features, num_classes, spatial_transformer, dynamic, image_feat_module)
        self.ec1 = EdgeConv(self.in_features, [64, 64], k, first_layer=True)
        self.ec2 = EdgeConv(64, [64], k)
        self.ec3 = EdgeConv(64, [64], k)
        self.global_feature = nn.Sequential(SharedFullyConnected(192, 1024, dim=1), nn.AdaptiveMaxPool1d(1))
        self.segmentation = nn.Sequential(SharedFullyConnected(192 + 1024, 256, dim=1),
                                          SharedFullyConnected(256, 256, dim=1),
                                          SharedFullyConnected(256, 128, dim=1),
                                          SharedFullyConnected(128, self.num_classes, dim=1, last_layer=True))
        self.init_weights()

    @F_hip.with_deferred_bn_counters
    def forward(self, x):
        x = super().forward(x)
        B, _, N = x.shape
        p1c, p2c, p3 = self.edge_levels(x)
        levels = torch.cat([p1c, p2c, p3], dim=2).view(B * N, 192)
        gf = self.global_feature[0].layers                                                   # conv, BN, LeakyReLU
        seg0 = self.segmentation[0]
        w = lambda conv: conv.weight.view(conv.out_channels, -1)
        if F_hip.seg_head_supported(levels, B, N, w(gf[0]), w(seg0.layers[0]), w(self.segmentation[1].layers[0]),
                                    w(self.segmentation[2].layers[0]), w(self.segmentation[3].layers[0]),
                                    blocks=[self.global_feature[0]] + list(self.segmentation)):
            # the whole head as ONE autograd node on the fused kernels of csrc/pointwise.hip: fp32-grade products on the bf16
            # matrix pipe, BatchNorm statistics / apply / backward inside the products' epilogues and prologues, the
            # (B*N, 1024) global-feature activation never written, its backward in Gram form
            y = F_hip.seg_head(levels, B, N, self.global_feature[0], list(self.segmentation))
        else:
            y = self._head_unfused(levels, B, N)
        out = y.view(B, N, self.num_classes).transpose(1, 2)
        # training: hand the loss the (B,cls,N) VIEW of the point-major logits (the fused loss and every ATen loss take
        # strided input; its gradient then arrives point-major, no transposing copies either way); inference keeps the
        # reference's contiguous layout
        return out if self.training and torch.is_grad_enabled() else out.contiguous()

    def edge_levels(self, x):
        """the three EdgeConv blocks (graph build + neighbour gather + shared MLP + max over the neighbours, models/dgcnn.py:
        130-132,150-154 of the reference) -> their point-major outputs (B,N,64) each; also what bench.py times as the
        forward "kNN + gather" group of the north star"""
        B, _, N = x.shape
        # EdgeConvs hand over both layouts: channel-major (B,C,N) feeds the next graph build, point-major (B,N,C)
        # feeds the GEMMs; the head runs point-major, so every 1x1 conv is ONE GEMM over the B*N points
        # p1 / p2 have two consumers (the next EdgeConv and the concatenation): "twice" hands out an alias for the second,
        # so that their gradients reach the EdgeConv backward kernel separately (summed there, slices taken by stride)
        w1, w2, w3 = EdgeConv.pq_weights([self.ec1, self.ec2, self.ec3])     # one launch for the three weight transforms
        if self.dynamic and all(e.fused for e in (self.ec1, self.ec2, self.ec3)):
            # the three dynamic graphs go into ONE buffer: with a backward pass ahead their reverse graphs (CSR by destination,
            # what the EdgeConv backward gathers through) are then built by one set of launches instead of three
            graphs = list(torch.empty(3, B, N, self.k, dtype=torch.int32, device=x.device).unbind(0))
            # the feature-space graph builds are PREPARED by the block that produces their points: its last pass emits the
            # norms and the coarse operand image of its output on the way (one launch less per build)
            ws2, ws3 = F_hip.knn_prep_workspace(B, N, 64, x.device), F_hip.knn_prep_workspace(B, N, 64, x.device)
            # the coordinate build also emits the first block's [P | Q] rows when the cloud IS the coordinates (K = 3 product)
            g1, pq1 = F_hip.knn_graph(x, self.k, c_knn=3, fix_diag=True, out=graphs[0], pq_weight=w1)
            # (the apply pass of a block also emits the NEXT block's [P | Q] rows from the tile it holds in LDS: w_next / pq_given)
            x1, p1, p1c, pq2 = self.ec1(x, g1, both="twice", w_cat=w1, knn_ws=ws2, w_next=w2, pq_given=pq1)
            g2 = F_hip.knn_graph(x1, self.k, fix_diag=True, out=graphs[1], prepared=None if ws2 is None else (ws2, p1))
            x2, p2, p2c, pq3 = self.ec2(x1, g2, x_pm=p1, both="twice", w_cat=w2, knn_ws=ws3, w_next=w3, pq_given=pq2)
            g3 = F_hip.knn_graph(x2, self.k, fix_diag=True, out=graphs[2], prepared=None if ws3 is None else (ws3, p2))
            _, p3 = self.ec3(x2, g3, x_pm=p2, both=True, w_cat=w3, pq_given=pq3)
            if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
                F_hip.build_reverse_graphs([g1, g2, g3])
        else:
            x1, p1, p1c = self.ec1(x, self.knn_graph, both="twice", w_cat=w1)
            x2, p2, p2c = self.ec2(x1, self.knn_graph, x_pm=p1, both="twice", w_cat=w2)
            _, p3 = self.ec3(x2, self.knn_graph, x_pm=p2, both=True, w_cat=w3)
        return p1c, p2c, p3

    def _head_unfused(self, levels, B, N):
        """the round-2 head: vendor GEMMs + fsg_bn_act_* stages (shapes outside the fused kernels' envelope, bf16 operand
        mode, cross-check of the fused head)"""
        gf = self.global_feature[0].layers
        # the two layers that read `levels` (global-feature conv and the `levels` half of the first head conv) share one
        # autograd node, so that their input gradients are accumulated by the second GEMM instead of an extra add
        seg0 = self.segmentation[0]
        w0 = seg0.layers[0].weight.view(seg0.layers[0].out_channels, -1)
        w0_levels, w0_global = F_hip.split_cols(w0, 192)
        yg, y0 = F_hip.linear_pm2(levels, gf[0].weight.view(gf[0].out_channels, -1), w0_levels)
        if yg.shape[1] % 64 == 0:  # BN + LeakyReLU + max over the points in one stage, activation never written
            g = F_hip.bn_act_max(yg.view(B, N, -1), gf[1], gf[2].negative_slope)              # (B,1024)
        else:
            g = _norm_act(yg, list(gf)[1:]).view(B, N, -1).max(dim=1)[0]
        # first head layer on cat([levels, g.repeat(N)]) (models/dgcnn.py:159-160 of the reference): the global part
        # is constant per cloud, so its product is computed once per cloud instead of once per point
        y = F_hip.add_per_cloud(y0.view(B, N, -1), nn.functional.linear(g, w0_global))
        y = _norm_act(y.view(B * N, -1), list(seg0.layers)[1:])
        for block in list(self.segmentation)[1:]:
            y = pointwise_block(y, block)
        return y


class DGCNNReg(DGCNNBase):
    """models/dgcnn.py:165-209."""

    def __init__(self, k, in_features, num_classes, spatial_transformer=False, dynamic=True,
                 image_feat_module=False):
        super().__init__(k, in_features, num_classes, spatial_transformer, dynamic, image_feat_module)
        self.ec1 = EdgeConv(self.in_features, [64], k, first_layer=True)
        self.ec2 = EdgeConv(64, [64], k)
        self.ec3 = EdgeConv(64, [128], k)
        self.ec4 = EdgeConv(128, [256], k)
        self.global_feature = nn.Sequential(SharedFullyConnected(512, 1024, dim=1), nn.AdaptiveMaxPool1d(1))
        self.regression = nn.Sequential(SharedFullyConnected(1024, 512, dim=1),
                                        SharedFullyConnected(512, 256, dim=1),
                                        SharedFullyConnected(256, self.num_classes, dim=1, last_layer=True))
        self.init_weights()

    @F_hip.with_deferred_bn_counters
    def forward(self, x):
        x = super().forward(x)
        feats = []
        for ec in (self.ec1, self.ec2, self.ec3, self.ec4):
            x = ec(x, self.knn_graph)
            feats.append(x)
        return self.regression(self.global_feature(torch.cat(feats, dim=1)))

    def predict_full_pointcloud(self, pc, sample_points=1024, n_runs_min=50):
        acc = torch.zeros(pc.shape[0], self.num_classes, 1, device=pc.device)
        if self._ensemble_batchable(pc):    # eval mode: the runs are independent clouds -> batches of runs, summed in run order
            B, per = pc.shape[0], max(1, self.ensemble_max_clouds // max(pc.shape[0], 1))
            pts = torch.stack([torch.randperm(pc.shape[-1], device=pc.device)[:sample_points] for _ in range(n_runs_min)])
            for r0 in range(0, n_runs_min, per):
                chunk = pts[r0:r0 + per]
                x = pc[:, :, chunk].permute(2, 0, 1, 3).reshape(chunk.shape[0] * B, pc.shape[1], chunk.shape[1])
                for o in self(x).view(chunk.shape[0], B, self.num_classes, 1):
                    acc += o
            return acc / n_runs_min
        for _ in range(n_runs_min):
            acc += self(pc[..., torch.randperm(pc.shape[-1], device=pc.device)[:sample_points]])
        return acc / n_runs_min
