"""Point-cloud auto-encoder (DGCNN encoder + folding / deforming decoder) on the HIP path.
Same classes, arguments and state_dict keys as the reference's models/folding_net.py; the encoder's
four graph builds + gathers run in libfsg_hip.so.  Mesh output (pytorch3d `Meshes`) is out of scope:
`decode_mesh=True` uses the mesh-vertex grid of get_plane_mesh but returns the (B,3,m) vertices."""
import torch
from torch import nn

from .. import functional as F_hip
from ..shapes.shape_constructor import get_gaussian, get_plane, get_plane_mesh, get_sphere
from ..norm import BatchNorm1d, BatchNorm2d
from .dgcnn import SharedFullyConnected
from .dgcnn_opensrc import get_graph_feature
from .modelio import LoadableModel, store_config_args

SHAPE_TYPES = ['sphere', 'gaussian', 'plane']


class DGCNN_Cls_Encoder(LoadableModel):
    """folding_net.py:83-141 (the BatchNorms are registered twice there -- `bnX` and `convX.1` -- and
    therefore appear under both names in the state_dict; kept)."""

    @store_config_args
    def __init__(self, k, n_embedding, static=False):
        super().__init__()
        self.static, self.k, self.n_embedding = static, k, n_embedding
        self.bn1, self.bn2, self.bn3, self.bn4 = (BatchNorm2d(c) for c in (64, 64, 128, 256))
        self.bn5 = BatchNorm1d(n_embedding)
        act = lambda: nn.LeakyReLU(negative_slope=0.2)  # noqa: E731
        self.conv1 = nn.Sequential(nn.Conv2d(6, 64, 1, bias=False), self.bn1, act())
        self.conv2 = nn.Sequential(nn.Conv2d(128, 64, 1, bias=False), self.bn2, act())
        self.conv3 = nn.Sequential(nn.Conv2d(128, 128, 1, bias=False), self.bn3, act())
        self.conv4 = nn.Sequential(nn.Conv2d(256, 256, 1, bias=False), self.bn4, act())
        self.conv5 = nn.Sequential(nn.Conv1d(512, n_embedding, 1, bias=False), self.bn5, act())

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("PC-AE encoder (HIP path) needs its input on the GPU")
        graph = F_hip.knn_graph(x, self.k, c_knn=3, fix_diag=False) if self.static else None
        B, _, N = x.shape
        feats, x_pm = [], None
        for block in (self.conv1, self.conv2, self.conv3, self.conv4):
            conv, bn, act = block
            if F_hip.edgeconv1_supported(conv.out_channels, self.k):  # fused gather+conv+BN+LeakyReLU+max
                idx = graph if graph is not None else F_hip.knn_graph(x, self.k, fix_diag=False)
                # the point-major output has two consumers (next block, concatenation): hand the concatenation an alias so
                # that the two gradients reach the backward kernel separately (summed there, the slice taken by stride)
                x, x_pm, x_cat = F_hip.edgeconv1(x, idx, conv.weight, bn, act.negative_slope, x_pm=x_pm, both="twice")
            else:
                x = block(get_graph_feature(x, k=self.k, idx=graph)).max(dim=-1)[0]
                x_pm = x_cat = x.transpose(1, 2).contiguous()
            feats.append(x_cat)
        conv5, bn5, act5 = self.conv5
        y = F_hip.linear_pm(torch.cat(feats, dim=2).view(B * N, -1), conv5.weight.view(conv5.out_channels, -1))
        if conv5.out_channels % 64 == 0:   # BN + LeakyReLU + max over the points, activation never materialised
            code = F_hip.bn_act_max(y.view(B, N, -1), bn5, act5.negative_slope)
        else:
            code = act5(bn5(y)).view(B, N, -1).max(dim=1)[0]
        return code.unsqueeze(1)  # (B, 1, n_embedding)


def _first_layer_split(conv, code, pts_pm):
    """1x1 conv on cat([code repeated over the m points, per-point input]) without building that tensor: the code part is
    one small GEMM per cloud, the per-point part (2 or 3 channels) one thin GEMM -- (B,E), (B,m,c) -> (B*m, Cout)."""
    E = code.shape[1]
    w = conv.weight.view(conv.out_channels, -1)
    per_cloud = nn.functional.linear(code, w[:, :E], conv.bias)                         # (B, Cout)
    per_point = F_hip.linear_pm(pts_pm.reshape(-1, pts_pm.shape[-1]), w[:, E:].contiguous())   # (B*m, Cout)
    B, m = pts_pm.shape[0], pts_pm.shape[1]
    return F_hip.add_per_cloud(per_point.view(B, m, -1), per_cloud).view(B * m, -1)


def _first_layer_relu(conv, code, pts_pm):
    """relu of `_first_layer_split` without materialising the pre-activation: the per-cloud part is one small GEMM, the rest
    one fused pass (fsg_fold_layer1_f32) when the width allows -- (B,E), (B,m,c) -> (B*m, Cout)"""
    E = code.shape[1]
    w = conv.weight.view(conv.out_channels, -1)
    if conv.out_channels % 4 or pts_pm.shape[-1] > 3 or conv.bias is None:
        return torch.relu(_first_layer_split(conv, code, pts_pm))
    w_code, w_pts = F_hip.split_cols(w, E)
    per_cloud = nn.functional.linear(code, w_code, conv.bias)                           # (B, Cout)
    B, m = pts_pm.shape[0], pts_pm.shape[1]
    return F_hip.fold_layer1(pts_pm, w_pts, per_cloud, relu=True).view(B * m, -1)


def _lin(conv, x_pm):
    return F_hip.linear_pm(x_pm, conv.weight.view(conv.out_channels, -1), conv.bias)


class Decoder(LoadableModel):
    """folding_net.py:144-183."""

    @store_config_args
    def __init__(self, shape_type, m=1024, decode_mesh=True):
        super().__init__()
        self.m, self.shape_type, self.decode_mesh = m, shape_type, decode_mesh
        self.folding_points = None
        self.faces = None

    def get_folding_points(self, batch_size):
        if self.folding_points is None or self.folding_points.shape[0] != batch_size:
            device = next(self.parameters()).device
            if self.shape_type == 'plane':
                if self.decode_mesh:
                    pts, faces = get_plane_mesh(n=self.m, xrange=(-0.3, 0.3), yrange=(-0.3, 0.3))
                    self.faces = faces.unsqueeze(0).expand(batch_size, -1, -1).to(device)
                else:
                    pts = torch.from_numpy(get_plane())
            elif self.shape_type == 'sphere':
                if self.decode_mesh:
                    raise NotImplementedError('No sphere mesh defined yet')
                pts = torch.from_numpy(get_sphere())
            elif self.shape_type == 'gaussian':
                if self.decode_mesh:
                    raise ValueError('No gaussian mesh is possible.')
                pts = torch.from_numpy(get_gaussian())
            else:
                raise ValueError(f'No shape named "{self.shape_type}". Use one of {SHAPE_TYPES}.')
            self.folding_points = pts.unsqueeze(0).repeat(batch_size, 1, 1).to(device).float()
        return self.folding_points


def _fold_mlp(cin, width):
    return nn.Sequential(nn.Conv1d(cin, width, 1), nn.ReLU(), nn.Conv1d(width, width, 1), nn.ReLU(),
                         nn.Conv1d(width, 3, 1))


class FoldingDecoder(Decoder):
    """folding_net.py:186-228."""

    @store_config_args
    def __init__(self, n_embedding, shape_type, m=1024, decode_mesh=True):
        super().__init__(shape_type, m, decode_mesh)
        self.folding1 = _fold_mlp(n_embedding + (2 if shape_type == 'plane' else 3), n_embedding)
        self.folding2 = _fold_mlp(n_embedding + 3, n_embedding)

    def forward(self, x):
        B = x.shape[0]
        code = x.reshape(B, -1)                                                  # (B, E)
        pts = self.get_folding_points(B).to(x.device)                            # (B, m, 2|3) point-major
        for fold in (self.folding1, self.folding2):                              # Conv1d-ReLU-Conv1d-ReLU-Conv1d as GEMMs
            y = _first_layer_relu(fold[0], code, pts)                            # one pass: code part + grid part + ReLU
            conv2 = fold[2]                                                      # bias + ReLU in the GEMM epilogue
            y = F_hip.linear_pm_relu(y, conv2.weight.view(conv2.out_channels, -1), conv2.bias)
            pts = _lin(fold[4], y).view(B, self.m, 3)
        return pts.transpose(1, 2).contiguous()                                  # (B, 3, m)


def _deform_mlp(width):
    return nn.Sequential(SharedFullyConnected(width + 3, width, dim=1), SharedFullyConnected(width, width, dim=1),
                         SharedFullyConnected(width, 3, dim=1, last_layer=True))


class DeformingDecoder(Decoder):
    """folding_net.py:231-288."""

    @store_config_args
    def __init__(self, n_embedding, shape_type, m=1024, decode_mesh=True, n_deforming_layers=2):
        super().__init__(shape_type, m, decode_mesh)
        if n_deforming_layers == 2:  # keeps the key names of older checkpoints (:236-250)
            self.deforming1, self.deforming2 = _deform_mlp(n_embedding), _deform_mlp(n_embedding)
            self.deforming_layers = nn.ModuleList([self.deforming1, self.deforming2])
        else:
            self.deforming_layers = nn.ModuleList(_deform_mlp(n_embedding) for _ in range(n_deforming_layers))

    def get_folding_points(self, batch_size):
        pts = super().get_folding_points(batch_size)
        if pts.shape[2] == 2:
            pts = torch.cat([pts, torch.zeros(*pts.shape[:2], 1, device=pts.device)], dim=2)
        return pts

    def forward(self, x):
        from .dgcnn import _norm_act, pointwise_block
        B = x.shape[0]
        code = x.reshape(B, -1)
        pts = self.get_folding_points(B).to(x.device)                            # (B, m, 3) point-major
        for mlp in self.deforming_layers:
            y = _first_layer_split(mlp[0].layers[0], code, pts)
            y = _norm_act(y, list(mlp[0].layers)[1:])
            y = pointwise_block(y, mlp[1])
            pts = pts + pointwise_block(y, mlp[2]).view(B, self.m, 3)
        return pts.transpose(1, 2).contiguous()


class DGCNNFoldingNet(LoadableModel):
    """folding_net.py:42-79."""

    @store_config_args
    def __init__(self, k, n_embedding, shape_type, n_input_points=1024, decode_mesh=True, deform=False,
                 static=False, dec_depth=2):
        super().__init__()
        self.encoder = DGCNN_Cls_Encoder(k, n_embedding, static=static)
        self.n_input_points = n_input_points
        m = int(round(float(n_input_points) ** 0.5)) ** 2  # closest square number (:51)
        if deform:
            self.decoder = DeformingDecoder(n_embedding, shape_type, m, decode_mesh, n_deforming_layers=dec_depth)
        else:
            self.decoder = FoldingDecoder(n_embedding, shape_type, m, decode_mesh)

    @F_hip.with_deferred_bn_counters
    def forward(self, x, return_hidden=False):
        h = self.encoder(x)
        out = self.decoder(h)
        return (out, h) if return_hidden else out

    def predict_full_pointcloud(self, pc, sample_points=1024, n_runs=50):
        acc = torch.zeros(pc.shape[0], self.decoder.m, 3, device=pc.device)
        for _ in range(n_runs):
            perm = torch.randperm(pc.shape[1], device=pc.device)[:sample_points]
            acc += self(pc[:, perm].transpose(1, 2)).transpose(1, 2)
        return acc / n_runs
