"""GPU diagnostic: per-tensor error report of the HIP-path models against the CPU oracle."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from golden_util import cloud, fill_state_dict, load
from oracle import ref_cpu
import fissure_segmentation_amd as fsg

dev = torch.device("cuda:0")


def report(tag, a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    print(f"  {tag:50s} max|ref|={b.abs().max():.3e} max|d|={(a-b).abs().max():.3e} rel2={(a-b).norm()/(b.norm()+1e-30):.3e}")


def compare(name, ref, net, x, gseed):
    net.load_state_dict(ref.state_dict())
    net = net.to(dev).train(); ref.train()
    xr = torch.from_numpy(x).requires_grad_(True)
    yr = ref(xr)
    gr = torch.from_numpy(np.random.default_rng(gseed).standard_normal(tuple(yr.shape)).astype(np.float32))
    yr.backward(gr)
    xt = torch.from_numpy(x).to(dev).requires_grad_(True)
    y = net(xt); y.backward(gr.to(dev))
    print(name)
    report("out", y, yr); report("grad_x", xt.grad, xr.grad)
    worst = []
    refp = dict(ref.named_parameters())
    for n, p in net.named_parameters():
        q = refp[n]
        a, b = p.grad.double().cpu().reshape(-1), q.grad.double().reshape(-1)
        worst.append((float((a - b).norm() / (b.norm() + 1e-30)), n))
    worst.sort(reverse=True)
    print("  worst param grads:", [(f"{e:.2e}", n) for e, n in worst[:4]])


for backend in ("torch", "c"):
    ref_cpu.KNN_BACKEND = backend
    print("==== oracle kNN backend:", backend)
    from fissure_segmentation_amd.models.folding_net import DGCNNFoldingNet
    ref = fill_state_dict(ref_cpu.DGCNNFoldingNet(k=8, n_embedding=64, n_input_points=2048), 601)
    compare("ae_fold", ref, DGCNNFoldingNet(k=8, n_embedding=64, shape_type="plane", n_input_points=2048, decode_mesh=False),
            cloud(1601, 2, 3, 2048), 2601)
    ref = fill_state_dict(ref_cpu.DGCNNFoldingNet(k=8, n_embedding=64, n_input_points=2048, deform=True, static=True), 602)
    compare("ae_deform_static", ref, DGCNNFoldingNet(k=8, n_embedding=64, shape_type="plane", n_input_points=2048,
            decode_mesh=False, deform=True, static=True), cloud(1602, 2, 3, 2048), 2602)
    from fissure_segmentation_amd.models.dgcnn import DGCNNSeg
    ref = fill_state_dict(ref_cpu.DGCNNSeg(k=20, in_features=3, num_classes=4), 9)
    compare("dgcnnseg N=1024 k=20", ref, DGCNNSeg(k=20, in_features=3, num_classes=4), cloud(19, 2, 3, 1024), 29)

from fissure_segmentation_amd.models.point_net import PointNetSeg
ref = fill_state_dict(ref_cpu.PointNetSeg(3, 4), 501)
compare("pointnet", ref, PointNetSeg(3, 4), cloud(1501, 8, 3, 1024), 2501)

from fissure_segmentation_amd.models.pointtransformer.seg_model import PointTransformerCompatibility
for seed_init in (None, 801):
    torch.manual_seed(0)
    ref = ref_cpu.PointTransformerCompatibility(6, 4)
    if seed_init:
        fill_state_dict(ref, seed_init)
    compare(f"pointtransformer init={'default' if not seed_init else 'filled'}", ref, PointTransformerCompatibility(6, 4),
            cloud(1801, 2, 6, 2048), 2801)
