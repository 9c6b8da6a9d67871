"""GPU diagnostic: per-parameter gradient error of a HIP model against the fp64 oracle, beside the fp32 oracle's own error
(the calibration used by tests/test_gpu_parity.py::_model_vs_oracle).  usage: diag_calibrated.py dgcnn_static|dgcnn|pt"""
import copy, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from golden_util import cloud, fill_state_dict
from oracle import ref_cpu
import fissure_segmentation_amd as fsg
ref_cpu.KNN_BACKEND = "c"
dev = torch.device("cuda:0")
what = sys.argv[1] if len(sys.argv) > 1 else "dgcnn_static"
if what.startswith("dgcnn"):
    from fissure_segmentation_amd.models.dgcnn import DGCNNSeg
    dyn = what == "dgcnn"
    ref = fill_state_dict(ref_cpu.DGCNNSeg(k=40, in_features=3, num_classes=4, dynamic=dyn), 7).train()
    net = DGCNNSeg(k=40, in_features=3, num_classes=4, dynamic=dyn)
    x = cloud(4000 + 2048 + 40, 2, 3, 2048)
else:
    from fissure_segmentation_amd.models.pointtransformer.seg_model import PointTransformerCompatibility
    ref = fill_state_dict(ref_cpu.PointTransformerCompatibility(6, 4), 803).train()
    net = PointTransformerCompatibility(6, 4)
    x = cloud(4300, 8, 6, 2048)
net.load_state_dict(ref.state_dict())
net = net.to(dev).train()
xt = torch.from_numpy(x).to(dev).requires_grad_(True)
y = net(xt)
gr = np.random.default_rng(4001).standard_normal(tuple(y.shape)).astype(np.float32)
y.backward(torch.from_numpy(gr).to(dev))
runs = {}
for name, mod in (("f32", ref), ("f64", copy.deepcopy(ref).double())):
    xr = torch.from_numpy(x).to(next(mod.parameters()).dtype).requires_grad_(True)
    yr = mod(xr)
    yr.backward(torch.from_numpy(gr).to(yr.dtype))
    runs[name] = (yr.detach().numpy(), xr.grad.numpy(), {n: p.grad.numpy().astype(np.float64) for n, p in mod.named_parameters()})
print("out max abs hip-f32", np.abs(y.detach().cpu().numpy() - runs["f32"][0]).max(), " f32-f64", np.abs(runs["f32"][0] - runs["f64"][0]).max())
rows = []
for n, p in net.named_parameters():
    g64, g32 = runs["f64"][2][n].reshape(-1), runs["f32"][2][n].reshape(-1)
    got = p.grad.double().cpu().numpy().reshape(-1)
    rows.append((np.linalg.norm(got - g64) / (np.linalg.norm(g32 - g64) + 1e-300), n, np.linalg.norm(g64), np.linalg.norm(got - g64), np.linalg.norm(g32 - g64)))
rows.sort(reverse=True)
for r in rows[:25]:
    print(f"ratio {r[0]:9.2e}  |g64| {r[2]:.3e}  hip-f64 {r[3]:.3e}  f32-f64 {r[4]:.3e}  {r[1]}")
if what.startswith("dgcnn"):
    n = "segmentation.2.layers.1.bias"
    got = dict(net.named_parameters())[n].grad.double().cpu().numpy()
    d = got - runs["f64"][2][n]
    print("elementwise err of", n, np.round(d[:16], 6), "max", np.abs(d).max(), "argmax", np.abs(d).argmax(), "g64 there", runs["f64"][2][n][np.abs(d).argmax()])
