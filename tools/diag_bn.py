import os, sys, time
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from golden_util import cloud, fill_state_dict
from oracle import ref_cpu
import fissure_segmentation_amd as fsg
from fissure_segmentation_amd.models.folding_net import DGCNNFoldingNet
from fissure_segmentation_amd.models.dgcnn import DGCNNSeg
dev = torch.device("cuda:0")
def err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return f"max|ref|={b.abs().max():.3e} max|d|={(a-b).abs().max():.3e}"
x = cloud(1602, 2, 3, 2048)
ref = fill_state_dict(ref_cpu.DGCNNFoldingNet(k=8, n_embedding=64, n_input_points=2048, deform=True, static=True), 602).train()
net = DGCNNFoldingNet(k=8, n_embedding=64, shape_type="plane", n_input_points=2048, decode_mesh=False, deform=True, static=True)
net.load_state_dict(ref.state_dict()); net = net.to(dev).train()
cr = ref.encoder(torch.from_numpy(x))
for en in (True, False):
    with torch.backends.cudnn.flags(enabled=en):
        print("miopen", en, "decoder:", err(net.decoder(cr.to(dev)), ref.decoder(cr)))
# BN-only check on an ill-conditioned input
t = torch.randn(2, 64, 2025) * 0.05 + 5.0
bn = torch.nn.BatchNorm1d(64).train()
yr = bn(t)
for en in (True, False):
    with torch.backends.cudnn.flags(enabled=en):
        print("miopen", en, "bn only:", err(bn.to(dev)(t.to(dev)), yr)); bn.cpu()
# speed of the DGCNN step
net = DGCNNSeg(k=20, in_features=3, num_classes=4).to(dev).train()
opt = torch.optim.Adam(net.parameters(), lr=1e-3)
xb = torch.rand(8, 3, 2048, device=dev) * 2 - 1; yb = torch.randint(0, 4, (8, 2048), device=dev)
for en in (True, False, True, False):
    with torch.backends.cudnn.flags(enabled=en):
        for i in range(13):
            if i == 3:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            opt.zero_grad(); F.cross_entropy(net(xb), yb).backward(); opt.step()
        torch.cuda.synchronize(); print("miopen", en, "ms/step", 100 * (time.perf_counter() - t0))
