import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from golden_util import cloud, fill_state_dict
from oracle import ref_cpu
import fissure_segmentation_amd as fsg
from fissure_segmentation_amd.models.folding_net import DGCNNFoldingNet
dev = torch.device("cuda:0")
def err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return f"max|ref|={b.abs().max():.3e} max|d|={(a-b).abs().max():.3e}"
x = cloud(1602, 2, 3, 2048)
for deform in (False, True):
    for static in (False, True):
        ref = fill_state_dict(ref_cpu.DGCNNFoldingNet(k=8, n_embedding=64, n_input_points=2048, deform=deform, static=static), 602).train()
        net = DGCNNFoldingNet(k=8, n_embedding=64, shape_type="plane", n_input_points=2048, decode_mesh=False, deform=deform, static=static)
        net.load_state_dict(ref.state_dict()); net = net.to(dev).train()
        xr = torch.from_numpy(x); xt = xr.to(dev)
        cr = ref.encoder(xr); c = net.encoder(xt)
        print(f"deform={deform} static={static}: code {err(c, cr)}")
        print("   decoder on oracle code:", err(net.decoder(cr.to(dev)), ref.decoder(cr)))
        # per-layer encoder check
        from fissure_segmentation_amd.models.dgcnn_opensrc import get_graph_feature
        g = fsg.functional.knn_graph(xt, 8, c_knn=3, fix_diag=False) if static else None
        gr_ = ref_cpu.knn_opensrc(xr[:, :3], 8) if static else None
        a, b = xt, xr
        for i in range(1, 5):
            idx_r = gr_ if gr_ is not None else ref_cpu.knn_opensrc(b, 8)
            ea = get_graph_feature(a, 8, g); eb = ref_cpu.edge_features(b, idx_r)
            print(f"   layer{i} edge {err(ea, eb)}", end="")
            a = getattr(net.encoder, f"conv{i}")(ea).max(-1)[0]; b = getattr(ref.encoder, f"conv{i}")(eb).max(-1)[0]
            print(f"  feat {err(a, b)}")
