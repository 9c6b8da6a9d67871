"""fwd+bwd(+Adam) timings of the other BASELINE configurations (sanity, not the headline)."""
import os, sys, time
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import fissure_segmentation_amd as fsg
from fissure_segmentation_amd.models.pointtransformer.seg_model import PointTransformerCompatibility
from fissure_segmentation_amd.models.folding_net import DGCNNFoldingNet
from fissure_segmentation_amd.models.dgcnn import DGCNNSeg
from fissure_segmentation_amd.losses.chamfer_loss import ChamferLoss
dev = torch.device("cuda:0")

GRAPH = "--graph" in sys.argv
if GRAPH: sys.argv.remove("--graph")

def run(name, net, x, lossfn, steps=10, warm=3):
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, capturable=GRAPH, fused=True)
    def step():
        opt.zero_grad(set_to_none=True); l = lossfn(net(x)); l.backward(); opt.step(); return l
    if GRAPH:   # static shapes: capture fwd + loss + bwd + Adam once, replay
        name = name.replace("eager", "hipGraph replay")
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warm): step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        opt.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            static_l = step()
        def step():
            g.replay(); return static_l
    for _ in range(warm): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): l = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    pts = x.shape[0] * x.shape[2]
    print(f"{name:55s} {1e3*dt:9.2f} ms/step  {pts/dt/1e6:7.3f} M points/s  loss {float(l):.4f}")

which = sys.argv[1:] or ["pt", "ae", "c4"]
torch.manual_seed(0)
if "pt" in which:
    for B in (8,):
        x = torch.rand(B, 3, 2048, device=dev) * 2 - 1; y = torch.randint(0, 4, (B, 2048), device=dev)
        run(f"C3 PointTransformer B={B} N=2048 in=3 (eager, fp32)", PointTransformerCompatibility(3, 4).to(dev).train(), x, lambda o: F.cross_entropy(o, y))
if "ae" in which:
    x = torch.rand(8, 3, 4096, device=dev) * 2 - 1
    net = DGCNNFoldingNet(k=20, n_embedding=512, shape_type="plane", n_input_points=4096, decode_mesh=True).to(dev).train()
    cl = ChamferLoss()
    run("C5 PC-AE FoldingNet + Chamfer B=8 N=4096 k=20 (eager)", net, x, lambda o: cl(o, x))
if "c4" in which:
    x = torch.rand(4, 3, 8192, device=dev) * 2 - 1; y = torch.randint(0, 4, (4, 8192), device=dev)
    run("C4 shape DGCNN-seg N=8192 k=40 B=4/GPU (eager, fp32)", DGCNNSeg(k=40, in_features=3, num_classes=4).to(dev).train(), x, lambda o: F.cross_entropy(o, y))
