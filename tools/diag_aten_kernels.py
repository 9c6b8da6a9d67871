"""Which ATen ops (with shapes) launch the small element-wise / reduce / cat kernels of one DGCNN-seg training step."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fissure_segmentation_amd as fsg
from fissure_segmentation_amd.models.dgcnn import DGCNNSeg
from fissure_segmentation_amd.losses.nnu_loss import NNULoss
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = DGCNNSeg(k=20, in_features=3, num_classes=4).to(dev).train()
x = torch.rand(8, 3, 2048, device=dev) * 2 - 1
y = torch.randint(0, 4, (8, 2048), device=dev)
crit = NNULoss(torch.tensor([0.4, 1.2, 1.2, 1.2])).to(dev)
def step():
    net.zero_grad(set_to_none=True)
    loss, _ = crit(net(x), y)
    loss.backward()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::")]
seen = 0
for e in sorted(evs, key=lambda e: e.time_range.start):
    ks = [k.name for k in e.kernels]
    if not ks: continue
    if any(("elementwise" in k or "reduce_kernel" in k or "CatArray" in k or "multi_tensor" in k) for k in ks):
        par = e.cpu_parent.name if e.cpu_parent is not None else "-"
        if par.startswith("aten::"): continue      # print the outermost aten op only
        print(f"{e.name:28s} {str(e.input_shapes)[:90]:90s} parent={par[:40]:40s} -> {[k[:40] for k in ks]}")
