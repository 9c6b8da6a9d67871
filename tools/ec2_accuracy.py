"""Accuracy of the fused two-layer EdgeConv (csrc/edgeconv2.hip) against an fp64 composition of the same layers on the same
graph: python tools/ec2_accuracy.py [B C N k]  (FSG_EC2_OLD=1: the fp32-MFMA kernels; default: the split-bf16 kernels)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fissure_segmentation_amd as fsg  # noqa: E402
from fissure_segmentation_amd.norm import BatchNorm2d  # noqa: E402

B, C, N, k = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (8, 3, 2048, 20)
dev = torch.device("cuda:0")
torch.manual_seed(5)
x = torch.rand(B, C, N, device=dev)
conv1 = torch.nn.Conv2d(2 * C, 64, 1, bias=False).to(dev)
conv2 = torch.nn.Conv2d(64, 64, 1, bias=False).to(dev)
bn1, bn2 = BatchNorm2d(64).to(dev), BatchNorm2d(64).to(dev)
with torch.no_grad():
    for bn in (bn1, bn2):
        bn.weight.copy_(torch.rand(64, device=dev) + 0.5)
        bn.bias.copy_(torch.randn(64, device=dev) * 0.2)
idx = fsg.functional.knn_graph(x, k, c_knn=3)
gr = torch.randn(B, 64, N, device=dev)

xt = x.clone().requires_grad_(True)
y = fsg.functional.edgeconv2(xt, idx, conv1.weight, bn1, conv2.weight, bn2, 0.2)
y.backward(gr)
got = dict(out=y.detach(), grad_x=xt.grad, grad_w1=conv1.weight.grad.clone(), grad_w2=conv2.weight.grad.clone(),
           grad_g1=bn1.weight.grad.clone(), grad_b1=bn1.bias.grad.clone(), grad_g2=bn2.weight.grad.clone(), grad_b2=bn2.bias.grad.clone())

# fp64 composition
xd = x.double().requires_grad_(True)
w1, w2 = conv1.weight.detach().double().view(64, 2 * C).requires_grad_(True), conv2.weight.detach().double().view(64, 64).requires_grad_(True)
g1, b1 = bn1.weight.detach().double().requires_grad_(True), bn1.bias.detach().double().requires_grad_(True)
g2, b2 = bn2.weight.detach().double().requires_grad_(True), bn2.bias.detach().double().requires_grad_(True)
xp = xd.transpose(1, 2)                                              # (B, N, C)
nb = torch.gather(xp.unsqueeze(1).expand(B, N, N, C), 2, idx.long().unsqueeze(-1).expand(B, N, k, C))   # (B, N, k, C)
ctr = xp.unsqueeze(2).expand(B, N, k, C)
e = torch.cat([nb - ctr, ctr], -1)                                   # models/dgcnn.py:28-36 edge features
def block(t, w, g, b_):
    t = t @ w.t()
    mu, var = t.mean((0, 1, 2)), t.var((0, 1, 2), unbiased=False)
    return torch.nn.functional.leaky_relu((t - mu) / torch.sqrt(var + 1e-5) * g + b_, 0.2)
yd = block(block(e, w1, g1, b1), w2, g2, b2).max(2)[0].permute(0, 2, 1)      # (B, 64, N)
yd.backward(gr.double())
want = dict(out=yd.detach(), grad_x=xd.grad, grad_w1=w1.grad.view(64, 2 * C, 1, 1), grad_w2=w2.grad.view(64, 64, 1, 1), grad_g1=g1.grad,
            grad_b1=b1.grad, grad_g2=g2.grad, grad_b2=b2.grad)
print("kernels:", "fp32 MFMA (FSG_EC2_OLD)" if os.environ.get("FSG_EC2_OLD") else "split bf16", "bf16 operands" if os.environ.get("FSG_MFMA_OPERANDS") else "")
for n in got:
    a, b = got[n].double(), want[n]
    print(f"{n:8s} max|err| / max|ref| = {float((a - b).abs().max() / b.abs().max()):9.3e}   rel L2 = {float((a - b).norm() / b.norm()):9.3e}")
