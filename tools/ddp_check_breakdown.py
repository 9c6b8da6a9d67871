"""Per-parameter breakdown of the N=2 rehearsal's gradient check (tests/test_gpu_parity.py::test_bench_two_rank_rehearsal_...):
python tools/ddp_check_breakdown.py PREFIX [--steps S] -- runs tools/ddp_rehearsal.sh, then prints, per parameter tensor, the
relative L2 error of the averaged gradient against the oracle's mean of the shard gradients."""
import json
import os
import subprocess
import sys

import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "oracle"))
prefix = sys.argv[1]
steps = sys.argv[sys.argv.index("--steps") + 1] if "--steps" in sys.argv else "3"
recompute = "--recompute" in sys.argv     # PREFIX_check.npz exists: recompute the HIP gradients eagerly at its parameters
if not recompute:
    r = subprocess.run([os.path.join(root, "tools", "ddp_rehearsal.sh"), prefix, "--steps", steps, "--warmup", "1", "--min-seconds", "0"],
                       cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, open(prefix + ".err").read()[-3000:]
from oracle import ref_cpu  # noqa: E402
from bench import synthetic_batch  # noqa: E402
from fissure_segmentation_amd.models.dgcnn import DGCNNSeg  # noqa: E402

ck = np.load(prefix + "_check.npz")
B, Np, k, world = int(ck["B"]), int(ck["N"]), int(ck["k"]), int(ck["world"])
names = [(n, tuple(p.shape)) for n, p in DGCNNSeg(k=k, in_features=3, num_classes=4).named_parameters()]
ref = ref_cpu.DGCNNSeg(k=k, in_features=3, num_classes=4).train()
refp = dict(ref.named_parameters())
off = 0
with torch.no_grad():
    for n, shp in names:
        cnt = int(np.prod(shp))
        refp[n].copy_(torch.from_numpy(ck["params"][off:off + cnt]).view(shp))
        off += cnt
crit = ref_cpu.NNULoss(torch.tensor([0.4, 1.2, 1.2, 1.2]))
ref_cpu.KNN_BACKEND = "c"
total = None
for rank in range(world):
    x, y = synthetic_batch(B, Np, 4, 1234 + rank, "cpu")
    ref.zero_grad()
    crit(ref(x), y)[0].backward()
    g = torch.cat([refp[n].grad.reshape(-1) for n, _ in names])
    total = g if total is None else total + g
want = (total / world).numpy()
got = ck["avg_grad"]
if recompute:
    import fissure_segmentation_amd as fsg
    dev = torch.device("cuda:0")
    net = DGCNNSeg(k=k, in_features=3, num_classes=4).to(dev).train()
    off = 0
    with torch.no_grad():
        for n, p in net.named_parameters():
            p.copy_(torch.from_numpy(ck["params"][off:off + p.numel()]).view(p.shape))
            off += p.numel()
    from fissure_segmentation_amd.losses.nnu_loss import NNULoss
    critg = NNULoss(torch.tensor([0.4, 1.2, 1.2, 1.2])).to(dev)
    tot = None
    for rank in range(world):
        x, y = synthetic_batch(B, Np, 4, 1234 + rank, dev)
        net.zero_grad()
        critg(net(x), y)[0].backward()
        g = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
        tot = g if tot is None else tot + g
    eager = (tot / world).cpu().numpy()
    print("eager recompute vs dumped avg_grad: rel L2", np.linalg.norm(eager - got) / np.linalg.norm(got),
          "(FSG_FUSED_HEAD=%s FSG_EC2_OLD=%s)" % (os.environ.get("FSG_FUSED_HEAD"), os.environ.get("FSG_EC2_OLD")))
    got = eager
print("total rel L2", np.linalg.norm(got - want) / np.linalg.norm(want))
off = 0
for n, shp in names:
    cnt = int(np.prod(shp))
    w, g = want[off:off + cnt], got[off:off + cnt]
    print(f"{n:40s} {str(shp):18s} |want| {np.linalg.norm(w):10.3e}  rel {np.linalg.norm(g - w) / max(np.linalg.norm(w), 1e-30):9.3e}  "
          f"share of err^2 {np.sum((g - w) ** 2) / np.sum((got - want) ** 2):6.3f}")
    off += cnt
