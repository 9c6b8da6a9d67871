"""fsg_adam_flat_f32 over 1.8 M (DGCNN-seg) and 7.8 M (PointTransformer) parameters: microseconds per update, eager launches."""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fissure_segmentation_amd.optim import FlatAdam
for n in (1_800_000, 7_800_000):
    p = torch.nn.Parameter(torch.randn(n, device="cuda"))
    o = FlatAdam([p], lr=1e-3)
    p.grad = torch.randn(n, device="cuda"); o.gather_grads()
    for _ in range(5): o.step_flat()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50): o.step_flat()
    e.record(); torch.cuda.synchronize()
    print(n, "%.1f us" % (1e3 * s.elapsed_time(e) / 50))
