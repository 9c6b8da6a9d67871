"""In-kernel cycle stamps of the MONOLITHIC coarse-sweep kNN kernel (flags 268435456 | 536870912; the two-launch form of round 4:
tools/knn_nominate_stamps.py): per phase, mean over waves and workgroups.
python3 tools/knn_split_stamps.py B C N k"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fissure_segmentation_amd as fsg  # noqa: E402
from fissure_segmentation_amd import functional as F  # noqa: E402

B, C, N, k = (int(a) for a in sys.argv[1:5])
extra = int(sys.argv[5]) if len(sys.argv) > 5 else 0
g = torch.Generator().manual_seed(7)
p = torch.rand(B, 3, N, generator=g)
w = torch.randn(C, 3, generator=g)
x = (torch.tanh(torch.einsum("cd,bdn->bcn", w, p)) + 1.0).contiguous().cuda() if C > 3 else (p * 2 - 1).cuda()
for _ in range(3):
    F.knn_graph(x, k, _debug_flags=268435456 | 536870912 | extra)
torch.cuda.synchronize()
lib = fsg._lib.lib
buf = (ctypes.c_ulonglong * (256 * 8 * 16))()
lib.fsg_debug_knn_split_stamps.argtypes = [ctypes.c_void_p]
assert lib.fsg_debug_knn_split_stamps(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 16).astype(np.int64)
nwg = min(256, B * ((N + 63) // 64))
t = t[:nwg]
names = ["setup", "barrier", "sweep 1", "barrier", "tau", "barrier", "sweep 2", "barrier", "bitmap rows + counts",
         "barrier", "decode", "first rows staged", "distances", "ranks + output"]
d = np.diff(t, axis=2)
print(f"B={B} C={C} N={N} k={k} extra flags {extra}: thousands of shader cycles per phase, mean / max over {nwg} workgroups x 8 waves")
for i, nm in enumerate(names):
    print(f"  {nm:24s} {d[:, :, i].mean() / 1000:8.2f} k   max {d[:, :, i].max() / 1000:8.2f} k")
print(f"  {'tau: up to the end of its bisection loop':24s} {(t[:, :, 15] - t[:, :, 4]).mean() / 1000:8.2f} k")
print(f"  {'whole kernel':24s} {(t[:, :, 14] - t[:, :, 0]).mean() / 1000:8.2f} k")
