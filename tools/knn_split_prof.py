"""rocprofv3 target: a handful of fsg_knn_dense_ws_f32 calls at one shape.  python3 tools/knn_split_prof.py B C N k [flags]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fissure_segmentation_amd import functional as F  # noqa: E402

B, C, N, k = (int(a) for a in sys.argv[1:5])
flags = int(sys.argv[5]) if len(sys.argv) > 5 else 0
g = torch.Generator().manual_seed(7)
p = torch.rand(B, 3, N, generator=g)
w = torch.randn(C, 3, generator=g)
x = (torch.tanh(torch.einsum("cd,bdn->bcn", w, p)) + 1.0).contiguous().cuda() if C > 3 else (p * 2 - 1).cuda()
for _ in range(20):
    F.knn_graph(x, k, _debug_flags=flags)
torch.cuda.synchronize()
