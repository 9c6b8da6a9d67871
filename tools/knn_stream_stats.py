"""Survivor / redo statistics of the streamed kNN kernel (flag 131072).  Needs a library built with -DFSG_KNN_STATS:
  cd fissure-segmentation_amd/csrc && hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -DFSG_KNN_STATS -c knn_rows_mfma.hip -o knn_rows_mfma.o && hipcc --offload-arch=gfx950 -shared -fPIC -o ../libfsg_hip.so *.o"""
import ctypes, sys, torch, numpy as np
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import fissure_segmentation_amd as fsg
from golden_util import cloud
lib = fsg._lib.lib
F = fsg.functional
dev = torch.device("cuda:0")
for (B, C, N, k) in [(8, 64, 2048, 20), (8, 3, 2048, 20), (4, 64, 8192, 40)]:
    x = torch.from_numpy(cloud(1, B, C, N)).to(dev)
    out = (ctypes.c_ulonglong * 8)()
    lib.fsg_debug_knn_stats(out, 1)
    F.knn_graph(x, k, _debug_flags=131072); torch.cuda.synchronize()
    lib.fsg_debug_knn_stats(out, 1)
    v = list(out)
    print(B, C, N, k, "fast epochs", v[0], "redo", v[1], "mean survivors/row/epoch", v[2] / max(v[3], 1), "max", v[4], "mean span", v[5] / max(v[0], 1))
