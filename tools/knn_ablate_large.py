import sys, torch, numpy as np
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import fissure_segmentation_amd as fsg
from golden_util import cloud
F = fsg.functional
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(1e3 * s.elapsed_time(e))
    return float(np.median(ts))
for (B, C, N, k) in [(4, 64, 8192, 40), (4, 3, 8192, 40), (8, 64, 4096, 20), (8, 3, 4096, 20), (8, 128, 4096, 20), (8, 64, 2048, 20), (8, 3, 2048, 20)]:
    x = torch.from_numpy(cloud(1, B, C, N)).to(dev)
    print(B, C, N, k, "full %.1f" % timeit(lambda: F.knn_graph(x, k)), "no phase B %.1f" % timeit(lambda: F.knn_graph(x, k, _debug_flags=256)))
