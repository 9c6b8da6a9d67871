"""Phase split of the production kNN kernel at the BASELINE config-2 shapes: full, without phase B (flag 256), without
phase A (flag 512), and A/B switches (65536: plain (not XCD-aware) workgroup placement; 32768: generic phase-A loop)."""
import sys, torch, numpy as np
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import fissure_segmentation_amd as fsg
from golden_util import cloud
F = fsg.functional
dev = torch.device("cuda:0")
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(1e3 * s.elapsed_time(e))
    return float(np.median(ts))
for (B, C, N, k) in [(8, 64, 2048, 20), (8, 3, 2048, 20), (4, 64, 8192, 40)]:
    x = torch.from_numpy(cloud(1, B, C, N)).to(dev)
    print(B, C, N, k, " ".join("%s %.1f" % (n, timeit(lambda: F.knn_graph(x, k, _debug_flags=f)))
                               for n, f in (("full", 0), ("noB", 256), ("generic", 32768), ("generic-noB", 32768 + 256),
                                            ("plain-placement", 65536), ("plain-noB", 65536 + 256))))
