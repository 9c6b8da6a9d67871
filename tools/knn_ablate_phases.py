"""Phase split / A-B switches of the production kNN kernel at the BASELINE shapes.  Variants are timed round-robin (five
rounds of 20 launches each, best median) so that clock ramp-up does not favour whichever variant runs last.
flags: 256 no phase B, 512 no phase A, 65536 plain (not XCD-aware) workgroup placement."""
import sys, torch, numpy as np
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import fissure_segmentation_amd as fsg
from golden_util import cloud
F = fsg.functional
dev = torch.device("cuda:0")
# 2097152 forces the two-phase kernel (knn_rows_mfma.hip): without it fsg_knn_dense_ws_f32 routes these shapes to knn_split.hip,
# which ignores the two-phase ablation bits
TP = 2097152
VARIANTS = (("split (default)", 0), ("two-phase", TP), ("tp 4-byte loads", TP | 1048576), ("tp-noB", TP | 256),
            ("tp-neither", TP | 768))
def med(fn, n=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(1e3 * s.elapsed_time(e))
    return float(np.median(ts))
for (B, C, N, k) in [(8, 64, 2048, 20), (8, 3, 2048, 20), (4, 64, 8192, 40), (8, 128, 4096, 20), (8, 100, 2048, 20)]:
    x = torch.from_numpy(cloud(1, B, C, N)).to(dev)
    best = {n: 1e9 for n, _ in VARIANTS}
    for _ in range(200): F.knn_graph(x, k)          # clocks up
    for r in range(5):
        for n, f in VARIANTS:
            best[n] = min(best[n], med(lambda: F.knn_graph(x, k, _debug_flags=f)))
    print(B, C, N, k, " ".join("%s %.1f" % (n, best[n]) for n, _ in VARIANTS))
