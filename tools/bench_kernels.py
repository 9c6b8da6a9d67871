"""Micro-benchmark of single C-ABI entry points (HIP-event timed, median of many launches)."""
import os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import fissure_segmentation_amd as fsg
from golden_util import cloud
F = fsg.functional
dev = torch.device("cuda:0")


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return statistics.median(ts), min(ts)


which = sys.argv[1:] or ["knn"]
if "knn" in which:
    for (B, C, N, k) in [(8, 3, 2048, 20), (8, 64, 2048, 20), (4, 3, 8192, 40), (4, 64, 8192, 40), (8, 128, 2048, 20), (8, 64, 4096, 20),
                         (8, 128, 4096, 20), (8, 3, 4096, 20)]:
        x = torch.from_numpy(cloud(1, B, C, N)).to(dev)
        for name, rows, dbg in (("rows_mfma(v2)", False, 0), ("rows_mfma 8 waves", False, 8192), ("mfma(v1)", False, 8), ("rows(v0)", True, 0)):
            med, mn = timeit(lambda: F.knn_graph(x, k, force_rows_kernel=rows, _debug_flags=dbg))
            print(f"knn B={B} C={C} N={N} k={k} {name:14s}: median {med:8.1f} us  min {mn:8.1f} us")
if "knn2ablate" in which:
    for (B, C, N, k) in [(8, 3, 2048, 20), (8, 64, 2048, 20)]:
        x = torch.from_numpy(cloud(1, B, C, N)).to(dev)
        for name, fl in [("full", 0), ("no phase B", 256), ("no phase A", 512), ("neither", 768)]:
            med, mn = timeit(lambda: F.knn_graph(x, k, _debug_flags=fl))
            print(f"knn v2 C={C} {name:12s}: median {med:8.1f} us")
if "knnablate" in which:
    for (B, C, N, k) in [(8, 3, 2048, 20), (8, 64, 2048, 20)]:
        x = torch.from_numpy(cloud(1, B, C, N)).to(dev)
        for name, fl in [("full", 0), ("nosort", 256), ("nomerge", 512), ("nofilter", 1024), ("nosort+nomerge", 768),
                         ("none", 256 + 512 + 1024)]:
            med, mn = timeit(lambda: F.knn_graph(x, k, _debug_flags=fl | 8))
            print(f"knn C={C} {name:16s}: median {med:8.1f} us")
if "edgeconv" in which:
    from fissure_segmentation_amd.models.dgcnn import EdgeConv
    for (B, C, N, k, couts) in [(8, 64, 2048, 20, [64]), (8, 3, 2048, 20, [64, 64]), (4, 64, 8192, 40, [64]), (4, 3, 8192, 40, [64, 64])]:
        ec = EdgeConv(C, couts, k, first_layer=(C == 3)).to(dev).train()
        x = torch.from_numpy(cloud(1, B, C, N)).to(dev).requires_grad_(True)
        idx = F.knn_graph(x, k, c_knn=3 if C == 3 else None)
        g = torch.randn(B, couts[-1], N, device=dev)
        med, mn = timeit(lambda: ec(x, idx))
        print(f"edgeconv fwd B={B} C={C} N={N} k={k} {couts}: median {med:8.1f} us min {mn:8.1f}")
        def fb():
            y = ec(x, idx); y.backward(g)
        med, mn = timeit(fb)
        print(f"edgeconv fwd+bwd                          : median {med:8.1f} us min {mn:8.1f}")
if "ec2" in which:
    from fissure_segmentation_amd.models.dgcnn import EdgeConv
    B, C, N, k = 8, 3, 2048, 20
    ec = EdgeConv(C, [64, 64], k, first_layer=True).to(dev).train()
    x = torch.from_numpy(cloud(1, B, C, N)).to(dev).requires_grad_(True)
    idx = F.knn_graph(x, k, c_knn=3)
    g = torch.randn(B, 64, N, device=dev)
    def fb():
        y = ec(x, idx); y.backward(g)
    fsg._lib.start_timing()
    for _ in range(20): fb()
    ms = fsg._lib.stop_timing()
    for n, v in ms.items():
        v = sorted(v)[len(v)//2]
        print(f"  {n}: median {1e3*v:.1f} us")
if "chamfer" in which:
    for (B, N) in [(8, 2048), (8, 4096)]:
        a = torch.rand(B, N, 3, device=dev); b = torch.rand(B, N, 3, device=dev)
        med, mn = timeit(lambda: F.chamfer_nn(a, b))
        print(f"chamfer_nn B={B} N={N}: median {med:8.1f} us min {mn:8.1f}")
