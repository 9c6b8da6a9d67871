import sys, torch
sys.path[:0] = ["/root/repo"]
import fissure_segmentation_amd as fsg
F = fsg.functional
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): g.replay()
    e.record(); torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / (10 * n)
for (M, N) in [(16384, 4), (32768, 3), (32768, 4), (131072, 4), (16384, 16)]:
    g2 = torch.randn(M, N, device=dev); ones = torch.ones(1, M, device=dev); onesc = torch.ones(M, 1, device=dev)
    print(M, N, "sum(0) %.1f" % timeit(lambda: g2.sum(0)), "ones@ %.1f" % timeit(lambda: ones @ g2),
          "gemm_small %.1f" % timeit(lambda: F.gemm_small(g2, 1, N, onesc, 1, 1, None, N, 1, M)))
print("per-cloud sum over points: (B,N,C)")
for (B, N, C) in [(8, 2048, 256), (8, 4096, 512), (32, 2048, 256), (4, 8192, 256)]:
    g = torch.randn(B, N, C, device=dev); ones = torch.ones(B, 1, N, device=dev)
    print(B, N, C, "sum(1) %.1f" % timeit(lambda: g.sum(1)), "bmm ones %.1f" % timeit(lambda: torch.bmm(ones, g)))
