import sys, torch
sys.path[:0] = ["/root/repo"]
import fissure_segmentation_amd as fsg
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
for (S, A, Bc) in [(16, 1024, 192), (16, 256, 192), (16, 256, 256), (16, 128, 256)]:
    p = torch.randn(S, A, Bc, device=dev); ones = torch.ones(1, S, device=dev)
    t1 = timeit(lambda: p.sum(0))
    t2 = timeit(lambda: (ones @ p.view(S, -1)).view(A, Bc))
    M = 16384
    g = torch.randn(M, A, device=dev); x = torch.randn(M, Bc, device=dev)
    t3 = timeit(lambda: torch.bmm(g.view(S, M // S, -1).transpose(1, 2), x.view(S, M // S, -1)))
    t4 = timeit(lambda: g.t() @ x)
    print(S, A, Bc, f"sum(0) {t1:.1f} us, ones@ {t2:.1f} us | bmm {t3:.1f} us, direct g.T@x {t4:.1f} us")
