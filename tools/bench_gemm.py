"""vendor GEMM vs fsg_gemm_small_f32 for the Linear shapes of the two models (y = x W^T, dX = dY W, dW = dY^T X)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import fissure_segmentation_amd as fsg
F = fsg.functional
dev = torch.device("cuda:0")

def timeit(fn, n=30):
    for _ in range(5): fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): g.replay()
    e.record(); torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / (10 * n)

shapes = [(16384, 128, 3), (16384, 128, 64), (16384, 4, 128), (8, 256, 1024), (16384, 96, 32), (4096, 192, 64), (1024, 384, 128),
          (256, 768, 256), (64, 1536, 512), (16384, 32, 32), (4096, 64, 64), (1024, 128, 128), (256, 256, 256), (64, 512, 512),
          (65536, 64, 35), (16384, 128, 67), (4096, 256, 131), (1024, 512, 259)]
print(f"{'M':>6} {'N':>5} {'K':>5} | fwd lib/small us | dX lib/small us | dW lib(bmm+sum)/small us")
for M, N, K in shapes:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev); g = torch.randn(M, N, device=dev)
    t = []
    t.append(timeit(lambda: torch.nn.functional.linear(x, w, b)))
    t.append(timeit(lambda: F.gemm_small(x, K, 1, w, 1, K, b, M, N, K)))
    t.append(timeit(lambda: g @ w))
    t.append(timeit(lambda: F.gemm_small(g, N, 1, w, K, 1, None, M, K, N)))
    S = 16 if (M % 16 == 0 and M >= 4096) else 1
    if S > 1:
        t.append(timeit(lambda: torch.bmm(g.view(S, M // S, -1).transpose(1, 2), x.view(S, M // S, -1)).sum(0)))
    else:
        t.append(timeit(lambda: g.t() @ x))
    t.append(timeit(lambda: F.gemm_small(g, 1, N, x, K, 1, None, N, K, M)))
    print(f"{M:6d} {N:5d} {K:5d} | {t[0]:7.1f} {t[1]:7.1f} | {t[2]:7.1f} {t[3]:7.1f} | {t[4]:7.1f} {t[5]:7.1f}")
