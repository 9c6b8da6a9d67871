import os, sys, torch
sys.path[:0] = ["/root/repo"]
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
for K in (64, 256, 768):
    print("K", K, "rows I, cols J: NT (x@w.T) | NN (g@w) | TN (g.T@x)")
    for I in (64, 128, 192, 256, 320, 512, 1024):
        line = f"I={I:5d} "
        for J in (64, 128, 256, 512):
            x = torch.randn(I, K, device=dev); w = torch.randn(J, K, device=dev)
            g = torch.randn(I, K, device=dev); w2 = torch.randn(K, J, device=dev)
            gt = torch.randn(K, I, device=dev); x2 = torch.randn(K, J, device=dev)
            a = timeit(lambda: x @ w.t()); b = timeit(lambda: g @ w2); c = timeit(lambda: gt.t() @ x2)
            line += f"| J={J:4d}: {a:6.1f} {b:6.1f} {c:6.1f} "
        print(line)
