import torch, statistics
dev = torch.device("cuda:0")
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(iters):
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e)*1e3)
    return statistics.median(ts)
M=16384
for (K,N) in [(192,1024),(192,256),(256,256),(256,128),(128,4),(3,128),(64,128)]:
    X=torch.randn(M,K,device=dev); W=torch.randn(N,K,device=dev); dY=torch.randn(M,N,device=dev)
    fl=2*M*K*N/1e9
    t_f=timeit(lambda: torch.nn.functional.linear(X,W))
    t_dx=timeit(lambda: dY@W)
    t_dw=timeit(lambda: dY.t()@X)
    res=[f"K={K} N={N} ({fl:.2f} GF): fwd {t_f:.0f}us dX {t_dx:.0f}us dW {t_dw:.0f}us"]
    for S in (8,16,32,64):
        t=timeit(lambda: torch.bmm(dY.view(S,M//S,N).transpose(1,2), X.view(S,M//S,K)).sum(0))
        res.append(f"dW-splitK{S} {t:.0f}us")
    ref=dY.t()@X; alt=torch.bmm(dY.view(32,M//32,N).transpose(1,2), X.view(32,M//32,K)).sum(0)
    res.append(f"maxdiff {float((ref-alt).abs().max()):.2e}")
    print("  ".join(res))
