"""Prototype of the Gram-matrix backward of  max_n LeakyReLU(BN(X W^T))  (DESIGN 6): torch ops only, checked against the
dense path (fsg_bn_act_max_bwd_f32 + the two GEMMs) and timed as a hipGraph."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fissure_segmentation_amd as fsg
from fissure_segmentation_amd import functional as F
from fissure_segmentation_amd.norm import BatchNorm1d

torch.manual_seed(0)
dev = torch.device("cuda:0")
B, N, K, C, slope = 8, 2048, 192, 1024, 0.2
M = B * N
X = (torch.randn(M, K, device=dev) * 0.7 + 0.3).requires_grad_()
W = (torch.randn(C, K, device=dev) * 0.1).requires_grad_()
bn = BatchNorm1d(C).to(dev).train()
with torch.no_grad():
    bn.weight.copy_(torch.rand(C, device=dev) + 0.5); bn.weight[::5] *= -1
    bn.bias.copy_(torch.randn(C, device=dev) * 0.1)
g = torch.randn(B, C, device=dev)

def dense():
    X.grad = W.grad = bn.weight.grad = bn.bias.grad = None
    y = F.linear_pm(X, W)
    out = F.bn_act_max(y.view(B, N, C), bn, slope)
    out.backward(g)
    return out.detach(), X.grad.clone(), W.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone()

out, dX0, dW0, dg0, db0 = dense()

# ---- Gram path from the same saved quantities
with torch.no_grad():
    y = F.linear_pm(X, W).view(B, N, C)
    mean = y.reshape(M, C).double().mean(0).float()
    var = y.reshape(M, C).double().var(0, unbiased=False).float()
    r = torch.rsqrt(var + bn.eps)
    gamma, beta = bn.weight, bn.bias
    a = gamma * r
    sgn = torch.where(gamma >= 0, 1.0, -1.0)
    ysel, arg = (y * sgn).max(dim=1)            # (B,C): max of y for gamma >= 0, min otherwise
    ysel = ysel * sgn

def gram():
    with torch.no_grad():
        u = a * (ysel - mean) + beta
        h = g * torch.where(u > 0, 1.0, slope)
        dbeta = h.sum(0)
        dgamma = (h * (ysel - mean) * r).sum(0)
        ha = h * a                                # (B,C)
        p = a * dbeta / M
        q = a * dgamma * r / M
        s = X.sum(0)                              # (K)
        G = X.t() @ X                             # (K,K)
        WG = W @ G                                # (C,K)
        Ws = W @ s                                # (C)
        rows = (torch.arange(B, device=dev)[:, None] * N + arg)          # (B,C) row of X selected by (b,c)
        S1a = (ha[:, :, None] * X[rows]).sum(0)                           # (C,K)
        dW = S1a - p[:, None] * s[None, :] - q[:, None] * (WG - Ws[:, None] * s[None, :] / M)
        Wq = q[:, None] * W
        M1 = W.t() @ Wq                           # (K,K)
        T = X @ M1
        cvec = (s / M) @ M1 - p @ W
        dX = cvec[None, :] - T
        dX.index_add_(0, rows.reshape(-1), (ha[:, :, None] * W[None]).reshape(-1, K))
        return dX, dW, dgamma, dbeta

dX1, dW1, dg1, db1 = gram()
def rel(a_, b_): return float((a_ - b_).abs().max() / b_.abs().max())
print("rel err dX %.2e dW %.2e dgamma %.2e dbeta %.2e" % (rel(dX1, dX0), rel(dW1, dW0), rel(dg1, dg0), rel(db1, db0)))
# oracle in float64
with torch.no_grad():
    Xd, Wd = X.double(), W.double()
    yd = (Xd @ Wd.t()).view(B, N, C)
    mu = yd.reshape(M, C).mean(0); va = yd.reshape(M, C).var(0, unbiased=False); rd = torch.rsqrt(va + bn.eps)
    ad = gamma.double() * rd
    ys = torch.gather(yd, 1, arg[:, None, :]).squeeze(1)
    ud = ad * (ys - mu) + beta.double()
    hd = g.double() * torch.where(ud > 0, 1.0, slope)
    dbd = hd.sum(0); dgd = (hd * (ys - mu) * rd).sum(0)
    dU = torch.zeros(B, N, C, dtype=torch.float64, device=dev)
    dU.scatter_(1, arg[:, None, :], hd[:, None, :])
    dy = ad * (dU - dbd / M - (yd - mu) * rd * dgd / M)
    dXd = dy.view(M, C) @ Wd; dWd = dy.view(M, C).t() @ Xd
print("vs fp64: dense dX %.2e dW %.2e | gram dX %.2e dW %.2e" % (rel(dX0.double(), dXd), rel(dW0.double(), dWd), rel(dX1.double(), dXd), rel(dW1.double(), dWd)))

def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(gph): fn()
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(5): gph.replay()
    torch.cuda.synchronize()
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a0.record()
    for _ in range(reps): gph.replay()
    a1.record(); torch.cuda.synchronize()
    return 1e3 * a0.elapsed_time(a1) / reps

yv = F.linear_pm(X, W).detach().view(B, N, C)
def dense_bwd_only():
    with torch.no_grad():
        gy = torch.empty_like(yv)
        dgam, dbet = torch.empty(C, device=dev), torch.empty(C, device=dev)
        import ctypes
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        fsg._lib.call("fsg_bn_act_max_bwd_f32", P(g), P(yv), P(ysel), P(arg.int()), P(gamma.detach()), P(beta.detach()), P(mean), P(r),
                      B, N, C, 1, slope, P(gy), P(dgam), P(dbet), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        gy2 = gy.view(M, C)
        return gy2 @ W, gy2.t() @ X
print("graph replay: dense backward %.1f us, gram (torch ops) %.1f us" % (timeit(dense_bwd_only), timeit(gram)))

# ---- component timings (graph replay)
with torch.no_grad():
    Xn, Wn = X.detach(), W.detach()
    G = Xn.t() @ Xn; M1 = torch.randn(K, K, device=dev); qv = torch.randn(C, device=dev)
    rows = (torch.arange(B, device=dev)[:, None] * N + arg)
    ha = torch.randn(B, C, device=dev)
    comps = {
        "X^T X (vendor)": lambda: Xn.t() @ Xn,
        "X^T X (gemm_small)": lambda: F.gemm_small(Xn, 1, K, Xn, K, 1, None, K, K, M),
        "X^T X (bmm split 64 + sum)": lambda: torch.bmm(Xn.view(64, M // 64, K).transpose(1, 2), Xn.view(64, M // 64, K)).sum(0),
        "X.sum(0)": lambda: Xn.sum(0),
        "W @ G": lambda: Wn @ G,
        "W^T (q W) (vendor)": lambda: Wn.t() @ (qv[:, None] * Wn),
        "W^T (q W) (gemm_small)": lambda: F.gemm_small(Wn, 1, K, qv[:, None] * Wn, K, 1, None, K, K, C),
        "X @ M1": lambda: Xn @ M1,
        "S1a gather-sum": lambda: (ha[:, :, None] * Xn[rows]).sum(0),
        "index_add": lambda: torch.zeros(M, K, device=dev).index_add_(0, rows.reshape(-1), (ha[:, :, None] * Wn[None]).reshape(-1, K)),
    }
    for n, f in comps.items():
        try:
            print("%-32s %7.1f us" % (n, timeit(f)))
        except Exception as e:
            print(n, "failed:", type(e).__name__, str(e)[:100])
