"""fsg_knn_dense_ws_f32 (coarse sweep + exact refine, csrc/knn_split.hip) against the two-phase kernel (flag 2097152) and the
C oracle: indices and distance bits, then graph-replayed timings.  python tools/knn_split_check.py [--time-only]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fissure_segmentation_amd as fsg  # noqa: E402
from fissure_segmentation_amd import functional as F  # noqa: E402

OLD = 2097152
MONO = 536870912   # the monolithic coarse-sweep kernel of rounds 2-3 (one launch)
dev = torch.device("cuda:0")


def feats(seed, B, C, N, kind):
    g = torch.Generator().manual_seed(seed)
    if kind == "uniform":
        x = torch.rand(B, C, N, generator=g) * 2 - 1
    elif kind == "biased":      # what an EdgeConv emits: max over k of LeakyReLU(BN(.)): positive mean, small spread
        x = 1.9 + 0.5 * torch.randn(B, C, N, generator=g)
    elif kind == "lowdim":      # features that are a smooth function of 3-D positions
        p = torch.rand(B, 3, N, generator=g)
        w = torch.randn(C, 3, generator=g)
        x = torch.tanh(torch.einsum("cd,bdn->bcn", w, p)) + 1.0
    elif kind == "dups":        # every point twice + a lattice: massive ties
        p = torch.randint(0, 6, (B, C, N // 2), generator=g).float()
        x = torch.cat([p, p], 2)
    elif kind == "const":
        x = torch.ones(B, C, N)
    elif kind == "outlier":     # one point far outside the sampled range: the fp16 image marks it, the cloud takes the slow path
        x = torch.rand(B, C, N, generator=g) * 2 - 1
        x[0, :, N // 3] = 3.0e5
    elif kind == "tiny":        # everything far below fp16's normal range before scaling
        x = (torch.rand(B, C, N, generator=g) * 2 - 1) * 1e-9
    return x.contiguous()


def check(B, C, N, k, kind, c_knn=None, drop=False, oracle=False, flags=0):
    x = feats(B * 1000 + C + N, B, C, N, kind).to(dev)
    i_new, d_new = F.knn_graph(x, k, c_knn=c_knn, drop_first=drop, return_dist=True, _debug_flags=flags)
    i_old, d_old = F.knn_graph(x, k, c_knn=c_knn, drop_first=drop, return_dist=True, _debug_flags=OLD)
    torch.cuda.synchronize()
    same_i = torch.equal(i_new, i_old)
    same_d = torch.equal(d_new.view(torch.int32), d_old.view(torch.int32))
    msg = f"B={B} C={C} c_knn={c_knn} N={N} k={k} {kind:8s} drop={drop} flags={flags}: idx {'==' if same_i else '!='} two-phase, dist bits {'==' if same_d else '!='}"
    if not same_i:
        bad = (i_new != i_old).any(-1).nonzero()
        msg += f"  ({bad.shape[0]} rows differ, first {bad[0].tolist()}: new {i_new[tuple(bad[0])].tolist()} old {i_old[tuple(bad[0])].tolist()})"
    if oracle:
        from oracle import c_api
        xc = x.cpu().numpy()
        if c_knn is not None:
            xc = np.ascontiguousarray(xc[:, :c_knn])
        io, do = c_api.knn_dense(xc, k + (1 if drop else 0))
        if drop:
            io, do = io[..., 1:], do[..., 1:]
        msg += f"; vs C oracle idx {'==' if np.array_equal(io, i_new.cpu().numpy()) else '!='} dist {'==' if np.array_equal(do.view(np.int32), d_new.cpu().numpy().view(np.int32)) else '!='}"
    print(msg, flush=True)
    return same_i and same_d


def timeit(B, C, N, k, kind, flags, reps=20):
    x = feats(7, B, C, N, kind).to(dev)
    F.knn_graph(x, k, _debug_flags=flags)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        F.knn_graph(x, k, _debug_flags=flags)
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                F.knn_graph(x, k, _debug_flags=flags)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / reps)
    return float(np.median(ts))


if __name__ == "__main__":
    ok = True
    if "--time-only" not in sys.argv:
        ok &= check(2, 64, 2048, 20, "uniform", oracle=True)
        ok &= check(2, 3, 2048, 20, "uniform", oracle=True)
        ok &= check(8, 64, 2048, 20, "biased")
        ok &= check(8, 64, 2048, 20, "lowdim")
        ok &= check(2, 6, 2048, 20, "uniform", c_knn=3)
        ok &= check(2, 10, 1024, 16, "uniform")
        ok &= check(2, 24, 1500, 20, "lowdim", drop=True)
        ok &= check(1, 64, 8192, 40, "lowdim", oracle=True)
        ok &= check(2, 3, 8192, 40, "uniform")
        ok &= check(1, 33, 4096, 63, "biased", drop=True)
        ok &= check(2, 3, 1024, 20, "dups")
        ok &= check(2, 128, 2048, 20, "lowdim", oracle=True)     # 8 k-steps: the PC-AE encoder's last graph
        ok &= check(2, 100, 4096, 33, "biased", drop=True)
        ok &= check(1, 128, 1024, 20, "uniform", flags=4194304)
        ok &= check(1, 16, 1024, 8, "const")
        ok &= check(2, 64, 2048, 20, "uniform", flags=4194304)   # everything through the slow path
        ok &= check(2, 16, 1024, 20, "outlier", oracle=True)
        ok &= check(2, 16, 1024, 20, "tiny", oracle=True)
        for fl in (1073741824,):                                 # the three-product bf16 form
            ok &= check(2, 64, 2048, 20, "biased", flags=fl)
            ok &= check(2, 3, 2048, 20, "uniform", flags=fl, oracle=True)
            ok &= check(1, 33, 4096, 63, "biased", drop=True, flags=fl)
        for args in ((8, 64, 2048, 20, "biased"), (2, 3, 2048, 20, "uniform"), (1, 64, 8192, 40, "lowdim"), (2, 16, 1024, 20, "outlier")):
            ok &= check(*args, flags=MONO)                        # the monolithic kernel still agrees
        print("ALL EQUAL" if ok else "MISMATCH", flush=True)
    import ctypes
    lib = fsg._lib.lib
    lib.fsg_debug_knn_split_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]
    st = (ctypes.c_ulonglong * 4)()
    for (B, C, N, k, kind) in [(8, 64, 2048, 20, "lowdim"), (8, 3, 2048, 20, "uniform"), (4, 64, 8192, 40, "lowdim"),
                               (4, 8, 1100, 60, "uniform")]:
        x = feats(7, B, C, N, kind).to(dev)
        lib.fsg_debug_knn_split_stats(st, 1)
        F.knn_graph(x, k, _debug_flags=33554432)
        torch.cuda.synchronize()
        lib.fsg_debug_knn_split_stats(st, 0)
        print(f"stats B={B} C={C} N={N} k={k} {kind:8s}: {st[1] / max(st[0], 1):.1f} listed per query, max {st[3]}, slow {st[2]} of {st[0]}", flush=True)
        for nm, fl in (("setup", 67108864), ("setup+sweep1", 8388608), ("..+tau+sweep2", 16777216)):
            print(f"   monolithic kernel, {nm}: {timeit(B, C, N, k, kind, fl | MONO):.1f} us", flush=True)
    for (B, C, N, k, kind) in [(8, 64, 2048, 20, "biased"), (8, 64, 2048, 20, "lowdim"), (8, 64, 2048, 20, "uniform"),
                               (8, 3, 2048, 20, "uniform"), (4, 64, 8192, 40, "lowdim"), (4, 3, 8192, 40, "uniform"),
                               (32, 3, 2048, 40, "uniform"), (8, 3, 4096, 20, "uniform"), (8, 64, 4096, 20, "lowdim"),
                               (8, 128, 4096, 20, "lowdim")]:
        tn, tb, to = timeit(B, C, N, k, kind, 0), timeit(B, C, N, k, kind, 1073741824), timeit(B, C, N, k, kind, OLD)
        tm = timeit(B, C, N, k, kind, MONO)
        print(f"time B={B} C={C} N={N} k={k} {kind:8s}: nominate + refine {tn:8.1f} us   (3 x bf16 form {tb:8.1f} us)   monolithic {tm:8.1f} us   two-phase {to:8.1f} us", flush=True)
    sys.exit(0 if ok else 1)
