"""fsg_pw_linear_f32 (three-piece bf16 split, six MFMA products, csrc/pointwise.hip) against the vendor fp32 GEMM at the shapes
of the DGCNN head (BASELINE config 2: M = 16384 points).  Both captured into a hipGraph (10 launches) and replayed between HIP
events; TF = algorithmic fp32 flop 2 M N K per second.
(profiles/r3_pw_rowgemm_ablation.txt: phase ablations of the first version of the kernel, taken with runtime switches that were
removed again because the branches themselves cost the 128x128 tile 30 %.)"""
import sys
import numpy as np
import torch
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import fissure_segmentation_amd as fsg
F = fsg.functional
dev = torch.device("cuda:0")


def replay_us(fn, reps=30, inner=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner):
            fn()
    for _ in range(3):
        g.replay()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        ts.append(1e3 * e0.elapsed_time(e1) / reps / inner)
    return float(np.median(ts))


shapes = [(16384, 1280, 192), (16384, 1024, 192), (16384, 256, 192), (16384, 256, 256), (16384, 128, 256), (16384, 192, 448),
          (16384, 192, 256), (32768, 1280, 192), (32768, 256, 256), (16384, 512, 512), (16384, 1024, 1024)]
for (M, N, K) in shapes:
    torch.manual_seed(0)
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.1
    img = F.pw_weight_image(w)
    flop = 2.0 * M * N * K
    t_v = replay_us(lambda: torch.nn.functional.linear(a, w))
    t_img = replay_us(lambda: F.pw_weight_image(w))
    row = "%6d x %4d x %3d  vendor fp32 %7.1f us %6.1f TF | image %5.1f us |" % (M, N, K, t_v, flop / t_v * 1e-6, t_img)
    for tile, name in ((1, "128x128 sb"), (11, "128x128 db"), (2, "64x128 db"), (12, "64x128 sb"), (4, "128x64 db"), (14, "128x64 sb"),
                       (3, "64x64 db"), (13, "64x64 sb")):
        t = replay_us(lambda: F.pw_linear(a, img, N, tile=tile))
        row += " %s %6.1f us %6.1f TF |" % (name, t, flop / t * 1e-6)
    print(row, flush=True)
