"""fsg_pw_linear_f32 as the FIRST GPU work of a process (what test_pw_linear_is_fp32_grade does), then repeated: where do the
outputs differ from torch's?  (diagnosis of an intermittent failure of that test)"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fissure_segmentation_amd as fsg
F = fsg.functional
dev = torch.device("cuda:0")
which = int(sys.argv[1]) if len(sys.argv) > 1 else 0
(M, N, K, tile) = [(16384, 1024, 192, 1), (16384, 256, 256, 2), (16384, 128, 256, 3)][which]
g = np.random.default_rng(M + N + K)
a = torch.from_numpy((g.standard_normal((M, K + 8)) * 10.0 ** g.uniform(-3, 2, (M, K + 8))).astype(np.float32)).to(dev)[:, :K]
w = torch.from_numpy((g.standard_normal((N, K + 5)) * 10.0 ** g.uniform(-3, 1, (N, K + 5))).astype(np.float32)).to(dev)[:, :K]
b = torch.from_numpy(g.standard_normal(N).astype(np.float32)).to(dev)
ys, imgs = [], []
for it in range(4):
    img = F.pw_weight_image(w)
    ys.append(F.pw_linear(a, img, N, bias=b, tile=tile))
    imgs.append(img)
torch.cuda.synchronize()
ref = (a.double() @ w.double().t() + b.double())
mag = a.double().abs() @ w.double().abs().t() + b.double().abs()
print((M, N, K, tile), "images equal:", [bool(torch.equal(imgs[0], im)) for im in imgs[1:]])
for it, y in enumerate(ys):
    e = ((y.double() - ref).abs() / mag)
    bad = (e > 1e-5).nonzero()
    print("  run", it, "max", float(e.max()), "bad", bad.shape[0], "row blocks", sorted(set((bad[:, 0] // 64 * 64).tolist()))[:8],
          "col blocks", sorted(set((bad[:, 1] // 32 * 32).tolist()))[:12])
    if bad.shape[0]:
        r, c = bad[0].tolist()
        print("    first bad", (r, c), float(y[r, c]), float(ref[r, c]))
