"""In-kernel cycle breakdown of the production kNN kernel (one wave of every workgroup, s_memtime stamps).  Needs the
instrumented source (python tools/knn_instrument.py) built with -DFSG_KNN_STATS; revert with
git checkout afterwards.  Prints average stamped cycles per workgroup for each section, for the first and the last wave."""
import ctypes, sys, torch
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import fissure_segmentation_amd as fsg
from golden_util import cloud
lib = fsg._lib.lib
dev = torch.device("cuda:0")
names = {8: "phase A (own tiles)", 9: "barrier after A", 10: "B: row read", 11: "B: threshold", 12: "B: count+scan+stores",
         13: "B: rank", 14: "B: tail", 15: "barrier after B"}
for (B, C, N, k, kw) in [(8, 64, 2048, 20, 0), (8, 64, 2048, 20, 3), (8, 64, 2048, 20, 12), (8, 64, 2048, 20, 15), (4, 64, 8192, 40, 0), (4, 64, 8192, 40, 15)]:
    x = torch.from_numpy(cloud(1, B, C, N)).to(dev)
    fl = kw << 24
    out = (ctypes.c_ulonglong * 32)()
    for _ in range(3):
        fsg.functional.knn_graph(x, k, _debug_flags=fl)
    torch.cuda.synchronize()
    lib.fsg_debug_knn_stats(out, 1)
    fsg.functional.knn_graph(x, k, _debug_flags=fl); torch.cuda.synchronize()
    lib.fsg_debug_knn_stats(out, 1)
    v = list(out)
    wgs = B * ((N + 31) // 32)
    tot = sum(v[i] for i in names)
    print(f"B={B} C={C} N={N} k={k} wave {kw}: {wgs} workgroups, {tot / wgs:.0f} stamped cycles per workgroup (s_memtime at 100 MHz units?)")
    for i, n in names.items():
        print(f"   {n:26s} {v[i] / wgs:10.1f}  {100 * v[i] / max(tot, 1):5.1f} %")
    print(f"   survivors per ranked row {v[16] / max(v[17], 1):.1f}")
