"""fsg_pw_linear_f32 alone, for rocprofv3 --pmc passes: python tools/pw_prof.py M N K tile [launches]"""
import sys
import torch
sys.path[:0] = ["/root/repo"]
import fissure_segmentation_amd as fsg
M, N, K, tile = (int(v) for v in sys.argv[1:5])
n = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = torch.device("cuda:0")
torch.manual_seed(0)
a = torch.randn(M, K, device=dev)
w = torch.randn(N, K, device=dev) * 0.1
img = fsg.functional.pw_weight_image(w)
for _ in range(n):
    fsg.functional.pw_linear(a, img, N, tile=tile)
torch.cuda.synchronize()
