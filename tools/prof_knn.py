import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import fissure_segmentation_amd as fsg
from golden_util import cloud
dev = torch.device("cuda:0")
for C in (3, 64):
    x = torch.from_numpy(cloud(1, 8, C, 2048)).to(dev)
    for _ in range(3):
        fsg.functional.knn_graph(x, 20)
torch.cuda.synchronize()
