"""kNN at config-2 shape, 64 channels; FSG_PROF_FLAGS selects the ablation (256: phase A only, 512: phase B only)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import fissure_segmentation_amd as fsg
from golden_util import cloud
dev = torch.device("cuda:0")
flags = int(os.environ.get("FSG_PROF_FLAGS", "0"))
C = int(os.environ.get("FSG_PROF_C", "64"))
x = torch.from_numpy(cloud(1, 8, C, 2048)).to(dev)
for _ in range(5):
    fsg.functional.knn_graph(x, 20, _debug_flags=flags)
torch.cuda.synchronize()
