"""Test-time ensembling `predict_full_pointcloud` (models/point_seg_net.py:21-48 of the reference -- the call the reference
itself times, train.py:383-392, as seconds per case): 50 runs of a `sample_points`-subset of one full cloud.

Times the batched eval-mode form (all runs of a phase in one forward + fsg_ensemble_accumulate_f32) against the
reference's sequential loop on the same kernels, same generator state, and prints one JSON line per model.

    python tools/bench_infer.py [--points 32768] [--sample 2048] [--runs 50] [--reps 5]
"""
import argparse
import contextlib
import io
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fissure_segmentation_amd as fsg  # noqa: E402,F401
from fissure_segmentation_amd.models.dgcnn import DGCNNSeg  # noqa: E402
from fissure_segmentation_amd.models.pointtransformer.seg_model import PointTransformerCompatibility  # noqa: E402


def timed(fn, reps):
    """median of `reps` calls, each bracketed by HIP events (the eager path is host-launched: single calls stall for tens
    of milliseconds now and then on a shared host, a mean would report those)"""
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:      # warm up: clocks, allocator, kernel attributes
        out = fn()
        torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2], out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=32768)
    ap.add_argument("--sample", type=int, default=2048)
    ap.add_argument("--runs", type=int, default=50)
    ap.add_argument("--reps", type=int, default=9)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1234)
    pc = (torch.rand(1, 3, args.points, generator=g) * 2 - 1).to(dev)
    models = {
        "DGCNN-seg k=20 dynamic graph (cli defaults)": lambda: DGCNNSeg(k=20, in_features=3, num_classes=4),
        "DGCNN-seg k=40 static graph (run_dgcnn_seg_experiments.sh:17)": lambda: DGCNNSeg(k=40, in_features=3, num_classes=4,
                                                                                         dynamic=False),
        "PointTransformer-seg": lambda: PointTransformerCompatibility(3, 4),
    }
    for name, make in models.items():
        torch.manual_seed(0)
        net = make().to(dev).eval()

        def run():
            torch.manual_seed(7)
            with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
                return net.predict_full_pointcloud(pc, sample_points=args.sample, n_runs_min=args.runs)
        ms_b, out_b = timed(run, args.reps)
        net._ensemble_batchable = lambda _pc: False
        ms_s, out_s = timed(run, args.reps)
        print(json.dumps({"metric": "seconds per case, predict_full_pointcloud", "model": name, "points": args.points,
                          "sample_points": args.sample, "n_runs_min": args.runs, "dtype": "f32",
                          "batched_ms": round(ms_b, 3), "sequential_loop_ms": round(ms_s, 3),
                          "speedup": round(ms_s / ms_b, 2),
                          "max_abs_diff_probabilities": float((out_b - out_s).abs().max())}))


if __name__ == "__main__":
    main()
