"""Headline benchmark: points/sec, forward + loss + backward + Adam step, DGCNN-seg N=2048 k=20
(BASELINE.json configs[1]) on N GPUs of one node, one process per GPU, batch sharded data-parallel with
an RCCL gradient average.  Prints ONE JSON line on rank 0.

    python bench.py [--gpus 1] [--steps 30] [--warmup 5] [--workload c2|c4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL's hipIpcGetMemHandle fails with the legacy mode); the GPU boxes
# export this already -- set it before the HIP runtime starts in case a launcher drops it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import fissure_segmentation_amd as fsg  # noqa: E402
from fissure_segmentation_amd import _lib, distributed as D  # noqa: E402
from fissure_segmentation_amd.losses.nnu_loss import NNULoss  # noqa: E402
from fissure_segmentation_amd.optim import FlatAdam  # noqa: E402
from fissure_segmentation_amd.models.dgcnn import DGCNNSeg  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 matrix peak (v_mfma_f32_16x16x4_f32 / 32x32x2, exact fp32)
L2_GATHER_PEAK = 17.8e12       # MI355X_MICROARCH.md, "Indexed rows": rows served from the XCD's L2, 66-73 GB/s per CU = 16.8-18.8 TB/s
WORKLOADS = {
    # name: (clouds per GPU, points, k, description)
    "c2": (8, 2048, 20, "DGCNN-seg N=2048 k=20, 8 clouds/GPU, dynamic graph, fp32 (BASELINE configs[1])"),
    "c4": (4, 8192, 40, "DGCNN-seg N=8192 k=40, 4 clouds/GPU, dynamic graph (BASELINE configs[3] shape, fp32)"),
    # secondary rows, same step definition and JSON line, no roofline object:
    "c3": (8, 2048, None, "PointTransformer seg N=2048, 8 clouds/GPU, nsample 8/16 (BASELINE configs[2] shape, fp32)"),
    "c5": (8, 4096, 20, "PC-AE DGCNNFoldingNet + Chamfer N=4096, 8 clouds/GPU, k=20 (BASELINE configs[4] shape, fp32)"),
    # SURVEY 8(d) secondary rows
    "c2s": (32, 2048, 40, "DGCNN-seg N=2048 k=40 STATIC graph, 32 clouds/GPU, fp32 (bash_scripts/run_dgcnn_seg_experiments.sh:17)"),
    "c3f": (8, 2048, None, "PointTransformer seg N=2048, 8 clouds/GPU, in_features=128 (3 coordinates + 125 features), fp32"),
    "c3b": (32, 2048, None, "PointTransformer seg N=2048, 32 clouds/GPU (bash_scripts/run_PointTransformer_experiments.sh:5), fp32"),
}
METRIC = {"c2": "points/sec fwd+bwd DGCNN-seg N=2048 k=20", "c4": "points/sec fwd+bwd DGCNN-seg N=8192 k=40",
          "c3": "points/sec fwd+bwd PointTransformer-seg N=2048", "c5": "points/sec fwd+bwd PC-AE FoldingNet+Chamfer N=4096",
          "c2s": "points/sec fwd+bwd DGCNN-seg N=2048 k=40 static", "c3f": "points/sec fwd+bwd PointTransformer-seg N=2048 128 features",
          "c3b": "points/sec fwd+bwd PointTransformer-seg N=2048 batch 32"}
EDGE_LAYERS_C = (3, 64, 64)  # input channels of ec1/ec2/ec3 (models/dgcnn.py:130-132 of the reference)


def synthetic_batch(B, N, classes, seed, device, inputs="uniform", features=3):
    """coordinates U(-1,1) ("uniform") or points on a noisy unit sphere / plane, sigma 0.01 ("surface": kNN
    rejection rates depend on the distribution); extra feature channels N(0,1); labels uniform."""
    g = torch.Generator().manual_seed(seed)
    if inputs == "surface":
        u = torch.randn(B, 3, N, generator=g)
        x = u / u.norm(dim=1, keepdim=True)                        # unit sphere
        half = B // 2
        if half:                                                   # every other cloud: the plane z = 0
            x[:half] = torch.rand(half, 3, N, generator=g) * 2 - 1
            x[:half, 2] = 0
        x = x + 0.01 * torch.randn(B, 3, N, generator=g)
    else:
        x = torch.rand(B, 3, N, generator=g) * 2 - 1
    if features > 3:
        x = torch.cat([x, torch.randn(B, features - 3, N, generator=g)], 1)
    y = torch.randint(0, classes, (B, N), generator=g)
    return x.to(device), y.to(device)


def knn_gather_bytes_per_point(k, s=4):
    """reference-semantic kNN+gather, per point: sum over layers of 4C + 4k + 2*C*k*s (fwd)."""
    return sum(4 * c + 4 * k + 2 * c * k * s for c in EDGE_LAYERS_C)


def knn_gather_min_bytes_per_point(k, s=4, cout=64):
    """the fused design's own minimal traffic, per point and layer 4C + 4k + Cout*s (the edge tensor is never
    written)."""
    return sum(4 * c + 4 * k + cout * s for c in EDGE_LAYERS_C)


def usable_cores():
    """Threads the CPU baseline may use: CPU affinity, capped by the cgroup quota and by the 16-core share a
    one-GPU box grants (more threads than the quota only adds throttling; FSG_CPU_THREADS overrides)."""
    if os.environ.get("FSG_CPU_THREADS"):
        return int(os.environ["FSG_CPU_THREADS"])
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_baseline(B, N, k, classes, budget_s=12.0, max_steps=8):
    """The oracle's pure-PyTorch CPU restatement (kind "port") on the host cores, same step definition; a bounded sample:
    whole steps of the same batch (callers pass fewer clouds for the N = 8192 workload: a step there materialises
    (B,N,N) and (B,2C,N,k) tensors) until ~budget_s seconds of CPU work are spent (at least 2, at most max_steps)."""
    from oracle import ref_cpu
    cores = usable_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    net = ref_cpu.DGCNNSeg(k=k, in_features=3, num_classes=classes).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    x, y = synthetic_batch(B, N, classes, 1234, "cpu")
    crit = ref_cpu.NNULoss(torch.tensor([0.4, 1.2, 1.2, 1.2][:classes]))

    def step():
        opt.zero_grad()
        crit(net(x), y)[0].backward()
        opt.step()
    step()
    t0 = time.perf_counter()
    steps = 0
    while steps < 2 or (steps < max_steps and time.perf_counter() - t0 < budget_s):
        step()
        steps += 1
    dt = (time.perf_counter() - t0) / steps
    return {"value": B * N / dt, "unit": "points/s", "cores": cores, "kind": "port",
            "sample": f"{steps} steps of the same workload (B={B}, N={N}, k={k}) after 1 warm-up, "
                      f"oracle/ref_cpu.DGCNNSeg fwd+CE+GDL+bwd+Adam, {dt:.2f} s/step"}


def _dump_memmap(path, net, opt):
    """address ranges of the step's long-lived buffers and of every allocator segment (with its graph pool), written
    before the first replay: a GPU memory fault names an address, this names its owner"""
    def rng(t):
        return [hex(t.data_ptr()), hex(t.data_ptr() + t.numel() * t.element_size())]
    out = {"tensors": {}, "segments": []}
    if isinstance(opt, FlatAdam):
        out["tensors"].update({"flat.data": rng(opt.flat.data), "flat.grad": rng(opt.flat.grad)})
        if opt.inner is None:
            out["tensors"].update({"exp_avg": rng(opt.exp_avg), "exp_avg_sq": rng(opt.exp_avg_sq)})
    for n, p in net.named_parameters():
        if p.grad is not None:
            out["tensors"]["grad:" + n] = rng(p.grad)
    for seg in torch.cuda.memory_snapshot():
        out["segments"].append({"address": hex(seg["address"]), "end": hex(seg["address"] + seg["total_size"]),
                                "pool": str(seg.get("segment_pool_id")), "type": seg.get("segment_type"),
                                "allocated": seg.get("allocated_size")})
    with open(path, "w") as f:
        json.dump(out, f, indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inputs", default="uniform", choices=["uniform", "surface"], help="synthetic input set (SURVEY 8d)")
    ap.add_argument("--dtype", default=None, choices=["f32", "bf16"],
                    help="MFMA operand type of the dense contractions (functional.set_mfma_operands); default: what "
                         "BASELINE names: fp32 for config 2, bf16 for configs 3 and 4; config 5 has no bf16 kernel and runs fp32")
    ap.add_argument("--min-seconds", type=float, default=3.0,
                    help="repeat the timed block of --steps steps until the GPU has been busy this long; the MEDIAN block is "
                         "reported (ms_per_step, value), every block is bracketed by barrier + synchronize")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python instead of replaying a hipGraph")
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam over the separate tensors instead of FlatAdam")
    ap.add_argument("--grad-sync", default="flat", choices=["flat", "bucketed"],
                    help="N>1: all-reduce FlatAdam's flat gradient buffer in place (one collective, no copies) or the "
                         "bucketed averager (cat -> all-reduce -> foreach copy back)")
    ap.add_argument("--dump-check", default=None,
                    help="after the timed steps run ONE more fwd/bwd + gradient average and write rank 0's parameters and "
                         "averaged gradient to this .npz (tests/test_gpu_parity.py compares it with the oracle's mean of "
                         "shard gradients)")
    args = ap.parse_args()

    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)  # warm-up and capture use side streams
    rank, world, device = D.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if device.type != "cuda":
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    ranks_seen = 1
    if world > 1:
        # the process group the gradients will cross: RCCL (torch's "nccl" backend on ROCm) unless a rehearsal names another
        # one explicitly, and every rank reachable -- a ones-tensor summed over the group must come back as `world`
        backend = dist.get_backend()
        if backend != "nccl" and not os.environ.get("FSG_DIST_BACKEND"):
            raise SystemExit(f"world_size {world} on GPUs needs the nccl (RCCL) backend, got {backend!r}; "
                             "set FSG_DIST_BACKEND to rehearse over another one")
        probe = torch.ones(1, device=device)
        dist.all_reduce(probe, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        ranks_seen = int(round(float(probe.item())))
        if ranks_seen != world:
            raise SystemExit(f"all-reduce of ones returned {ranks_seen}, expected {world}: the process group is incomplete")
    B, N, k, desc = WORKLOADS[args.workload]
    classes = 4
    # bf16 exists where a hand-written dense contraction exists: the two-layer EdgeConv of DGCNN-seg (configs 2 / 4) and the
    # nn.Linear products of the PointTransformer (config 3: fsg_gemm_small_bf16 forward, dX and dW; the layer's c -> c/8
    # contraction, BatchNorm statistics and all stored tensors stay fp32).  The PC-AE encoder (one-layer EdgeConvs: a per-POINT
    # GEMM) has no bf16 kernel, so config 5 runs -- and is labelled -- fp32 (the reference disables autocast there too).
    dtype = args.dtype or ("bf16" if args.workload in ("c4", "c3", "c3f", "c3b") else "f32")
    fsg.functional.set_mfma_operands(dtype)      # graph build, BatchNorm statistics and stored tensors stay fp32 either way
    if dtype == "bf16" and args.workload in ("c3", "c3f", "c3b"):
        # bf16 operands for the nn.Linear products: parity held after 20 Adam steps (mean |logit error| 4e-3, gradient cosine
        # 0.985 against the fp32 oracle: tests/test_gpu_parity.py), 6.71 against 6.84 ms per step
        fsg.functional.set_bf16_linear(True)
    desc = desc.replace("fp32", "bf16 MFMA operands" if dtype == "bf16" else "fp32")

    torch.manual_seed(0)
    dgcnn = args.workload in ("c2", "c4", "c2s")
    features = 128 if args.workload == "c3f" else 3
    if dgcnn:
        net = DGCNNSeg(k=k, in_features=3, num_classes=classes, dynamic=args.workload != "c2s").to(device).train()
    elif args.workload in ("c3", "c3f", "c3b"):
        from fissure_segmentation_amd.models.pointtransformer.seg_model import PointTransformerCompatibility
        net = PointTransformerCompatibility(features, classes).to(device).train()
    else:
        from fissure_segmentation_amd.losses.chamfer_loss import ChamferLoss
        from fissure_segmentation_amd.models.folding_net import DGCNNFoldingNet
        net = DGCNNFoldingNet(k=k, n_embedding=512, shape_type="plane", n_input_points=N, decode_mesh=True).to(device).train()
    D.broadcast_parameters(net)
    use_graph = not args.eager
    # eager mode overlaps the head bucket's all-reduce with the EdgeConv backward through autograd hooks; under hipGraph
    # replay no Python runs inside the step, so the gradients go out as one bucket between the two graphs
    averager = D.BucketedGradAverager(
        net, early=(lambda n: False) if use_graph else
        (lambda n: dgcnn and (n.startswith("segmentation") or n.startswith("global_feature"))))
    # fused=True: one multi-tensor kernel for the whole model (the foreach/capturable path issues ~65 tiny kernels)
    if args.torch_adam:
        opt = torch.optim.Adam(net.parameters(), lr=1e-3, capturable=use_graph, fused=True)
    else:   # same update, parameters re-pointed into one flat buffer: one cat + one fused kernel per step (optim.py)
        opt = FlatAdam(net.parameters(), lr=1e-3, capturable=use_graph)
    x, y = synthetic_batch(B, N, classes, 1234 + rank, device, args.inputs, features)
    # the criterion train.py:38 builds by default (--loss nnunet, cli_args.py:16-17): class-weighted cross-entropy +
    # generalised Dice, here on the fused HIP loss kernel; weights as ds.get_class_weights() would hand over (train.py:34)
    criterion = NNULoss(torch.tensor([0.4, 1.2, 1.2, 1.2][:classes])).to(device)
    if args.workload == "c5":     # train_pc_ae.py:172-185: reconstruct the input cloud, Chamfer loss (target = input)
        chamfer = ChamferLoss()

    def fwd_bwd():
        averager.zero_grad()
        if args.workload == "c5":
            loss = chamfer(net(x), x)
        else:
            loss, _parts = criterion(net(x), y)
        loss.backward()
        return loss

    # N>1 with FlatAdam: the gradients are gathered into the optimizer's flat buffer (one cat, the tail of the fwd/bwd
    # graph), that buffer is all-reduced IN PLACE (one collective over 2.5 MB, nothing copied back), and the optimizer
    # graph scales by 1/world and updates.  The buffer is allocated by FlatAdam's constructor -- before any capture, from
    # the ordinary allocator -- and lives as long as the optimizer; the two graphs do not share a memory pool.
    flat_sync = world > 1 and isinstance(opt, FlatAdam) and args.grad_sync == "flat"

    def sync_grads():
        if flat_sync:
            # synchronous call on the CURRENT (replay) stream: RCCL enqueues the collective stream-ordered after graph 1 and
            # before graph 2, the host does not wait for it
            dist.all_reduce(opt.flat.grad, op=dist.ReduceOp.SUM, async_op=False)
        else:
            averager.finish()

    def apply_grads():
        if flat_sync:
            opt.flat.grad.mul_(1.0 / world)
            opt.step_flat()
        else:
            opt.step()

    def eager_step():
        loss = fwd_bwd()
        if flat_sync:
            opt.gather_grads()
        sync_grads()
        apply_grads()
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(max(args.warmup, 3)):   # also sets every kernel attribute / autotune choice before any capture
        eager_step()
    fence()
    launch = "eager"
    step = eager_step
    breakdown, step_no = D.StepBreakdown(on_gpu=True), [0]
    if use_graph:
        # the step has static shapes: capture fwd+loss+bwd (and, on one GPU, Adam) once into a hipGraph and replay it;
        # with data parallelism the gradient all-reduce runs between the fwd/bwd graph and the optimizer graph
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    eager_step()
            torch.cuda.current_stream().wait_stream(side)
            g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            # with a process group alive its watchdog thread issues runtime calls of its own: capture in thread-local mode so
            # that only this thread's calls are checked against the capture
            cmode = os.environ.get("FSG_CAPTURE_MODE", "thread_local" if world > 1 else "global")
            with torch.cuda.graph(g1, capture_error_mode=cmode):
                static_loss = fwd_bwd()
                if world == 1:
                    opt.step()
                elif flat_sync:
                    opt.gather_grads()
            if world > 1:
                with torch.cuda.graph(g2, capture_error_mode=cmode):     # its own pool: nothing of g1's is recycled
                    apply_grads()
            if os.environ.get("FSG_DUMP_MEMMAP"):
                _dump_memmap(os.environ["FSG_DUMP_MEMMAP"] + f".rank{rank}.json", net, opt)

            def graph_step():
                if flat_sync:     # graph 1 -> in-place all-reduce of flat.grad on the replay stream -> graph 2
                    step_no[0] += 1
                    D.flat_sync_step(g1.replay, opt.flat.grad, g2.replay,
                                     probe=breakdown if step_no[0] % 10 == 0 else None)   # HIP events every 10th step
                    return static_loss
                g1.replay()
                if world > 1:
                    sync_grads()
                    g2.replay()
                return static_loss
            step, launch = graph_step, "hipGraph replay"
            for _ in range(2):
                step()
            fence()
        except Exception as e:  # capture is an optimisation, never a requirement
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            step, launch = eager_step, "eager"
            torch.cuda.synchronize()

    # EXACTLY --steps steps per timed block, barrier + synchronize on both sides; the block is repeated until --min-seconds
    # of GPU time have been spent (same count on every rank) and the MEDIAN block is the reported one
    def timed_block():
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out_loss = step()
        fence()
        return time.perf_counter() - t0, out_loss
    first, loss = timed_block()
    nblocks = torch.tensor([max(1, min(400, int(args.min_seconds / max(first, 1e-6))))], device=device)
    if world > 1:
        dist.all_reduce(nblocks, op=dist.ReduceOp.MAX)
    blocks = [first]
    for _ in range(int(nblocks.item()) - 1):
        dt, loss = timed_block()
        blocks.append(dt)
    # per-kernel durations: HIP events around every C-ABI call in a few EAGER steps of the same workload (events cannot
    # be read back from inside a replayed graph); the rocprofv3 summary under profiles/ covers the replayed region
    _lib.start_timing()
    n_timed = 5
    for _ in range(n_timed):
        eager_step()
    kernel_ms = _lib.stop_timing()
    # the headline kernel group (forward graph build + neighbour gather of the three EdgeConv layers) on its own: captured
    # into a hipGraph and replayed between two HIP events, so the time carries no Python launch gaps -- exactly what runs
    # inside the replayed training step
    def replay_us(fn, reps=50):
        """fn captured into a hipGraph and replayed between two HIP events on the replay stream: kernel time without Python
        launch gaps -- what the same launches cost inside the replayed training step"""
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        for _ in range(3):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / reps

    group_us, knn64_us, knn_prepared, knn_nominees = None, None, False, None
    try:
        if not dgcnn:
            raise LookupError("no EdgeConv group in this workload")

        def group():     # the three EdgeConv layers exactly as DGCNNSeg.forward runs them (DGCNNSeg.edge_levels)
            with torch.no_grad():
                if not net.dynamic:
                    net.knn_graph = fsg.functional.knn_graph(x, net.k, c_knn=3, fix_diag=True, drop_first=True)
                return net.edge_levels(x)[0].transpose(1, 2).contiguous()
        feat = group()
        group_us = replay_us(group, 20)
        if net.dynamic:   # the dominant kernel on its own: one feature-space graph build (64 channels), 10 launches per replay
            F_ = fsg.functional
            ws_k = F_.knn_prep_workspace(B, N, 64, device)
            if ws_k is not None and net.ec1.fused:
                # as inside the step: prepared by the EdgeConv that produces the features (its apply pass emits the prep
                # products), the build itself = the main kernel
                with torch.no_grad():
                    g0 = F_.knn_graph(x, k, c_knn=3, fix_diag=True)
                    feat, feat_pm = net.ec1(x, g0, both=True, knn_ws=ws_k)
                knn_prepared = True

                def knn10():
                    for _ in range(10):
                        F_.knn_graph(feat, k, prepared=(ws_k, feat_pm))
            else:
                knn_prepared = False

                def knn10():
                    for _ in range(10):
                        F_.knn_graph(feat, k)
            knn64_us = replay_us(knn10, 20) / 10
            try:     # nominees per query of that build, from the kernel's own counters (debug flag 33554432)
                import ctypes
                st_ = (ctypes.c_ulonglong * 4)()
                _lib.lib.fsg_debug_knn_split_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]
                _lib.lib.fsg_debug_knn_split_stats(st_, 1)
                F_.knn_graph(feat, k, _debug_flags=33554432)
                torch.cuda.synchronize()
                _lib.lib.fsg_debug_knn_split_stats(st_, 0)
                knn_nominees = st_[1] / max(st_[0], 1)
            except Exception as e:
                print(f"[bench] nominee count failed ({type(e).__name__}: {e})", file=sys.stderr)
    except LookupError:
        pass
    except Exception as e:
        print(f"[bench] group timing by graph replay failed ({type(e).__name__}: {e})", file=sys.stderr)
    head_us, vendor_us = None, None
    if dgcnn and rank == 0:
        # the widest product of the fused point-wise head on its own (levels (B N, 192) x [W_global ; W0_levels]^T (1280, 192),
        # csrc/pointwise.hip): plain member of the kernel family, same tile and loop as inside the step
        try:
            a_h = torch.randn(B * N, 192, device=device)
            w_h = torch.randn(1280, 192, device=device) * 0.1
            img_h = fsg.functional.pw_weight_image(w_h)
            head_us = replay_us(lambda: fsg.functional.pw_linear(a_h, img_h, 1280, tile=1), 20)
            if not args.no_cpu_baseline:     # (comparison legs stay out of the profiled runs: no stray vendor kernel in their traces)
                w_t = w_h.t().contiguous()
                vendor_us = replay_us(lambda: torch.mm(a_h, w_t), 20)     # the vendor library's fp32 GEMM at the same shape, same run
        except Exception as e:
            print(f"[bench] head product timing failed ({type(e).__name__}: {e})", file=sys.stderr)
    if not torch.isfinite(loss):
        raise SystemExit("non-finite loss")
    if args.dump_check:
        # one more fwd/bwd + gradient average through the SAME launch path as the timed steps, without the update
        import numpy as np
        params = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).cpu().numpy()
        def local_dump():   # every rank's OWN gradient, before the collective: the average must be their mean, exactly
            loc = opt.flat.grad if flat_sync else torch.cat([p.grad.reshape(-1) for p in net.parameters()])
            np.save(args.dump_check + f".rank{rank}.npy", loc.detach().cpu().numpy())
        if launch == "hipGraph replay":
            g1.replay()
            local_dump()
            if world > 1:
                sync_grads()
        else:
            fwd_bwd()
            if flat_sync:
                opt.gather_grads()
            local_dump()
            if world > 1:
                sync_grads()
        if flat_sync:
            avg = (opt.flat.grad / world).cpu().numpy()
        else:
            avg = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu().numpy()
        torch.cuda.synchronize()
        if rank == 0:
            np.savez(args.dump_check, params=params, avg_grad=avg, world=world, B=B, N=N, k=k or 0, launch=launch,
                     grad_sync="flat" if flat_sync else "bucketed")
    # N > 1: where the step's time goes (HIP events around graph 1 / the all-reduce / graph 2 on every 10th step), max over ranks
    breakdown_us = None
    if world > 1:
        torch.cuda.synchronize()
        mean = breakdown.mean_us()
        bt = torch.tensor([mean[n] for n in D.StepBreakdown.NAMES] if mean else [0.0, 0.0, 0.0], dtype=torch.float64, device=device)
        dist.all_reduce(bt, op=dist.ReduceOp.MAX)
        if mean:
            breakdown_us = {n: round(v, 1) for n, v in zip(D.StepBreakdown.NAMES, bt.tolist())}
            breakdown_us["probed_steps"] = len(breakdown.steps)
            breakdown_us["reduced_as"] = "mean over the probed steps per rank, maximum over the ranks"
    t = torch.tensor(blocks, dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)       # per block: the slowest rank
    blocks = sorted(t.tolist())
    elapsed = blocks[len(blocks) // 2]                 # median block of --steps steps

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        per_kernel = {}
        for name, vals in kernel_ms.items():
            per_kernel[name] = {"launches_per_step": len(vals) / n_timed, "avg_us": 1e3 * sum(vals) / len(vals)}
        traffic_all = {}
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                traffic_all = json.load(f)
        roofline, roofline_group = None, None
        # (the coordinate build of DGCNN-seg goes through the entry that also emits the first block's [P | Q] rows)
        knn_calls = kernel_ms.get("fsg_knn_dense_ws_f32", []) + kernel_ms.get("fsg_knn_dense_ws_pq_f32", [])
        knn_prep_calls = kernel_ms.get("fsg_knn_dense_prepared_f32", [])   # feature-space builds prepared by their producer
        if knn_calls and args.workload != "c2s":
            # DOMINANT KERNEL GROUP of the DGCNN-type workloads: the feature-space graph build (csrc/knn_split.hip, two launches:
            # knn_nominate_kernel + knn_refine_kernel).  Its compulsory HBM traffic is tiny (4C + 4k bytes per point), so the
            # ceilings that can bind are on-chip.  It is priced against what it EXECUTES / MOVES, every term derived in THIS run
            # (shapes, the nominee count read back from the kernel's own counters, the time by HIP events), none can be exceeded:
            #   mfma_f16    issued matrix flop (two coarse sweeps on v_mfma_f32_32x32x16_f16 = 2 x 2 B N^2 Cpad) / 2.5 PFLOP/s
            #   refine_f32  exact fp32 fma chains of the nominated candidates (B N x nominees x 2C) / 157.3 TFLOP/s
            #   l2_to_cu    bytes every CU has to pull out of its XCD's L2: each 64-query workgroup reads the cloud's whole coarse
            #               image once (operand tiles resident for both sweeps at N = 2048, twice otherwise) + the fp32 rows of the
            #               nominees, against the chip's L2 -> CU gather rate (MI355X_MICROARCH.md, "Indexed rows": 66-73 GB/s per
            #               CU from the XCD's L2 = 16.8-18.8 TB/s; 17.8 used)
            # `frac` is the largest and `bound` names it.  candidates_per_s (B N^2 / t) is a throughput figure, not a fraction.
            chans = {"c5": (3, 64, 64, 128)}.get(args.workload, EDGE_LAYERS_C)
            if knn_prep_calls:     # DGCNN-seg: build 1 (coordinates) through the plain entry, builds 2 and 3 prepared
                pp = len(knn_prep_calls) // n_timed
                by_layer = [knn_calls] + [[v for i, v in enumerate(knn_prep_calls) if i % pp == q] for q in range(pp)]
                per_step = 1 + pp
            else:
                per_step = len(knn_calls) // n_timed             # graph builds per step (3 for DGCNN-seg, 4 for the PC-AE)
                by_layer = [[v for i, v in enumerate(knn_calls) if i % per_step == li] for li in range(per_step)]
            # the dominant build: the channel count whose builds take the most time per step TOGETHER (DGCNN-seg: two 64-channel
            # builds against one on the coordinates -- single builds are within the eager timing's noise of each other), then the
            # most expensive build of that width
            avg_of = lambda q: sum(by_layer[q]) / max(len(by_layer[q]), 1)  # noqa: E731
            tot_by_c = {}
            for q in range(per_step):
                tot_by_c[chans[q]] = tot_by_c.get(chans[q], 0.0) + avg_of(q)
            c_dom = max(tot_by_c, key=tot_by_c.get)
            li = max((q for q in range(per_step) if chans[q] == c_dom), key=avg_of)
            avg_ms = sum(by_layer[li]) / len(by_layer[li])
            timed_as = "HIP events around the C-ABI entry point on its stream (eager steps of the same workload)"
            if knn64_us is not None and chans[li] == 64:
                avg_ms = knn64_us * 1e-3
                timed_as = ("HIP events around hipGraph replays of 10 back-to-back calls of the entry point on the replay "
                            "stream (" + ("fsg_knn_dense_prepared_f32: the prep products come out of the producing EdgeConv's "
                                          "apply pass, as inside the step" if knn_prepared else "prep kernel included") +
                            ", no Python launch gaps); eager per-call timing: "
                            f"{1e3 * sum(by_layer[li]) / len(by_layer[li]):.1f} us")
            t_s = avg_ms * 1e-3
            cpad = 4 if chans[li] <= 4 else 16 * -(-chans[li] // 16)
            products = 3 if chans[li] <= 4 else 1                        # two bf16 pieces (three products) up to 4 channels
            issued = 2.0 * products * 2.0 * B * N * N * max(cpad, 16)    # two sweeps, k padded to one 16-deep MFMA step
            nominees = knn_nominees if (knn_nominees and chans[li] == 64) else {20: 27.4, 40: 77.2}.get(k, float(k) * 1.5)
            image_bpp = 32.0 if chans[li] <= 4 else 2.0 * cpad           # coarse image bytes per point (1 KiB blocks per 32 points)
            image_reads = 1.0 if N == 2048 else 2.0                      # resident operand tiles at 64 tiles per cloud
            l2_bytes = B * (N / 64.0) * N * image_bpp * image_reads + B * N * nominees * cpad * 4.0
            ceilings = {"mfma_f16": round(issued / t_s / 2.5e15, 4),
                        "refine_f32": round(B * N * nominees * 2.0 * chans[li] / t_s / (MFMA_FP32_PEAK_TFLOPS * 1e12), 4),
                        "l2_to_cu": round(l2_bytes / t_s / L2_GATHER_PEAK, 4)}
            bound = max(ceilings, key=ceilings.get)
            knn_traffic = None
            if args.workload == "c2":
                kt = [rec.get("hbm_bytes_per_launch") for kname, rec in traffic_all.get("kernels", {}).items()
                      if kname.startswith("knn_nominate_kernel<4") or kname.startswith("knn_refine_kernel<64")]
                knn_traffic = sum(kt) if len(kt) == 2 else None
            unit = {"mfma_f16": ("TFLOP/s", issued / t_s / 1e12, 2500.0),
                    "refine_f32": ("TFLOP/s", ceilings["refine_f32"] * MFMA_FP32_PEAK_TFLOPS, MFMA_FP32_PEAK_TFLOPS),
                    "l2_to_cu": ("GB/s", l2_bytes / t_s / 1e9, L2_GATHER_PEAK / 1e9)}[bound]
            roofline = {"bound": {"mfma_f16": "mfma", "refine_f32": "mfma", "l2_to_cu": "l2"}[bound],
                        "achieved": round(unit[1], 2), "peak": round(unit[2], 1), "unit": unit[0], "frac": ceilings[bound],
                        "ceilings": ceilings,
                        "ceilings_derived_from": "shapes + the kernel's own nominee counter + HIP-event time, all of this run; peaks "
                                                 "from MI355X_MICROARCH.md (2.5 PFLOP/s f16 matrix, 157.3 TFLOP/s fp32, 17.8 TB/s L2 -> CU)",
                        "traffic": knn_traffic,
                        "traffic_source": "profiles/hbm_traffic.json (PMC passes of an earlier run of the same command), not measured in this run",
                        "kernel": f"feature-space graph build on {chans[li]} channels (csrc/knn_split.hip: " +
                                  ("knn_split_kernel, one launch -- the form kept for k > 32 at N > 4096" if (k > 32 and N > 4096 and chans[li] >= 64)
                                   else "knn_nominate_kernel + knn_refine_kernel") +
                                  "; its prep products are emitted by the producing EdgeConv's apply pass): two "
                                  "coarse sweeps on the matrix cores nominate candidates under a rigorous error bound; exact fp32 "
                                  "fma chains + ranking for the nominees (bit-identical to the fp32 oracle)",
                        "issued_matrix_flops_per_launch": issued, "l2_to_cu_bytes_per_launch": l2_bytes,
                        "nominees_per_query": round(nominees, 2),
                        "avg_us": round(1e3 * avg_ms, 1), "launches_per_step": per_step,
                        "candidates_per_s": round(B * N * N / t_s, 1),
                        "timed_as": timed_as}
        if dgcnn:
            # the north-star HBM view of the forward "kNN + gather" group, against BOTH byte counts of SURVEY 8(d)
            grp = ["fsg_knn_dense_ws_f32", "fsg_knn_dense_ws_pq_f32", "fsg_knn_dense_prepared_f32", "fsg_edge_gather_fwd_f32",
                   "fsg_edgeconv1_fwd_f32", "fsg_edgeconv2_fwd_f32", "fsg_edgeconv2_fwd_bf16", "fsg_edgeconv_apply_f32",
                   "fsg_edgeconv_apply_pq_f32"]   # (bf16 operand mode and the by-product variants: entries of their own)
            grp_ms = sum(sum(kernel_ms.get(n, [])) for n in grp) / n_timed
            ref_bytes = knn_gather_bytes_per_point(k) * B * N
            min_bytes = knn_gather_min_bytes_per_point(k) * B * N
            t_best = (group_us * 1e-6) if group_us else grp_ms * 1e-3
            tr = traffic_all.get(args.workload)
            roofline_group = {
                "bound": "mfma+latency (graph builds: %.0f of %.0f us); the gather/MLP stages alone are L2/HBM-bound" % (
                    1e3 * (sum(knn_calls) + sum(knn_prep_calls)) / n_timed, 1e3 * grp_ms),
                "kernel": "forward kNN graph + neighbour gather (+ fused edge MLP / BN / max) of the 3 EdgeConv layers: "
                          "fsg_knn_dense_ws[_pq]_f32 / fsg_knn_dense_prepared_f32 + fsg_edgeconv{1,2}_fwd_f32 + fsg_edgeconv_apply[_pq]_f32 "
                          "(the HIP-event sum leaves out the library GEMMs of the per-point products that are not fused yet; the "
                          "graph-replay figure times DGCNNSeg.edge_levels whole)",
                "unit": "GB/s", "peak": HBM_PEAK_GBS,
                "us_per_step_hip_events": round(1e3 * grp_ms, 1),
                "us_per_step_graph_replay": None if group_us is None else round(group_us, 1),
                "reference_materialised_bytes": ref_bytes, "own_minimal_bytes": min_bytes,
                "achieved_vs_reference_bytes": round(ref_bytes / t_best / 1e9, 1),
                "frac_vs_reference_bytes": round(ref_bytes / t_best / 1e9 / HBM_PEAK_GBS, 4),
                "achieved_vs_own_minimal_bytes": round(min_bytes / t_best / 1e9, 1),
                "frac_vs_own_minimal_bytes": round(min_bytes / t_best / 1e9 / HBM_PEAK_GBS, 4),
                "traffic": tr if isinstance(tr, (int, float)) else None,
                "traffic_source": "profiles/hbm_traffic.json (PMC passes of an earlier run), not measured in this run",
                "target": "north_star: >= 0.40 of the HBM roofline on the reference-materialised bytes"}
        if args.workload in ("c3", "c3f", "c3b"):
            # dominant kernel of the PointTransformer step: the fused vector-attention layer, backward (18 launches).
            # Algorithmic bytes = the tensors the reference materialises per layer (seg_model.py:37-52): grouped keys
            # (n,ns,3+c), grouped values (n,ns,c), p_r (n,ns,c), w before / after linear_w ((n,ns,c) + 2 (n,ns,c/8)), out;
            # written once and read once in the forward, the same again for their gradients in the backward.
            planes, ns, npts = [32, 64, 128, 256, 512], [8, 16, 16, 16, 16], [B * N // 4 ** i for i in range(5)]
            layers = [2, 3, 4, 6, 3]
            bytes_bwd = sum(L * 2 * 4 * n * m * (3 + c + c + c + c + 2 * c // 8) + L * 2 * 4 * n * c * 4
                            for L, c, m, n in zip(layers, planes, ns, npts))
            calls = kernel_ms.get("fsg_pt_attn_bwd_f32", [])
            if calls:
                t_step = sum(calls) / n_timed * 1e-3
                ach = bytes_bwd / t_step / 1e9
                roofline = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                            "kernel": "fsg_pt_attn_bwd_f32 (fused PointTransformerLayer backward), all 18 layers of a step; "
                                      "algorithmic bytes = the (n,ns,c) tensors the reference materialises, gradients included",
                            "algorithmic_bytes_per_step": bytes_bwd, "us_per_step": round(1e6 * t_step, 1),
                            "launches_per_step": len(calls) / n_timed,
                            "note": "the layer is latency-bound at these sizes (levels 3-5 hold 1024 / 256 / 64 points)"}
        out = {"metric": METRIC[args.workload], "value": round(B * N * world * args.steps / elapsed, 1),
               "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": dtype, "data": "synthetic",
               "precision": ("fp32 storage, fp32 accumulation, fp32-grade products everywhere.  Graph build: coarse fp16 / bf16 matrix "
                             "sweeps only NOMINATE, every ranked or returned distance is the oracle's fp32 fma chain (indices and distance "
                             "bits equal the CPU oracle's).  The per-edge 64 x 64 contraction of the two-layer EdgeConv (ec2s_fwd / "
                             "ec2s_bwd, csrc/edgeconv2.hip) and the point-wise head's products (csrc/pointwise.hip) run as three bf16 "
                             "pieces per fp32 operand and six v_mfma_f32_32x32x16_bf16 products each, fp32 accumulation (dropped terms "
                             "<= 2^-26 |a||b|; error vs fp64 at or below an fp32 fma chain's / the vendor fp32 GEMM's: "
                             "tests/test_gpu_parity.py::test_pw_linear_is_fp32_grade, ::test_ec2s_is_fp32_grade); one-layer EdgeConvs "
                             "are per-point fp32 products + fp32 gathers"
                             if dtype == "f32" else
                             ("bf16 operands / fp32 accumulation in every nn.Linear product of the PointTransformer -- forward, input and "
                              "weight gradients: operands rounded to bf16 inside fsg_gemm_small_bf16, v_mfma_f32_32x32x16_bf16 -- ; kNN, "
                              "farthest point sampling, BatchNorm statistics, the attention layer's c -> c/8 contraction, softmax, stored "
                              "activations and gradients fp32.  Parity at weights after 20 Adam steps: mean |logit error| 4e-3, "
                              "parameter-gradient cosine 0.985 against the fp32 oracle (the oracle under bf16 autocast: 8e-3 / 0.964)"
                              if args.workload in ("c3", "c3f", "c3b") else
                              "bf16 operands / fp32 accumulation in the per-edge EdgeConv contraction (v_mfma_f32_32x32x16_bf16) and "
                              "the vendor GEMMs; graph build, BatchNorm statistics, stored activations and gradients fp32")),
               "config": {"workload": desc, "clouds_per_gpu": B, "points_per_cloud": N, "k": k, "inputs": args.inputs,
                          "global_batch": B * world, "step": "fwd + (cross-entropy + generalised Dice) + bwd + grad all-reduce + Adam",
                          "launch": launch, "optimizer": "torch.optim.Adam(fused)" if args.torch_adam else
                          "Adam over one flat parameter buffer (optim.FlatAdam: fsg_adam_flat_f32, one launch)",
                          "parallelism": f"dp{world}",
                          "dist_backend": None if world == 1 else dist.get_backend(), "ranks_seen": ranks_seen,
                          "grad_sync": None if world == 1 else
                          ("one in-place all-reduce of FlatAdam's flat gradient buffer between the fwd/bwd graph and the "
                           "optimizer graph" if flat_sync else "bucketed averager (cat, all-reduce, copy back)"),
                          "allreduce_payload_bytes": None if world == 1 else (opt.flat.grad.numel() * 4 if flat_sync else None),
                          "step_breakdown_us": breakdown_us},
               "timing": {"blocks": len(blocks), "steps_per_block": args.steps, "reported": "median block",
                          "block_ms_min_median_max": [round(1e3 * blocks[0], 3), round(1e3 * elapsed, 3), round(1e3 * blocks[-1], 3)],
                          "gpu_busy_s": round(sum(blocks), 2)},
               "roofline": roofline, "roofline_group_hbm": roofline_group,
               "roofline_head_product": None if head_us is None else {
                   "bound": "mfma", "unit": "TFLOP/s", "peak": 2500.0,
                   "kernel": "pw_rowgemm_kernel (csrc/pointwise.hip) at the widest product of the fused head: (B N, 192) x (1280, 192)^T, "
                             "fp32-grade through three bf16 pieces per operand and six v_mfma_f32_32x32x16_bf16 products",
                   "avg_us": round(head_us, 1),
                   "issued_matrix_flops_per_launch": 6 * 2.0 * B * N * 1280 * 192,
                   "achieved": round(6 * 2.0 * B * N * 1280 * 192 / (head_us * 1e-6) / 1e12, 1),
                   "frac": round(6 * 2.0 * B * N * 1280 * 192 / (head_us * 1e-6) / 2.5e15, 4),
                   "fp32_equivalent_tflops": round(2.0 * B * N * 1280 * 192 / (head_us * 1e-6) / 1e12, 1),
                   "vendor_fp32_gemm_same_shape_tflops": None if vendor_us is None else
                   round(2.0 * B * N * 1280 * 192 / (vendor_us * 1e-6) / 1e12, 1),
                   "vendor_fp32_gemm_timed_as": "torch.mm(a, w^T) at (B N, 192) x (192, 1280), hipGraph replay, this run"},
               "entry_points": {n: {k2: round(v, 2) for k2, v in d.items()} for n, d in sorted(per_kernel.items())}}
        if not dgcnn:
            out["config"]["step"] = ("fwd + cross-entropy + generalised Dice + bwd + Adam" if args.workload in ("c3", "c3f", "c3b")
                                     else "fwd + Chamfer(reconstruction, input) + bwd + Adam")
        if world == 1 and not args.no_cpu_baseline and dgcnn and args.workload != "c2s":
            # (N = 8192: one cloud per CPU step -- a step of the oracle materialises (B,N,N) and (B,2C,N,k) tensors)
            out["cpu_baseline"] = cpu_baseline(B if N <= 2048 else 1, N, k, classes, budget_s=12.0 if N <= 2048 else 20.0,
                                               max_steps=8 if N <= 2048 else 3)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
