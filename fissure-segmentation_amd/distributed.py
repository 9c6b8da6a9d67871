"""Data-parallel harness: one process per GPU, batch sharded over ranks, gradients averaged with
RCCL all-reduces over xGMI (torch.distributed backend "nccl" == RCCL on ROCm; "gloo" for CPU tests).

The reference is single-process; this is the build's addition for BASELINE config 4.
Every cloud is independent (kNN, FPS and attention never cross clouds), BatchNorm statistics stay
per replica like stock DDP, so the only exchange is the gradient average.  The payload is small
(DGCNNSeg: 631 428 fp32 = 2.5 MB) and latency-bound.

Default scheme (bench.py, hipGraph replay): graph 1 = forward + loss + backward + `FlatAdam.gather_grads()`
(one cat into the optimizer's flat gradient buffer, allocated before any capture) -> ONE in-place
`dist.all_reduce(opt.flat.grad)` on the replay stream (synchronous call, stream-ordered on the GPU) ->
graph 2 = scale by 1/world + fused Adam over the flat parameter buffer.  Nothing is copied back.
On xGMI's fully connected topology RCCL moves the bucket over all 7 links at once; more buckets only add
launch latency.

`BucketedGradAverager` is the eager-mode alternative (`bench.py --grad-sync bucketed --eager`): TWO flat
buckets, the point-wise head (whose gradients are complete first during backward) goes out on a side stream
while the EdgeConv backward is still running, the EdgeConv gradients follow at the end of backward.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).  Returns (rank, world, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("FSG_SHARE_GPU0"):  # rehearsal on a one-GPU box: every rank on device 0 (use with gloo)
        local = 0
    backend = backend or os.environ.get("FSG_DIST_BACKEND")
    use_gpu = torch.cuda.is_available()
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kwargs = {"device_id": device} if (use_gpu and (backend or "nccl") == "nccl") else {}
        dist.init_process_group(backend or ("nccl" if use_gpu else "gloo"), rank=rank, world_size=world, **kwargs)
    return rank, world, device


def shard_batch(global_batch, rank, world):
    """Clouds [lo, hi) of a global batch owned by `rank` (contiguous, remainder to the low ranks)."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


@torch.no_grad()
def broadcast_parameters(model, src=0):
    """Identical initial weights and BN buffers on every rank (one flat broadcast)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    tensors = [p.data for p in model.parameters()] + [b.data for b in model.buffers() if b.dtype.is_floating_point]
    flat = torch.cat([t.reshape(-1).float() for t in tensors])
    dist.broadcast(flat, src)
    off = 0
    for t in tensors:
        t.copy_(flat[off:off + t.numel()].view_as(t))
        off += t.numel()


def flat_sync_step(run_fwd_bwd, flat_grad, run_update, collective=None, probe=None):
    """One data-parallel step of the flat scheme, in the only order that is correct: `run_fwd_bwd()` (graph 1: forward, loss,
    backward, gather into `flat_grad`) -> ONE in-place SUM all-reduce of `flat_grad`, issued synchronously (async_op=False) on
    the current stream, so the device executes it after everything `run_fwd_bwd` enqueued and before anything `run_update`
    enqueues -> `run_update()` (graph 2: scale by 1/world + Adam).  No handle escapes, nothing runs on a side stream: there is
    no window in which the update could read an un-reduced buffer.  `collective` replaces dist.all_reduce in tests.
    `probe` (a StepBreakdown) marks the four boundaries of this step: fwd_bwd | allreduce | update."""
    if probe is not None:
        probe.mark()
    out = run_fwd_bwd()
    if probe is not None:
        probe.mark()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        work = (collective or dist.all_reduce)(flat_grad, op=dist.ReduceOp.SUM, async_op=False)
        if work is not None:          # a backend that hands a handle back even for a synchronous call
            work.wait()
    if probe is not None:
        probe.mark()
    run_update()
    if probe is not None:
        probe.mark()
    return out


class StepBreakdown:
    """Where a data-parallel step spends its time: fwd_bwd (graph 1) | allreduce | update (graph 2).  On a GPU the marks are HIP
    events recorded on the current (replay) stream -- stream-ordered, the host does not wait; on the CPU (gloo tests) they are
    host clock readings.  bench.py passes one to `flat_sync_step` every 10th step and reports the mean spans, maximum over the
    ranks, as config.step_breakdown_us: a first hardware run at N > 1 then shows whether a scaling loss is collective latency,
    rank skew (a long allreduce span on the fast ranks) or host launch."""
    NAMES = ("fwd_bwd", "allreduce", "update")

    def __init__(self, on_gpu):
        self.on_gpu = bool(on_gpu)
        self.steps = []          # per probed step: four marks
        self._cur = []

    def mark(self):
        if self.on_gpu:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
        else:
            import time
            e = time.perf_counter()
        self._cur.append(e)
        if len(self._cur) == 4:
            self.steps.append(self._cur)
            self._cur = []

    def spans_us(self):
        """(n_probed, 3) microseconds; call after a device synchronize"""
        out = []
        for m in self.steps:
            if self.on_gpu:
                out.append([1e3 * m[i].elapsed_time(m[i + 1]) for i in range(3)])
            else:
                out.append([1e6 * (m[i + 1] - m[i]) for i in range(3)])
        return out

    def mean_us(self):
        sp = self.spans_us()
        if not sp:
            return None
        return {n: sum(r[i] for r in sp) / len(sp) for i, n in enumerate(self.NAMES)}


class BucketedGradAverager:
    """Averages gradients across ranks in (at most) two flat buckets.  `early` = predicate on parameter names
    selecting the bucket that is reduced as soon as all of its gradients exist (on a side stream, overlapping the
    rest of backward); everything else is reduced in `finish()`, which also joins the side stream and writes the
    averages back into the .grad tensors (one foreach copy per bucket).  Between steps gradients are dropped
    (`zero_grad` sets them to None), so autograd assigns instead of accumulating -- no per-parameter add/fill
    kernels.  With world_size 1 nothing is launched at all."""

    def __init__(self, model, early=lambda name: False):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        self.params = [p for _, p in named]
        groups = [[p for n, p in named if early(n)], [p for n, p in named if not early(n)]]
        self.buckets = [{"params": g, "pending": len(g), "flat": None, "work": None} for g in groups if g]
        self.early_bucket = self.buckets[0] if (len(self.buckets) == 2 and self.world > 1) else None
        on_gpu = self.params[0].is_cuda
        self.side = torch.cuda.Stream() if (self.early_bucket is not None and on_gpu) else None
        if self.early_bucket is not None:
            for p in self.early_bucket["params"]:
                p.register_post_accumulate_grad_hook(self._on_grad)

    def zero_grad(self):
        for p in self.params:
            p.grad = None
        for b in self.buckets:
            b["pending"], b["flat"], b["work"] = len(b["params"]), None, None

    def _launch(self, b, side):
        if self.params[0].is_cuda:      # weight gradients whose split sums were left for the end of the backward pass
            from . import functional as F_hip
            F_hip.flush_deferred_reduces()
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in b["params"]]
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                b["flat"] = torch.cat([g.reshape(-1) for g in grads])
                b["work"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, async_op=True)
        else:
            b["flat"] = torch.cat([g.reshape(-1) for g in grads])
            b["work"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, async_op=True)

    def _on_grad(self, p):
        b = self.early_bucket
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch(b, self.side)

    def finish(self):
        """Call after backward(): reduces what is left, waits, writes the averaged gradients back."""
        if self.world == 1:
            return
        for b in self.buckets:
            if b["work"] is None:
                self._launch(b, None)
        for b in self.buckets:
            b["work"].wait()
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
        for b in self.buckets:
            flat = b["flat"].div_(self.world)
            outs, off = [], 0
            for p in b["params"]:
                if p.grad is None:
                    p.grad = torch.empty_like(p)
                outs.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            torch._foreach_copy_([p.grad for p in b["params"]], outs)
        # ready for the next step even when zero_grad() is not called in between: under hipGraph replay the Python of the
        # captured region (zero_grad included) does not run again, only finish() does
        for b in self.buckets:
            b["pending"], b["flat"], b["work"] = len(b["params"]), None, None
