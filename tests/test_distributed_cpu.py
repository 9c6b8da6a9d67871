"""CPU tests of the data-parallel harness with the gloo backend, world_size 2 (the N>1 path of bench.py
minus the GPU): the averaged gradient equals the mean of the per-shard gradients, weights are identical
after broadcast, batch shards tile the global batch."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _net():
    return nn.Sequential(nn.Conv1d(3, 8, 1), nn.BatchNorm1d(8), nn.LeakyReLU(0.2), nn.Conv1d(8, 4, 1))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from fissure_segmentation_amd import distributed as D
    r, w, dev = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and dev.type == "cpu"
    torch.manual_seed(100 + rank)           # different init per rank on purpose
    model = _net()
    D.broadcast_parameters(model)
    avg = D.BucketedGradAverager(model, early=lambda n: n.startswith("3."))
    torch.manual_seed(7)
    x = torch.randn(6, 3, 32)
    lo, hi = D.shard_batch(6, rank, world)
    for _ in range(2):                       # second iteration checks zero_grad / hook re-arming
        avg.zero_grad()
        model(x[lo:hi]).square().mean().backward()
        avg.finish()
    out[rank] = {"w": [p.detach().clone() for p in model.parameters()],
                 "g": [p.grad.detach().clone() for p in model.parameters()], "shard": (lo, hi)}
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_gradient_average():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert a["shard"] == (0, 3) and b["shard"] == (3, 6)
    for wa, wb in zip(a["w"], b["w"]):
        assert torch.equal(wa, wb)                              # broadcast made the replicas identical
    for ga, gb in zip(a["g"], b["g"]):
        assert torch.equal(ga, gb)                              # all-reduce: same averaged gradient
    # expected: mean of the two per-shard gradients computed serially with rank 0's weights
    torch.manual_seed(100)
    ref = _net()
    torch.manual_seed(7)
    x = torch.randn(6, 3, 32)
    grads = []
    for lo, hi in ((0, 3), (3, 6)):
        ref.zero_grad()
        ref(x[lo:hi]).square().mean().backward()
        grads.append([p.grad.clone() for p in ref.parameters()])
    for ga, g0, g1 in zip(a["g"], *grads):
        torch.testing.assert_close(ga, (g0 + g1) / 2, rtol=1e-6, atol=1e-7)


def test_shard_batch_tiles():
    from fissure_segmentation_amd import distributed as D
    for gb in (1, 7, 8, 32):
        for world in (1, 2, 3, 8):
            spans = [D.shard_batch(gb, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


def _worker_replay(rank, world, port, out):
    """the hipGraph-replay pattern of bench.py at N > 1: gradients are rewritten IN PLACE by the replayed graph and only
    finish() runs between replays -- zero_grad() belongs to the captured region and does not run again"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from fissure_segmentation_amd import distributed as D
    D.init_from_env(backend="gloo")
    torch.manual_seed(0)
    model = _net()
    avg = D.BucketedGradAverager(model)
    avg.zero_grad()                                   # capture time
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    res = []
    for step in range(3):                             # "replays"
        for i, p in enumerate(model.parameters()):
            p.grad.fill_(float((rank + 1) * (step + 1) + i))      # what the replayed backward would leave behind
        avg.finish()
        res.append([float(p.grad.flatten()[0]) for p in model.parameters()])
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_finish_without_zero_grad_between_steps():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_replay, args=(world, port, out), nprocs=world, join=True)
    assert out[0] == out[1]
    nparam = len(out[0][0])
    for step in range(3):
        for i in range(nparam):                       # mean over ranks of (rank+1)(step+1) + i
            assert abs(out[0][step][i] - (1.5 * (step + 1) + i)) < 1e-6, (step, i, out[0][step][i])


def _worker_flat(rank, world, port, out):
    """the flat scheme of bench.py at N > 1 (`--grad-sync flat`): backward -> FlatAdam.gather_grads() -> ONE in-place
    all-reduce of the flat gradient buffer -> scale by 1/world -> FlatAdam.step_flat(); the buffer is the optimizer's own,
    allocated once, never copied back into the .grad tensors"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from fissure_segmentation_amd import distributed as D
    from fissure_segmentation_amd.optim import FlatAdam
    D.init_from_env(backend="gloo")
    torch.manual_seed(100 + rank)
    model = _net()
    D.broadcast_parameters(model)
    opt = FlatAdam(model.parameters(), lr=1e-2)
    ptr = opt.flat.grad.data_ptr()
    torch.manual_seed(7)
    x = torch.randn(6, 3, 32)
    lo, hi = D.shard_batch(6, rank, world)
    grads = []
    for _ in range(3):
        opt.zero_grad()
        model(x[lo:hi]).square().mean().backward()
        opt.gather_grads()
        dist.all_reduce(opt.flat.grad, op=dist.ReduceOp.SUM)
        opt.flat.grad.mul_(1.0 / world)
        grads.append(opt.flat.grad.clone())
        opt.step_flat()
        assert opt.flat.grad.data_ptr() == ptr
    out[rank] = {"w": [p.detach().clone() for p in model.parameters()], "g": grads}
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_flat_gradient_allreduce_in_place():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_flat, args=(world, port, out), nprocs=world, join=True)
    for wa, wb in zip(out[0]["w"], out[1]["w"]):
        assert torch.equal(wa, wb)                    # replicas stay identical through three optimizer steps
    # serial reference: mean of the two shard gradients, torch.optim.Adam over the separate tensors
    torch.manual_seed(100)
    ref = _net()
    opt = torch.optim.Adam(ref.parameters(), lr=1e-2)
    twin = _net()
    torch.manual_seed(7)
    x = torch.randn(6, 3, 32)
    for step in range(3):
        shard_grads = []
        for lo, hi in ((0, 3), (3, 6)):
            twin.load_state_dict(ref.state_dict())    # BatchNorm running stats are per replica; the gradients do not read them
            twin.zero_grad()
            twin(x[lo:hi]).square().mean().backward()
            shard_grads.append(torch.cat([p.grad.reshape(-1) for p in twin.parameters()]))
        mean = (shard_grads[0] + shard_grads[1]) / 2
        torch.testing.assert_close(out[0]["g"][step], mean, rtol=1e-5, atol=1e-7)
        off = 0
        for p in ref.parameters():
            p.grad = mean[off:off + p.numel()].view_as(p).clone()
            off += p.numel()
        opt.step()
    for wa, p in zip(out[0]["w"], ref.parameters()):
        torch.testing.assert_close(wa, p.detach(), rtol=1e-4, atol=1e-6)


def _worker_order(rank, world, port, out):
    """flat_sync_step with recording stand-ins for the two graphs and an event-ordered fake collective: the collective must
    see graph 1 finished and graph 2 not started, must be called synchronously on the optimizer's own buffer, and graph 2
    must read the REDUCED values"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from fissure_segmentation_amd import distributed as D
    D.init_from_env(backend="gloo")
    flat = torch.zeros(5)
    log, seen = [], []

    def g1():
        log.append("g1")
        flat.fill_(float(rank + 1))
        return "loss"

    def fake_all_reduce(t, op=None, async_op=None):
        assert log == ["g1"], log                      # after graph 1, before graph 2
        assert async_op is False and op == dist.ReduceOp.SUM
        assert t.data_ptr() == flat.data_ptr()         # in place, the optimizer's own buffer
        log.append("allreduce")
        return dist.all_reduce(t, op=op, async_op=False)

    def g2():
        assert log == ["g1", "allreduce"], log
        log.append("g2")
        seen.append(flat.clone())

    for _ in range(2):
        del log[:]
        assert D.flat_sync_step(g1, flat, g2, collective=fake_all_reduce) == "loss"
        assert log == ["g1", "allreduce", "g2"]
    out[rank] = [t.tolist() for t in seen]
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_flat_sync_step_orders_collective_between_the_graphs():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_order, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        assert out[r] == [[3.0] * 5, [3.0] * 5]        # 1 + 2 from both ranks: the update read the reduced buffer


def _worker_breakdown(rank, world, port, out):
    """flat_sync_step with a StepBreakdown probe on every 2nd step: the three spans (fwd_bwd | allreduce | update) of a probed
    step must add up to the time of the whole step measured around the call, and the collective must still see the right
    buffer.  Stand-ins for the graphs sleep for known times (rank 1 is the slow rank: its peer waits inside the collective)."""
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from fissure_segmentation_amd import distributed as D
    D.init_from_env(backend="gloo")
    flat = torch.zeros(1 << 16)
    probe = D.StepBreakdown(on_gpu=False)

    def g1():
        time.sleep(0.020 + 0.015 * rank)
        flat.fill_(float(rank + 1))

    def g2():
        time.sleep(0.010)

    whole = []
    for i in range(6):
        dist.barrier()
        t0 = time.perf_counter()
        D.flat_sync_step(g1, flat, g2, probe=probe if i % 2 == 0 else None)
        whole.append(1e6 * (time.perf_counter() - t0))
        assert float(flat[0]) == 3.0
    out[rank] = {"spans": probe.spans_us(), "whole": whole[0::2], "mean": probe.mean_us()}
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_step_breakdown_adds_up():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_breakdown, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        rec = out[r]
        assert len(rec["spans"]) == 3 and set(rec["mean"]) == {"fwd_bwd", "allreduce", "update"}
        for spans, whole in zip(rec["spans"], rec["whole"]):
            assert abs(sum(spans) - whole) <= 0.05 * whole, (spans, whole)     # the three spans ARE the step
            assert spans[0] >= 0.9 * 1e6 * (0.020 + 0.015 * r) and spans[2] >= 0.9 * 1e4
    # rank skew shows up where it belongs: the fast rank waits for the slow one INSIDE its allreduce span
    assert out[0]["mean"]["allreduce"] > out[1]["mean"]["allreduce"] + 5e3
