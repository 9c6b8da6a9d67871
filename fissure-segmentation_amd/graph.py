"""hipGraph replay of a training step with static shapes.

The reference's loop (`ModelTrainer.forward_step`, model_trainer.py:154-213: `model(x)` -> `loss_fn(out, y)` -> `backward`
-> optimizer step -> `zero_grad`) launches ~150 (DGCNN) to ~1300 (PointTransformer) kernels per step from Python; on an
MI355X the GPU then waits for the interpreter.  All shapes of a step are static (fixed batch, fixed N), every HIP stage
of this package takes an explicit stream and owns no hidden state, and no host synchronisation happens inside the models,
so the whole step can be captured once and replayed:

    step = GraphedTrainStep(model, criterion, optimizer, x, y)     # captures; x, y become the static input buffers
    for xb, yb in loader:
        loss = step(xb, yb)                                         # copies into the static buffers, replays

`criterion(out, y)` may return a tensor or `(tensor, dict)` like the reference's losses.  The optimizer must be
capturable (`torch.optim.Adam(..., capturable=True)` or `optim.FlatAdam(..., capturable=True)`).  Falls back to eager
launches when capture fails (`step.captured` tells which).
"""
import torch


class GraphedTrainStep:
    def __init__(self, model, criterion, optimizer, x, y, warmup=3):
        if not x.is_cuda:
            raise RuntimeError("GraphedTrainStep needs GPU tensors")
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.x, self.y = x.clone(), y.clone()
        self.captured = False
        self._graph = None
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side):       # optimizer state, kernel attributes, library heuristics: all before the capture
            for _ in range(max(1, warmup)):
                self._eager()
        torch.cuda.current_stream(x.device).wait_stream(side)
        torch.cuda.synchronize(x.device)
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._loss = self._eager()
            self._graph, self.captured = graph, True
        except Exception as e:   # capture is an optimisation, never a requirement
            import warnings
            warnings.warn(f"hipGraph capture failed ({type(e).__name__}: {e}); the step runs eagerly")
            torch.cuda.synchronize(x.device)

    def _eager(self):
        self.optimizer.zero_grad(set_to_none=True)
        out = self.criterion(self.model(self.x), self.y)
        loss = out[0] if isinstance(out, tuple) else out
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def __call__(self, x=None, y=None):
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        if y is not None:
            self.y.copy_(y, non_blocking=True)
        if self._graph is None:
            return self._eager()
        self._graph.replay()
        return self._loss
