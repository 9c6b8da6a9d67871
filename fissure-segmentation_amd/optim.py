"""Adam over ONE flat parameter buffer.

`torch.optim.Adam(model.parameters())` (the reference's optimizer, model_trainer.py:57) walks ~450 small tensors of the
PointTransformer in 16 chunked multi-tensor launches (470 us per step on MI355X for 7.8 M parameters, launch-bound); the
same arithmetic over one contiguous buffer is a single streaming kernel.  `FlatAdam` re-points every parameter at a view of
one flat fp32 buffer (the modules, their `state_dict` and autograd are unaffected), gathers the gradients with one
`torch.cat` per step and lets torch's own fused Adam kernel update the flat buffer: bit-identical to per-tensor Adam
(element-wise update, same kernel), capturable into a hipGraph.
"""
import torch


class FlatAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=False):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        if any(p.device != dev or p.dtype != torch.float32 for p in self.params):
            raise ValueError("FlatAdam needs fp32 parameters on one device")
        sizes = [p.numel() for p in self.params]
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p, n in zip(self.params, sizes):
                flat[off:off + n].copy_(p.reshape(-1))
                p.data = flat[off:off + n].view(p.shape)      # the module parameter now lives inside the flat buffer
                off += n
        self.flat = torch.nn.Parameter(flat)
        self.flat.grad = torch.zeros_like(flat)
        self._views = []
        off = 0
        for n in sizes:
            self._views.append(self.flat.grad[off:off + n])
            off += n
        self.inner = torch.optim.Adam([self.flat], lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      capturable=capturable, fused=dev.type == "cuda")

    @property
    def param_groups(self):
        return self.inner.param_groups

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def gather_grads(self):
        """concatenate the parameters' gradients into the flat gradient buffer (`self.flat.grad`); under data parallelism
        that buffer is what gets all-reduced -- one collective, no unflatten"""
        grads = [p.grad.reshape(-1) if p.grad is not None else torch.zeros_like(v) for p, v in zip(self.params, self._views)]
        torch.cat(grads, out=self.flat.grad)
        return self.flat.grad

    def step_flat(self):
        """Adam update from the flat gradient buffer as it stands"""
        self.inner.step()

    def step(self):
        self.gather_grads()
        self.step_flat()

    def state_dict(self):
        return self.inner.state_dict()

    def load_state_dict(self, sd):
        self.inner.load_state_dict(sd)
