"""Adam over ONE flat parameter buffer.

`torch.optim.Adam(model.parameters())` (the reference's optimizer, model_trainer.py:57) walks ~450 small tensors of the
PointTransformer in 16 chunked multi-tensor launches (470 us per step on MI355X for 7.8 M parameters, launch-bound); the
same arithmetic over one contiguous buffer is a single streaming kernel.  `FlatAdam` re-points every parameter at a view of
one flat fp32 buffer (the modules, their `state_dict` and autograd are unaffected), gathers the gradients with one
`torch.cat` per step and updates the flat buffer with one launch of `fsg_adam_flat_f32` (csrc/adam.hip: float4 streaming,
one workgroup per 1024 elements, step count on the device -- torch's fused kernel, also one launch here, runs 64 Ki
elements per workgroup = 28 workgroups for DGCNN-seg).  Same update rule as torch/optim/adam.py; always capturable into
a hipGraph.  On a CPU model (host-logic tests) the flat buffer is handed to torch.optim.Adam instead.

Drop-in contract with `torch.optim.Adam(model.parameters())` as `model_trainer.py:57,189-195` uses it:
* `param_groups[0]["params"]` are the model's own parameters, so `GradScaler.step(opt)` unscales and inf-checks the real
  `p.grad` tensors and skips the step on overflow like it does for torch's Adam (the flat gradient is gathered afterwards,
  inside `step()`);
* a parameter whose `grad` is None at `step()` is left untouched (value and moments), as torch does; the step counter is
  global, so its bias correction follows the optimizer's step count rather than a per-parameter one;
* `state_dict()` / `load_state_dict()` speak torch.optim.Adam's per-parameter layout (state[i] = step, exp_avg,
  exp_avg_sq shaped like parameter i), so checkpoints move between the two optimizers in both directions.
"""
import ctypes

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
        self._views, self._spans = [], []
        off = 0
        for n in sizes:
            self._views.append(self.flat.grad[off:off + n])
            self._spans.append((off, off + n))
            off += n
        self.inner = None
        if dev.type == "cuda":
            # lr may be a float or a 1-element device tensor (then it is read on the device at every replay)
            self._groups = [dict(params=self.params, lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)]
            self.exp_avg, self.exp_avg_sq = torch.zeros_like(flat), torch.zeros_like(flat)
            self._state = torch.zeros(2, dtype=torch.float32, device=dev)   # { step, ticket } of fsg_adam_flat_f32
        else:
            self.inner = torch.optim.Adam([self.flat], lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)

    @property
    def param_groups(self):
        return self.inner.param_groups if self.inner is not None else self._groups

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def gather_grads(self):
        """concatenate the parameters' gradients into the flat gradient buffer (`self.flat.grad`); under data parallelism
        that buffer is what gets all-reduced -- one collective, no unflatten"""
        if self.flat.is_cuda:      # (a no-op after a finished backward pass: the engine's end-of-pass callback has run)
            from . import functional as F_hip
            F_hip.flush_deferred_reduces()
        grads = [p.grad.reshape(-1) if p.grad is not None else torch.zeros_like(v) for p, v in zip(self.params, self._views)]
        torch.cat(grads, out=self.flat.grad)
        return self.flat.grad

    def step_flat(self):
        """Adam update from the flat gradient buffer as it stands"""
        if self.inner is not None:
            self.inner.step()
            return
        from . import _lib
        g = self._groups[0]
        lr = g["lr"]
        lr_dev = None
        if isinstance(lr, torch.Tensor):
            if lr.device != self.flat.device or lr.dtype != torch.float32 or lr.numel() != 1:
                raise ValueError("a tensor learning rate must be one fp32 element on the parameters' device")
            lr_dev, lr = ctypes.c_void_p(lr.data_ptr()), 0.0
        P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        _lib.call("fsg_adam_flat_f32", P(self.flat.data), P(self.flat.grad), P(self.exp_avg), P(self.exp_avg_sq),
                  P(self._state), self.flat.numel(), float(lr), lr_dev, float(g["betas"][0]), float(g["betas"][1]),
                  float(g["eps"]), float(g["weight_decay"]), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))

    def step(self):
        missing = [i for i, p in enumerate(self.params) if p.grad is None]
        self.gather_grads()
        if not missing or self.inner is not None:
            self.step_flat()
            return
        # parameters without a gradient keep their value and their moments (torch.optim.Adam skips them)
        keep = [(a, b, self.flat.data[a:b].clone(), self.exp_avg[a:b].clone(), self.exp_avg_sq[a:b].clone())
                for a, b in (self._spans[i] for i in missing)]
        self.step_flat()
        with torch.no_grad():
            for a, b, w, m, v in keep:
                self.flat.data[a:b].copy_(w)
                self.exp_avg[a:b].copy_(m)
                self.exp_avg_sq[a:b].copy_(v)

    def state_dict(self):
        """torch.optim.Adam's layout over the model's parameters (loadable by `torch.optim.Adam(model.parameters())`)"""
        if self.inner is not None:
            return self.inner.state_dict()
        group = {k: v for k, v in self._groups[0].items() if k != "params"}
        group.update(amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                     decoupled_weight_decay=False)
        if isinstance(group["lr"], torch.Tensor):
            group["lr"] = float(group["lr"])
        group["params"] = list(range(len(self.params)))
        step = self._state[0].detach().cpu().clone()
        state = {}
        if float(step) > 0:
            for i, (p, (a, b)) in enumerate(zip(self.params, self._spans)):
                state[i] = {"step": step.clone(), "exp_avg": self.exp_avg[a:b].view(p.shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[a:b].view(p.shape).clone()}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """accepts torch.optim.Adam's per-parameter layout (and this class's earlier single-flat-parameter layout)"""
        if self.inner is not None:
            self.inner.load_state_dict(sd)
            return
        st = sd["state"]
        with torch.no_grad():
            self._state.zero_(), self.exp_avg.zero_(), self.exp_avg_sq.zero_()
            if len(st) == 1 and 0 in st and st[0]["exp_avg"].numel() == self.flat.numel() and len(self.params) > 1:
                self._state[0] = float(st[0]["step"])
                self.exp_avg.copy_(st[0]["exp_avg"].reshape(-1))
                self.exp_avg_sq.copy_(st[0]["exp_avg_sq"].reshape(-1))
            elif st:
                steps = {float(v["step"]) for v in st.values()}
                if len(steps) != 1:
                    raise ValueError("FlatAdam keeps one step counter; the checkpoint has per-parameter steps " + str(steps))
                self._state[0] = steps.pop()
                for i, v in st.items():
                    a, b = self._spans[int(i)]
                    if v["exp_avg"].numel() != b - a:
                        raise ValueError(f"optimizer state {i} does not match parameter {i}")
                    self.exp_avg[a:b].copy_(v["exp_avg"].reshape(-1))
                    self.exp_avg_sq[a:b].copy_(v["exp_avg_sq"].reshape(-1))
        for k, v in sd["param_groups"][0].items():
            if k != "params" and k in self._groups[0]:
                self._groups[0][k] = tuple(v) if k == "betas" else v
