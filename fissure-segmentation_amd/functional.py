"""Tensor-level entry points over the C ABI: argument checking, output allocation, stream/device
plumbing and the autograd glue.  Every function requires CUDA (HIP) tensors -- no CPU path."""
import contextlib as _contextlib
import ctypes
import os as _os

import torch

from . import _lib


def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("fissure_segmentation_amd runs on the GPU only (got a CPU tensor); "
                               "the CPU restatement lives in oracle/ and is test infrastructure")


def _p(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


# ------------------------------------------------------------------ mixed precision (model_trainer.py:75-76,157,189-193)
# The reference runs every segmentation step under `autocast` (fp16) + GradScaler.  The HIP kernels take fp32 rows and
# the graph is always built in fp32 (indices equal the fp32 CPU path), so every autograd.Function leaves the autocast
# region: floating-point inputs are cast to fp32 on entry, forward and backward run with autocast off, outputs and
# gradients are fp32.  Reduced precision is an explicit choice here (`set_mfma_operands("bf16")`), not an ambient one.
def _amp_fwd(fn):
    return torch.amp.custom_fwd(fn, device_type="cuda", cast_inputs=torch.float32)


def _amp_bwd(fn):
    return torch.amp.custom_bwd(fn, device_type="cuda")


def no_autocast():
    """context: autocast off (entered only when it is on, so CPU-only hosts see no device warning)"""
    if torch.is_autocast_enabled("cuda"):
        return torch.autocast("cuda", enabled=False)
    return _contextlib.nullcontext()


# bf16 OPERAND MODE (BASELINE configs 3-5 name bf16 as their compute type).  What it changes: the operands of the dense
# contractions -- the per-edge 64 x C2 product of the two-layer EdgeConv (fsg_edgeconv2_{fwd,bwd}_bf16 on
# v_mfma_f32_32x32x16_bf16) and the point-wise / Linear GEMMs that go to the vendor library -- are rounded to bf16,
# accumulation stays fp32.  What it never changes: the graph build (always fp32: indices equal the fp32 CPU path), BatchNorm
# statistics, selection, every stored activation and every gradient tensor (fp32).  Switched on explicitly
# (`set_mfma_operands("bf16")` / `with mfma_operands("bf16")`) or by an ambient `torch.autocast("cuda", torch.bfloat16)`
# around a model forward; the backward of every op follows the mode its forward ran in.
_mfma_mode = "f32"


def set_mfma_operands(kind):
    global _mfma_mode
    if kind not in ("f32", "bf16"):
        raise ValueError("MFMA operand type must be 'f32' or 'bf16'")
    _mfma_mode = kind


@_contextlib.contextmanager
def mfma_operands(kind):
    global _mfma_mode
    old = _mfma_mode
    set_mfma_operands(kind)
    try:
        yield
    finally:
        _mfma_mode = old


def bf16_operands():
    return _mfma_mode == "bf16"


# The vendor GEMMs (point-wise head, Linear layers) can run with bf16 operands too, but every call then pays a cast pass
# over its fp32 activations and gradients (all stored tensors stay fp32): measured on MI355X at the config-4 shape the
# step got SLOWER (4.64 vs 4.17 ms), so this half of the mode is a separate opt-in (FSG_BF16_VENDOR_GEMM=1 /
# `set_bf16_vendor_gemm(True)`); the hand-written kernels convert their operands on the way out of LDS at no extra traffic.
_bf16_vendor = _os.environ.get("FSG_BF16_VENDOR_GEMM", "0") == "1"


def set_bf16_vendor_gemm(flag):
    global _bf16_vendor
    _bf16_vendor = bool(flag)


def _bf16_gemm():
    return _mfma_mode == "bf16" and _bf16_vendor


def _ambient_bf16():
    return torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16


_MM_OUT_DTYPE = True


def _bf16_mm(a, b):
    """a (M,K) @ b (K,N) with bf16 operands and fp32 accumulation -> fp32 (vendor GEMM, hipBLASLt)"""
    global _MM_OUT_DTYPE
    a16, b16 = a.to(torch.bfloat16), b.to(torch.bfloat16)
    if _MM_OUT_DTYPE:
        try:
            return torch.mm(a16, b16, out_dtype=torch.float32)
        except (TypeError, RuntimeError):
            _MM_OUT_DTYPE = False
    return torch.mm(a16, b16).float()


_deterministic = bool(_os.environ.get("FSG_DETERMINISTIC"))


def set_deterministic(flag=True):
    """Reproducible (ordered) reductions where the default scatters with fp32 atomics: today the Chamfer backward, whose
    ordered form walks the reverse graph of the arg-min and is slow when many points share one nearest neighbour (a
    collapsed reconstruction early in training: 5.3 -> 6.3 ms per PC-AE step).  Also on under
    `torch.use_deterministic_algorithms(True)` and FSG_DETERMINISTIC=1.  The DGCNN path is reproducible regardless."""
    global _deterministic
    _deterministic = bool(flag)


def deterministic():
    return _deterministic or torch.are_deterministic_algorithms_enabled()


# ------------------------------------------------------------------ dense kNN (utils/general_utils.py:315)
_KNN_EXPERIMENT_FLAGS = 8   # the first MFMA design (libfsg_hip_experiments.so): independent cross-check for tests / tools


def knn_graph(x, k, c_knn=None, fix_diag=True, drop_first=False, return_dist=False, force_rows_kernel=False,
              _debug_flags=0, out=None, prepared=None, pq_weight=None):
    """x: (B,C,N) -> idx (B,N,k) int32 [, dist (B,N,k) fp32].  Channel slices are passed by stride.  `out`: a contiguous
    (B,N,k) int32 tensor to write the graph into (e.g. a slice of one buffer holding all graphs of a step, so that their
    reverse graphs can be built in one launch: build_reverse_graphs).  `prepared` = (workspace, x_pm): the producer of x has
    already emitted the build's prep products into `workspace` (edgeconv1 / edgeconv2 with knn_ws=) and x_pm is its point-major
    copy (B,N,C): the build starts at its main kernel (include/fsg_hip.h: fsg_knn_dense_prepared_f32).
    `pq_weight` (rows, C) with C = c_knn <= 4 (the [W_rel ; W_ctr - W_rel] weight of the FIRST EdgeConv's first conv): the build
    also emits that block's per-point rows pq (B, N, rows) = x^T pq_weight^T (fsg_knn_dense_ws_pq_f32) -> returns (idx, pq); pq is
    None when the shape does not qualify (then the caller runs the product itself)."""
    _need_gpu(x)
    if x.dim() != 3:
        raise ValueError(f"expected (B,C,N), got {tuple(x.shape)}")
    x = x.detach()
    if x.dtype != torch.float32:
        x = x.float()  # the graph is always built in fp32 (precision)
    if x.stride(2) != 1:
        x = x.contiguous()
    B, C, N = x.shape
    c_knn = C if c_knn is None else c_knn
    if out is not None:
        if out.shape != (B, N, k) or out.dtype != torch.int32 or not out.is_contiguous() or out.device != x.device:
            raise ValueError("knn_graph: `out` must be a contiguous (B,N,k) int32 tensor on the input's device")
        idx = out
    else:
        idx = torch.empty(B, N, k, dtype=torch.int32, device=x.device)
    dist = torch.empty(B, N, k, dtype=torch.float32, device=x.device) if return_dist else None
    flags = (_lib.KNN_FIX_DIAG if fix_diag else 0) | (_lib.KNN_DROP_FIRST if drop_first else 0) | \
        (_lib.KNN_FORCE_ROWS if force_rows_kernel else 0) | _debug_flags
    if prepared is not None and c_knn == C and not return_dist and not force_rows_kernel and not _debug_flags:
        ws, x_pm = prepared
        with torch.cuda.device(x.device):
            _lib.call("fsg_knn_dense_prepared_f32", _p(x_pm), B, N, C, k, flags, _p(idx), None, _p(ws), ws.numel() * ws.element_size(),
                      _stream())
        return idx
    ws_bytes = _lib.lib.fsg_knn_dense_workspace_bytes(B, N, c_knn)
    xx = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=x.device)
    if pq_weight is not None:
        rows = pq_weight.shape[0]
        if (C == c_knn and c_knn <= 4 and tuple(pq_weight.shape) == (rows, C) and 256 % rows == 0 and not return_dist and
                not (_debug_flags & _KNN_EXPERIMENT_FLAGS)):
            pq = torch.empty(B, N, rows, dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device):
                _lib.call("fsg_knn_dense_ws_pq_f32", _p(x), B, N, x.stride(0), x.stride(1), c_knn, k, flags, _p(idx), None, _p(xx),
                          ws_bytes, _p(_f32c(pq_weight.detach())), rows, _p(pq), _stream())
            return idx, pq
    with torch.cuda.device(x.device):
        if _debug_flags & _KNN_EXPERIMENT_FLAGS:   # superseded designs (tests / tools): libfsg_hip_experiments.so
            xl = _lib.experiments()
            rc = xl.fsg_knn_experiment_f32(_p(x), B, N, x.stride(0), x.stride(1), c_knn, k, flags, _p(idx), _p(dist), _p(xx),
                                           _stream())
            if rc == 0:
                return (idx, dist) if return_dist else idx
            if rc != 3:   # FSG_ERR_UNSUPPORTED: shape outside that kernel's envelope -> the production kernel below
                raise RuntimeError(xl.fsg_last_error().decode())
            flags &= ~_KNN_EXPERIMENT_FLAGS
        _lib.call("fsg_knn_dense_ws_f32", _p(x), B, N, x.stride(0), x.stride(1), c_knn, k, flags, _p(idx), _p(dist), _p(xx),
                  ws_bytes, _stream())
    if pq_weight is not None:
        return idx, None
    return (idx, dist) if return_dist else idx


# ------------------------------------------------------------------ edge features (models/dgcnn.py:31-36)
class _EdgeGather(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, x, idx):
        B, C, N = x.shape
        k = idx.shape[2]
        half = x.dtype == torch.bfloat16       # bf16 storage: fsg_edge_gather_*_bf16
        xc = x.detach().contiguous() if half else _f32c(x)
        out = torch.empty(B, 2 * C, N, k, dtype=xc.dtype, device=x.device)
        with torch.cuda.device(x.device):
            _lib.call("fsg_edge_gather_fwd_bf16" if half else "fsg_edge_gather_fwd_f32", _p(xc), _p(idx), _p(out), B, C, N, k,
                      _stream())
        ctx.save_for_backward(idx)
        ctx.shape = (B, C, N, k)
        ctx.half = half
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        B, C, N, k = ctx.shape
        half = ctx.half and g.dtype == torch.bfloat16
        g = g.detach().contiguous() if half else _f32c(g)
        gx = torch.empty(B, C, N, dtype=torch.float32, device=g.device)      # accumulated in fp32 in both modes
        with torch.cuda.device(g.device):
            _lib.call("fsg_edge_gather_bwd_bf16" if half else "fsg_edge_gather_bwd_f32", _p(g), _p(idx), _p(gx), B, C, N, k,
                      _stream())
        return (gx.to(torch.bfloat16) if ctx.half else gx), None


def knn_edge_features(x, k, knn_only_over_coords=False):
    """create_neighbor_features (models/dgcnn.py:15-36) with a dynamic graph, one C-ABI call: x (B,C,N) fp32 ->
    (edge (B,2C,N,k), idx (B,N,k) int32).  Forward only (the differentiable composition is knn_graph + edge_features)."""
    _need_gpu(x)
    xc = _f32c(x)
    B, C, N = xc.shape
    idx = torch.empty(B, N, k, dtype=torch.int32, device=x.device)
    edge = torch.empty(B, 2 * C, N, k, dtype=torch.float32, device=x.device)
    c_knn = 3 if knn_only_over_coords else C
    ws_bytes = _lib.lib.fsg_knn_dense_workspace_bytes(B, N, c_knn)
    ws = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.call("fsg_knn_gather_fused_ws_f32", _p(xc), B, C, N, k, c_knn, _p(idx), _p(edge), _p(ws), ws_bytes, _stream())
    return edge, idx


def edge_features(x, idx):
    """x (B,C,N) fp32, idx (B,N,k) int32/int64 -> (B,2C,N,k) = cat(x_j - x_i, x_i)."""
    _need_gpu(x, idx)
    if idx.dtype != torch.int32:
        idx = idx.to(torch.int32)
    return _EdgeGather.apply(x, idx.contiguous())


# ------------------------------------------------------------------ point-major linear layer (1x1 conv as one GEMM)
SMALL_GEMM_FLOPS = 6e8   # below this the vendor GEMM tends to pick one huge macro-tile (one workgroup): use fsg_gemm_small_f32


def gemm_small(a, sa_i, sa_k, b, sb_k, sb_j, bias, I, J, K, rowsum=False, defer=False, bf16=False):
    """C (I,J) = A(i,k) B(k,j) (+ bias[j]) with explicit element strides -- include/fsg_hip.h: fsg_gemm_small_f32.
    rowsum=True: also sum_k A(i,k) (fsg_gemm_small_rowsum_f32: the bias gradient next to a weight gradient) -> (C, rowsum).
    defer=True (no bias): the split reduction is left to ONE launch for all deferred products of the running backward pass
    (fsg_gemm_small_reduce_many_f32 from an end-of-backward callback of the autograd engine; at once outside a backward pass):
    the returned tensors are complete when `loss.backward()` returns -- for weight gradients, which nothing reads before.
    bf16=True: operands rounded to bf16 inside the kernel, bf16 matrix instruction, fp32 accumulation (fsg_gemm_small_bf16)."""
    out = torch.empty(I, J, dtype=torch.float32, device=a.device)
    nbytes = _lib.lib.fsg_gemm_small_workspace_bytes(I, J, K)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=a.device) if nbytes else None
    rs = torch.empty(I, dtype=torch.float32, device=a.device) if rowsum else None
    with torch.cuda.device(a.device):
        if defer and bias is None:
            splits = ctypes.c_int(0)
            if bf16:
                _lib.call("fsg_gemm_small_bf16", _p(a), sa_i, sa_k, _p(b), sb_k, sb_j, None, _p(out), J, I, J, K, _p(rs), _p(ws),
                          ctypes.byref(splits), _stream())
            else:
                _lib.call("fsg_gemm_small_deferred_f32", _p(a), sa_i, sa_k, _p(b), sb_k, sb_j, _p(out), J, I, J, K, _p(rs), _p(ws),
                          ctypes.byref(splits), _stream())
            if splits.value > 1:
                # raw addresses, not the tensors: AccumulateGrad takes a gradient over as `.grad` only when nobody else holds it
                # (it would copy -- the unreduced bytes -- otherwise); `.grad` then keeps the memory alive past the flush
                _defer_reduce((ws, out.data_ptr(), rs.data_ptr() if rs is not None else None, splits.value, I, J, a.device))
        elif bf16:
            _lib.call("fsg_gemm_small_bf16", _p(a), sa_i, sa_k, _p(b), sb_k, sb_j, _p(bias), _p(out), J, I, J, K, _p(rs), _p(ws), None,
                      _stream())
        else:
            _lib.call("fsg_gemm_small_rowsum_f32", _p(a), sa_i, sa_k, _p(b), sb_k, sb_j, _p(bias), _p(out), J, I, J, K, _p(rs),
                      _p(ws), _stream())
    return (out, rs) if rowsum else out


_pending_reduces = []     # (partials, &C, &rowsum | None, S, I, J, device) of the running backward pass
_defer_weight_grads = True


def set_deferred_weight_grads(flag):
    """Weight gradients of the small Linears that go straight to a leaf parameter are summed over their reduction splits by ONE
    launch at the end of the backward pass (default on).  Turn off when something reads parameter gradients DURING the backward
    pass (gradient hooks that copy them, e.g. torch's DistributedDataParallel buckets); this package's own
    BucketedGradAverager flushes before it reads."""
    global _defer_weight_grads
    _defer_weight_grads = bool(flag)


def _grad_targets(w):
    """the leaf parameters that the gradient of `w` ends in WITHOUT being read on the way, or None: `w` itself when it is a leaf;
    the parameters a producer names in `w._fsg_grad_unread` when its backward only hands views of the gradient on (_PackQKV)"""
    if w.is_leaf:
        return (w,)
    return getattr(w, "_fsg_grad_unread", None)


def _may_defer(targets):
    """Deferring a weight gradient's split sum to the end of the backward pass is sound when the autograd engine merely STORES the
    tensor until then: it goes to parameters whose `.grad` is empty (AccumulateGrad adopts the tensor; with a gradient already
    there it would add -- unreduced bytes), outside create_graph, and the switch is on."""
    return (_defer_weight_grads and targets is not None and not torch.is_grad_enabled() and
            all(p.grad is None for p in targets))


def flush_deferred_reduces():
    """sum the split partials of every deferred gemm_small product (see gemm_small(defer=True)); idempotent"""
    global _pending_reduces
    jobs, _pending_reduces = _pending_reduces, []
    for i0 in range(0, len(jobs), _lib.GEMM_REDUCE_MAX_JOBS):
        chunk = jobs[i0:i0 + _lib.GEMM_REDUCE_MAX_JOBS]
        tab = _lib.GemmReduceJobs()
        for i, (ws, out, rs, S, I, J, _dev) in enumerate(chunk):
            tab.part[i], tab.C[i], tab.rowsum[i] = ws.data_ptr(), out, rs
            tab.ldc[i], tab.S[i], tab.I[i], tab.J[i] = J, S, I, J
        tab.n = len(chunk)
        with torch.cuda.device(chunk[0][6]):
            _lib.call("fsg_gemm_small_reduce_many_f32", ctypes.byref(tab), _stream())


def _defer_reduce(job):
    _pending_reduces.append(job)
    # (queued per product, not once per pass: a pass that dies half-way never runs its callbacks, and a flag would then keep the
    # next pass from queueing; the first callback to run does the work, the others find the list empty)
    try:     # runs when the engine has finished this backward pass, with the caller's current stream
        torch.autograd.Variable._execution_engine.queue_callback(flush_deferred_reduces)
    except RuntimeError:     # not inside a backward pass
        flush_deferred_reduces()


def _small(I, J, K, *tensors):
    """Route the product C (I,J) = A (I,K) B (K,J) to fsg_gemm_small_f32?  Yes when both output dimensions are <= 256 --
    the vendor library then runs ONE workgroup and its time grows with I*J*K (tools/probe_vendor_gemm.py) -- and the
    product is big enough for that to cost more than the ~8 us of the small kernel, yet below SMALL_GEMM_FLOPS."""
    if not all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in tensors):
        return False
    vol = float(I) * J * K
    return 0 < I <= 256 and 0 < J <= 256 and (1 << 21) <= vol and 2.0 * vol < SMALL_GEMM_FLOPS


class _LinearPM(torch.autograd.Function):
    """y = x W^T (+ b) over point-major rows.  Large products go to the vendor GEMM (the weight gradient dW = dY^T X, a
    tiny output behind a 16k-long reduction, as a 16-way split-K batched GEMM + a sum: 36-78 us instead of 106-132 us);
    small ones -- everything in the PointTransformer path -- to fsg_gemm_small_f32, because the vendor library runs
    them as a single workgroup."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        ctx.bf16 = ctx.pw16 = False
        tw, tb = _grad_targets(w), (_grad_targets(b) if b is not None else ())
        ctx.defer_targets = tw + tb if (tw is not None and tb is not None) else None
        x2 = x.reshape(-1, x.shape[-1])
        N, K = w.shape
        M = x2.shape[0]
        ctx.gs16 = (_bf16_linear and bf16_operands() and M > 0 and
                    all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in (x2, w)))
        if ctx.gs16:     # bf16 operand mode: every product of the Linear on fsg_gemm_small_bf16 (same launches as the fp32 route)
            y = gemm_small(x2, K, 1, w, 1, K, b.contiguous() if b is not None else None, M, N, K, bf16=True)
            return y if x.dim() == 2 else y.view(*x.shape[:-1], N)     # (no view of a 2-D result: an in-place ReLU may follow)
        if not (_bf16_linear and bf16_operands()) and _small(M, N, K, x2, w):
            y = gemm_small(x2, K, 1, w, 1, K, b.contiguous() if b is not None else None, M, N, K)
            return y if x.dim() == 2 else y.view(*x.shape[:-1], N)
        ctx.pw16 = _pw_bf16_ok(x2, w)
        if ctx.pw16:     # bf16 operand mode on the hand-written kernel: one bf16 piece per operand, fp32 accumulation
            y = torch.empty(*x.shape[:-1], N, dtype=torch.float32, device=x.device)      # final shape: no view leaves the Function
            pw_linear_bf16(x2, w, b, out=y.view(-1, N))
            return y
        ctx.bf16 = _bf16_gemm() and x.is_cuda
        if ctx.bf16:
            y = _bf16_mm(x2, w.t())
            if b is not None:
                y += b
            return y.view(*x.shape[:-1], N)
        return torch.nn.functional.linear(x, w, b)

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g2 = g.reshape(-1, g.shape[-1])
        x2 = x.reshape(-1, x.shape[-1])
        gx = gw = gb = None
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        if ctx.gs16:     # dX = bf16(dY) bf16(W), dW = bf16(dY)^T bf16(X) (+ the bias gradient: row sums of the rounded dY^T)
            N, K = w.shape
            M = g2.shape[0]
            if ctx.needs_input_grad[0]:
                gx = gemm_small(g2, N, 1, w, K, 1, None, M, K, N, bf16=True).view_as(x)
            want_gb = ctx.has_bias and ctx.needs_input_grad[2]
            if ctx.needs_input_grad[1]:
                x2c = x2 if x2.is_contiguous() else x2.contiguous()
                gw = gemm_small(g2, 1, N, x2c, K, 1, None, N, K, M, rowsum=want_gb, defer=_may_defer(ctx.defer_targets), bf16=True)
                if want_gb:
                    gw, gb = gw
            elif want_gb:
                gb = _bias_grad(g2)
            return gx, gw, gb
        if ctx.pw16:     # both gradients on bf16 operands too: dX = bf16(dY) bf16(W), dW = bf16(dY)^T bf16(X)
            if ctx.needs_input_grad[0]:
                gx = (pw_linear_bf16(g2, w.t(), None) if w.shape[0] % 32 == 0 else g2 @ w).view_as(x)
            if ctx.needs_input_grad[1]:
                gw = pw_tn_bf16(g2, x2 if x2.is_contiguous() else x2.contiguous())
            if ctx.has_bias and ctx.needs_input_grad[2]:
                gb = _bias_grad(g2)
            return gx, gw, gb
        if ctx.needs_input_grad[0]:
            gx = _linear_dx(g2, w, bf16=ctx.bf16).view_as(x)
        want_gb = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            gw = _linear_dw(g2, x2, bf16=ctx.bf16, with_bias_grad=want_gb, defer=_may_defer(ctx.defer_targets))
            if isinstance(gw, tuple):      # the small-GEMM path hands the bias gradient (row sums of dY^T) over with the product
                gw, gb = gw
        if want_gb and gb is None:
            gb = _bias_grad(g2)
        return gx, gw, gb


def _linear_dx(g2, w, out=None, bf16=False):
    """dX = dY W for dY (M,N), W (N,K); `out`: accumulate into an existing (M,K) gradient instead (one GEMM with beta = 1)"""
    N, K = w.shape
    M = g2.shape[0]
    if bf16:
        r = _bf16_mm(g2, w)
        return r if out is None else out.add_(r)
    if out is not None:
        return out.addmm_(g2, w)
    return gemm_small(g2, N, 1, w, K, 1, None, M, K, N) if _small(M, K, N, g2, w) else g2 @ w


def _linear_dw(g2, x2, bf16=False, with_bias_grad=False, defer=False):
    """dW = dY^T X: a tiny output behind a long reduction (see _LinearPM).  with_bias_grad: the small-GEMM path returns
    (dW, db) -- db = column sums of dY as a by-product of the same launch; the other paths return dW alone.
    defer: the caller knows that nothing reads the result inside this backward pass (gemm_small(defer=True))"""
    M, N = g2.shape
    K = x2.shape[1]
    if bf16:
        return _bf16_mm(g2.t(), x2)
    S = 16 if (M % 16 == 0 and M >= 4096) else 1
    if _small(N, K, M, g2, x2):
        return gemm_small(g2, 1, N, x2, K, 1, None, N, K, M, rowsum=with_bias_grad, defer=defer)
    if S > 1 and x2.is_contiguous():
        return torch.bmm(g2.view(S, M // S, -1).transpose(1, 2), x2.view(S, M // S, -1)).sum(0)
    return g2.t() @ x2


def _bias_grad(g2):
    """column sums of dY (M,N).  Narrow outputs (the class logits: N = 4) go to fsg_colsum_narrow_f32 -- ATen's dim-0
    reduction of a (16384, 4) tensor takes 17 us; otherwise `sum(0)` (a one-row product with ones is 3-15x slower:
    tools/probe_bias_grad.py)"""
    M, N = g2.shape
    # (one workgroup: worth it up to ~128 K elements -- (16384, 32) takes it 26 us against ATen's 9)
    if g2.is_cuda and g2.dtype == torch.float32 and N <= 32 and (N & (N - 1)) == 0 and 4096 <= M * N <= 131072:
        out = torch.empty(N, dtype=torch.float32, device=g2.device)
        with torch.cuda.device(g2.device):
            _lib.call("fsg_colsum_narrow_f32", _p(g2), M, N, _p(out), _stream())
        return out
    return g2.sum(0)


class _LinearPMGiven(torch.autograd.Function):
    """y = x W^T whose VALUE another kernel has already computed (the previous EdgeConv's apply pass emits the next block's [P | Q]
    rows from the tile it holds in LDS: fsg_edgeconv_apply_pq_f32): the forward hands that tensor through, the backward is the
    ordinary one of the product (dX = dY W, dW = dY^T X)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, w, value):
        ctx.save_for_backward(x, w)
        return value.view_as(value)

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g2 = g.reshape(-1, g.shape[-1])
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        x2 = x.reshape(-1, x.shape[-1])
        gx = _linear_dx(g2, w).view_as(x) if ctx.needs_input_grad[0] else None
        gw = _linear_dw(g2, x2 if x2.is_contiguous() else x2.contiguous()) if ctx.needs_input_grad[1] else None
        return gx, gw, None


def linear_pm(x, w, b=None):
    # (routing the EdgeConvs' per-point P/Q products -- M = B N rows, K = 64, N = 128 -- through fsg_pw_linear_f32 /
    # fsg_pw_tn_f32 was measured SLOWER than the vendor GEMM + split-K pair at these sizes: 1.212 vs 1.183 ms per config-2 step,
    # PointTransformer 8.45 vs 8.32 ms -- each product pays its weight-image launch)
    return _LinearPM.apply(x, w, b)


class _LinearPM2(torch.autograd.Function):
    """Two bias-free point-wise layers on the SAME input rows (DGCNNSeg: the global-feature conv 192 -> 1024 and the
    `levels` half of the first head conv 192 -> 256, models/dgcnn.py:134-160): forward as two products, backward with the
    second input gradient accumulated into the first by its GEMM (beta = 1) -- autograd would otherwise add the two
    (M,K) gradients in a kernel of its own."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, w1, w2):
        ctx.save_for_backward(x, w1, w2)
        x2 = x.reshape(-1, x.shape[-1])
        ctx.bf16 = _bf16_gemm() and x.is_cuda
        if ctx.bf16:
            x16 = x2.to(torch.bfloat16)
            return _bf16_mm(x16, w1.t()), _bf16_mm(x16, w2.t())
        return torch.nn.functional.linear(x2, w1), torch.nn.functional.linear(x2, w2)

    @staticmethod
    @_amp_bwd
    def backward(ctx, g1, g2):
        x, w1, w2 = ctx.saved_tensors
        x2 = x.reshape(-1, x.shape[-1])
        gs = [g.contiguous() if g is not None else None for g in (g1, g2)]
        gx = None
        if ctx.needs_input_grad[0]:
            for g, w in zip(gs, (w1, w2)):
                if g is not None:
                    gx = _linear_dx(g, w, bf16=ctx.bf16) if gx is None else _linear_dx(g, w, out=gx, bf16=ctx.bf16)
            gx = gx.view_as(x) if gx is not None else None
        gw1 = _linear_dw(gs[0], x2, bf16=ctx.bf16) if (gs[0] is not None and ctx.needs_input_grad[1]) else None
        gw2 = _linear_dw(gs[1], x2, bf16=ctx.bf16) if (gs[1] is not None and ctx.needs_input_grad[2]) else None
        return gx, gw1, gw2


def linear_pm2(x, w1, w2):
    """(x W1^T, x W2^T) for x (M,K) contiguous fp32 rows; see _LinearPM2"""
    return _LinearPM2.apply(x, w1, w2)


class _SplitCols(torch.autograd.Function):
    """W (R, C) -> the two column blocks (W[:, :c0], W[:, c0:]) as views (the GEMMs take the row stride); the gradient is
    ONE concatenation.  Slicing a weight twice costs autograd a zero-filled (R, C) tensor + a strided copy per slice and
    an add to merge them."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, w, c0):
        ctx.c0, ctx.c1 = c0, w.shape[1] - c0
        return w[:, :c0], w[:, c0:]

    @staticmethod
    @_amp_bwd
    def backward(ctx, ga, gb):
        if ga is None or gb is None:   # one of the two blocks was not used downstream: its gradient is a zero block
            ref = ga if ga is not None else gb
            ga = ga if ga is not None else torch.zeros(ref.shape[0], ctx.c0, dtype=ref.dtype, device=ref.device)
            gb = gb if gb is not None else torch.zeros(ref.shape[0], ctx.c1, dtype=ref.dtype, device=ref.device)
        return torch.cat([ga, gb], dim=1), None


def split_cols(w, c0):
    return _SplitCols.apply(w, c0)


_ones_memo = {}


def _ones(*shape_and_device):
    """constant ones tensor, memoised (capturable: no fill kernel per call)"""
    *shape, device = shape_and_device
    key = (tuple(shape), str(device))
    t = _ones_memo.get(key)
    if t is None:
        if len(_ones_memo) > 64:
            _ones_memo.clear()
        t = _ones_memo[key] = torch.ones(*shape, dtype=torch.float32, device=device)
    return t


class _AddPerCloud(torch.autograd.Function):
    """y (B,N,C) + c (B,C) broadcast over the points.  The gradient of `c` is the sum over the N points of a cloud; ATen's
    reduction over the middle dimension takes 29 us for (8, 2048, 256), the same sum as a batched one-row product 6 us."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, y, c):
        return y + c.unsqueeze(1)

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        B, N, C = g.shape
        gc = g if g.is_contiguous() else g.contiguous()
        return g, torch.bmm(_ones(B, 1, N, g.device), gc).view(B, C)


def add_per_cloud(y, c):
    return _AddPerCloud.apply(y, c)


class _LinearReLU(torch.autograd.Function):
    """relu(x W^T + b) over point-major rows with the bias and the ReLU in the GEMM's epilogue (`torch._addmm_activation`:
    hipBLASLt RELU_BIAS, bit-identical to linear + relu and 16 % faster at 32768 x 512 x 512 -- no separate pass over the
    output); backward masks the gradient once and reuses the routing of `_LinearPM`."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, w, b):
        x2 = x.reshape(-1, x.shape[-1])
        ctx.bf16 = _bf16_gemm() and x.is_cuda
        if ctx.bf16:
            out = torch.relu_(_bf16_mm(x2, w.t()).add_(b))
        else:
            out = torch._addmm_activation(b, x2, w.t(), use_gelu=False)
        ctx.save_for_backward(x, w, out)
        return out.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        x, w, out = ctx.saved_tensors
        g2 = torch.ops.aten.threshold_backward(g.reshape(-1, g.shape[-1]).contiguous(), out, 0)
        x2 = x.reshape(-1, x.shape[-1])
        gx = _linear_dx(g2, w, bf16=ctx.bf16).view_as(x) if ctx.needs_input_grad[0] else None
        gw = _linear_dw(g2, x2, bf16=ctx.bf16) if ctx.needs_input_grad[1] else None
        gb = _bias_grad(g2) if ctx.needs_input_grad[2] else None
        return gx, gw, gb


def linear_pm_relu(x, w, b):
    """relu(linear(x, w, b)) for fp32 GPU rows, bias required; see _LinearReLU"""
    _need_gpu(x, w, b)
    return _LinearReLU.apply(x, w, b)


class _FoldLayer1(torch.autograd.Function):
    """First layer of a folding MLP (models/folding_net.py:205-221) after the per-cloud part has been split off:
    [relu]( per_cloud[b] + pts[b,i,:] W_p^T ) in ONE pass that writes the (B,m,Cout) activation once
    (fsg_fold_layer1_f32; as thin GEMM + broadcast add + ReLU the tensor is written three times and read twice)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, pts, w_p, per_cloud, relu):
        pts, per_cloud = _f32c(pts), _f32c(per_cloud)
        B, m, cp = pts.shape
        Cout = w_p.shape[0]
        if w_p.dtype != torch.float32 or w_p.stride(1) != 1:
            w_p = w_p.float().contiguous()
        out = torch.empty(B, m, Cout, dtype=torch.float32, device=pts.device)
        with torch.cuda.device(pts.device):
            _lib.call("fsg_fold_layer1_f32", _p(pts), cp, _p(w_p), w_p.stride(0), _p(per_cloud), B, m, Cout, int(relu), _p(out),
                      _stream())
        ctx.save_for_backward(pts, w_p, out)
        ctx.relu = bool(relu)
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        pts, w_p, out = ctx.saved_tensors
        B, m, cp = pts.shape
        Cout = w_p.shape[0]
        g = g.contiguous()
        if ctx.relu:
            g = torch.ops.aten.threshold_backward(g, out, 0)
        g2 = g.view(B * m, Cout)
        g_pts = (g2 @ w_p).view(B, m, cp) if ctx.needs_input_grad[0] else None
        g_w = _linear_dw(g2, pts.view(B * m, cp)) if ctx.needs_input_grad[1] else None
        g_pc = torch.bmm(_ones(B, 1, m, g.device), g).view(B, Cout) if ctx.needs_input_grad[2] else None
        return g_pts, g_w, g_pc, None


def fold_layer1(pts, w_p, per_cloud, relu=True):
    """pts (B,m,cp<=3), w_p (Cout,cp) (may be a column slice of the conv weight), per_cloud (B,Cout) -> (B,m,Cout)"""
    _need_gpu(pts, w_p, per_cloud)
    return _FoldLayer1.apply(pts, w_p, per_cloud, relu)


class _EdgeWeights(torch.autograd.Function):
    """W = [W_rel | W_ctr] (Co, 2C) -> [W_rel ; W_ctr - W_rel] (2Co, C): the P/Q form of the first EdgeConv layer; one
    launch forward and one backward (fsg_edge_weights_*; ATen's slice / subtract / cat chain is 2 + 4)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, W):
        Co, C2 = W.shape
        Wc = W if (W.dtype == torch.float32 and W.is_contiguous()) else W.float().contiguous()
        out = torch.empty(2 * Co, C2 // 2, dtype=torch.float32, device=W.device)
        with torch.cuda.device(W.device):
            _lib.call("fsg_edge_weights_fwd_f32", _p(Wc), Co, C2 // 2, _p(out), _stream())
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        Co, C = g.shape[0] // 2, g.shape[1]
        gc = g if (g.dtype == torch.float32 and g.is_contiguous()) else g.float().contiguous()
        out = torch.empty(Co, 2 * C, dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            _lib.call("fsg_edge_weights_bwd_f32", _p(gc), Co, C, _p(out), _stream())
        return out


class _EdgeWeightsMany(torch.autograd.Function):
    """`_EdgeWeights` for all EdgeConv layers of a model in one launch each way: the transforms depend on the weights
    only, so they can run together at the head of the forward -- and, because autograd runs a node once all its output
    gradients are in, together at the tail of the backward."""

    @staticmethod
    def _run(srcs, shapes, backward):
        dev = srcs[0].device
        jobs = _lib.EdgeWeightJobs()
        jobs.n = len(srcs)
        outs = []
        for j, (t, (Co, C)) in enumerate(zip(srcs, shapes)):
            out = torch.empty((Co, 2 * C) if backward else (2 * Co, C), dtype=torch.float32, device=dev)
            jobs.src[j], jobs.dst[j], jobs.Co[j], jobs.C[j] = t.data_ptr(), out.data_ptr(), Co, C
            outs.append(out)
        with torch.cuda.device(dev):
            _lib.call("fsg_edge_weights_many_f32", ctypes.byref(jobs), int(backward), _stream())
        return outs

    @staticmethod
    @_amp_fwd
    def forward(ctx, *Ws):
        Wc = [W if (W.dtype == torch.float32 and W.is_contiguous()) else W.float().contiguous() for W in Ws]
        ctx.shapes = [(W.shape[0], W.shape[1] // 2) for W in Wc]
        return tuple(_EdgeWeightsMany._run(Wc, ctx.shapes, False))

    @staticmethod
    @_amp_bwd
    def backward(ctx, *gs):
        dev = next(g for g in gs if g is not None).device
        gc = [torch.zeros(2 * Co, C, dtype=torch.float32, device=dev) if g is None else
              (g if (g.dtype == torch.float32 and g.is_contiguous()) else g.float().contiguous())
              for g, (Co, C) in zip(gs, ctx.shapes)]
        return tuple(_EdgeWeightsMany._run(gc, ctx.shapes, True))


def edge_weights_many(conv_weights):
    """[(Co,2C,1,1) or (Co,2C) first-layer EdgeConv weights] -> [(2Co,C) P/Q weights], one launch (up to 8 layers)"""
    Ws = [w.reshape(w.shape[0], w.shape[1]) for w in conv_weights]
    if not 1 <= len(Ws) <= 8:
        raise ValueError("edge_weights_many: 1..8 layers")
    return list(_EdgeWeightsMany.apply(*Ws))


# ------------------------------------------------------------------ BatchNorm call counters
_deferred_counters = None


def bump_bn_counter(bn):
    """`num_batches_tracked += 1` as torch's BatchNorm does on every training call; returns the momentum of this call.
    Inside `deferred_bn_counters()` the increments of a whole forward are issued as ONE multi-tensor add instead of one
    tiny kernel per BatchNorm (8 per DGCNN-seg step)."""
    if bn.momentum is None:                       # cumulative average: the value is needed now
        bn.num_batches_tracked.add_(1)
        return 1.0 / float(bn.num_batches_tracked)
    if _deferred_counters is not None:
        _deferred_counters.append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked.add_(1)
    return bn.momentum


@_contextlib.contextmanager
def deferred_bn_counters():
    global _deferred_counters
    if _deferred_counters is not None:            # nested: the outermost context flushes
        yield
        return
    _deferred_counters = []
    try:
        yield
    finally:
        pending, _deferred_counters = _deferred_counters, None
        if pending:
            torch._foreach_add_(pending, 1)


def with_deferred_bn_counters(forward):
    """decorator for a top-level model forward: one multi-tensor add for the BatchNorm call counters, and the whole
    forward outside any ambient autocast region (the plain torch glue between the HIP stages -- concatenations, the small
    per-cloud products -- would otherwise round to fp16 between fp32 kernels); a half-precision input is widened."""
    import functools

    @functools.wraps(forward)
    def wrapped(self, x, *args, **kwargs):
        mode = mfma_operands("bf16") if (x.is_cuda and _ambient_bf16()) else _contextlib.nullcontext()
        with deferred_bn_counters(), mode, no_autocast():
            if torch.is_tensor(x) and x.is_floating_point() and x.dtype != torch.float32:
                x = x.float()
            return forward(self, x, *args, **kwargs)
    return wrapped


# ------------------------------------------------------------------ fused EdgeConv (models/dgcnn.py:234-241)

# Building the reverse graphs on a side stream during the forward was MEASURED SLOWER inside the replayed hipGraph
# (2.06 vs 1.83 ms/step on MI355X: the fork/join edges cost more than the ~30 us builder hides), so it is opt-in.
_ASYNC_CSR = _os.environ.get("FSG_ASYNC_CSR", "0") == "1"
_side_streams = {}


def _side_stream(device):
    st = _side_streams.get(device)
    if st is None:
        st = _side_streams[device] = torch.cuda.Stream(device=device)
    return st


def _build_csr(idx):
    B, N, k = idx.shape
    rowptr = torch.empty(B, N + 1, dtype=torch.int32, device=idx.device)
    col = torch.empty(B, N * k, dtype=torch.int32, device=idx.device)
    ws = torch.empty(_lib.lib.fsg_graph_reverse_csr_workspace_bytes(B, N, k) // 4, dtype=torch.int32, device=idx.device)
    with torch.cuda.device(idx.device):
        _lib.call("fsg_graph_reverse_csr", _p(idx), B, N, k, _p(rowptr), _p(col), _p(ws), _stream())
    return rowptr, col


def knn_prep_workspace(B, N, C, device):
    """workspace for a graph build over (B, C, N) points whose PRODUCER prepares it (edgeconv1 / edgeconv2 with knn_ws=, then
    knn_graph(..., prepared=(ws, x_pm))); None when the shape is outside the prepared path (then build the graph as usual)"""
    if C != 64 or N % 64 != 0 or not (1024 <= N <= 8192):
        return None
    return torch.empty(_lib.lib.fsg_knn_dense_workspace_bytes(B, N, C), dtype=torch.uint8, device=device)


def build_reverse_graphs(graphs):
    """Reverse graphs (CSR by destination) of several kNN graphs of the same shape in ONE set of launches instead of one per
    graph: `graphs` are the (B,N,k) slices, in order, of one contiguous (G,B,N,k) buffer (knn_graph(..., out=slice)).  The
    results are cached on the slice tensors exactly as reverse_graph() would (3 x 3 small launches -> 3 per DGCNN-seg step)."""
    g0 = graphs[0]
    G = len(graphs)
    B, N, k = g0.shape
    step = B * N * k * g0.element_size()
    if any(g.shape != g0.shape or g.data_ptr() != g0.data_ptr() + i * step or not g.is_contiguous() for i, g in enumerate(graphs)):
        return          # not one buffer (e.g. graphs substituted by a test): each is built on first use by reverse_graph()
    if all(getattr(g, "_fsg_csr", None) is not None for g in graphs):
        return
    flat = torch.as_strided(g0, (G * B, N, k), (N * k, k, 1))
    rowptr, col = _build_csr(flat)
    for i, g in enumerate(graphs):
        g._fsg_csr = (rowptr[i * B:(i + 1) * B], col[i * B:(i + 1) * B])


def prefetch_reverse_graph(idx):
    """Start building the reverse graph of `idx` on a side stream (the builder runs 8 workgroups: ~30 us of an otherwise
    idle GPU if it sat on the main stream).  The backward joins the side stream before its first use; inside a hipGraph
    capture this becomes a parallel branch.  Only called when a backward pass will follow."""
    if getattr(idx, "_fsg_csr", None) is not None or getattr(idx, "_fsg_csr_pending", None) is not None:
        return
    main = torch.cuda.current_stream(idx.device)
    side = _side_stream(idx.device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        rowptr, col = _build_csr(idx)
        done = torch.cuda.Event()
        done.record(side)
    for t in (rowptr, col):
        t.record_stream(main)
    idx.record_stream(side)
    idx._fsg_csr_pending = (rowptr, col, done)


def reverse_graph(idx):
    """CSR-by-destination of a kNN graph, cached on the index tensor (a static graph is shared by all layers)."""
    cached = getattr(idx, "_fsg_csr", None)
    if cached is None:
        pending = getattr(idx, "_fsg_csr_pending", None)
        if pending is not None:
            rowptr, col, done = pending
            torch.cuda.current_stream(idx.device).wait_event(done)
            idx._fsg_csr_pending = None
            cached = (rowptr, col)
        else:
            cached = _build_csr(idx)
        idx._fsg_csr = cached
    return cached


def _pm_grad(g, B, N, C):
    """point-major output gradient (B,N,C) -> (tensor, row stride) for the kernels: rows of C contiguous floats with ANY
    row stride pass as they are (e.g. a slice of the gradient of concatenated features), everything else is copied"""
    if g is None:
        return None, 0
    if g.dtype == torch.float32 and g.shape == (B, N, C) and g.stride(2) == 1 and g.stride(1) >= C and \
            (B == 1 or g.stride(0) == N * g.stride(1)):
        return g, g.stride(1)
    return _f32c(g), C


def _apply_pass(ysel, gamma, beta, mean, invstd, B, N, Co, slope, out, out_pm, knn_ws, w_next):
    """the BatchNorm + LeakyReLU pass of a fused EdgeConv when it also prepares the next layer's graph build (knn_ws) and, with
    w_next (128, 64), emits the next fused EdgeConv's [P | Q] rows (returned; else None).  Inside a `with torch.cuda.device`."""
    if knn_ws is None:
        return None
    nb = knn_ws.numel() * knn_ws.element_size()
    if w_next is not None and Co == 64 and tuple(w_next.shape) == (128, 64):
        w_next = _f32c(w_next.detach())
        pq_next = torch.empty(B, N, 128, dtype=torch.float32, device=ysel.device)
        _lib.call("fsg_edgeconv_apply_pq_f32", _p(ysel), _p(gamma), _p(beta), _p(mean), _p(invstd), B, N, Co, slope, _p(out),
                  _p(out_pm), _p(knn_ws), nb, _p(w_next), 128, _p(pq_next), _stream())
        return pq_next
    _lib.call("fsg_edgeconv_apply_f32", _p(ysel), _p(gamma), _p(beta), _p(mean), _p(invstd), B, N, Co, slope, _p(out),
              _p(out_pm), _p(knn_ws), nb, _stream())
    return None


class _EdgeConv1(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, pq, idx, gamma, beta, running_mean, running_var, training, momentum, eps, slope, knn_ws=None, w_next=None):
        pq = _f32c(pq)
        B, N, two_co = pq.shape
        Co, k, dev = two_co // 2, idx.shape[2], pq.device
        gamma, beta = _f32c(gamma), _f32c(beta)
        out = torch.empty(B, Co, N, dtype=torch.float32, device=dev)
        out_pm = torch.empty(B, N, Co, dtype=torch.float32, device=dev)
        ysel = torch.empty(B, N, Co, dtype=torch.float32, device=dev)
        arg = torch.empty(B, N, Co, dtype=torch.uint8, device=dev)
        if training:
            ssum = torch.empty(B, N, Co, dtype=torch.float32, device=dev)
            mean = torch.empty(Co, dtype=torch.float32, device=dev)
            invstd = torch.empty(Co, dtype=torch.float32, device=dev)
            ws = torch.empty(_lib.lib.fsg_edgeconv1_workspace_bytes(B, N, Co) // 4, dtype=torch.float32, device=dev)
        else:
            ssum, ws = None, None
            mean = running_mean.detach().float().contiguous()
            invstd = torch.rsqrt(running_var.detach().float() + eps).contiguous()
        with torch.cuda.device(dev):
            _lib.call("fsg_edgeconv1_fwd_f32", _p(pq), _p(idx), _p(gamma), _p(beta),
                      _p(running_mean if training else None), _p(running_var if training else None), B, N, k, Co,
                      int(training), momentum, eps, slope, _p(None if knn_ws is not None else out), _p(out_pm), _p(ysel), _p(arg),
                      _p(ssum), _p(mean), _p(invstd), _p(ws), _stream())
            pq_next = _apply_pass(ysel, gamma, beta, mean, invstd, B, N, Co, slope, out, out_pm, knn_ws, w_next)
        ctx.save_for_backward(pq, idx, gamma, beta, mean, invstd, ysel, arg, ssum)
        ctx.meta = (B, N, k, Co, bool(training), slope)
        ctx.set_materialize_grads(False)   # the layout that only feeds the next graph build gets None, not a zero tensor
        # the point-major output twice (second one an alias): two consumers -- the next layer and the concatenated
        # features -- then deliver their gradients separately and the backward kernel sums them (no ATen add, no copy)
        if pq_next is None:
            return out, out_pm, out_pm.view(B, N, Co)
        ctx.mark_non_differentiable(pq_next)     # (its gradient arrives through _LinearPMGiven on out_pm and the weight)
        return out, out_pm, out_pm.view(B, N, Co), pq_next

    @staticmethod
    @_amp_bwd
    def backward(ctx, g, g_pm, g_pm2, _g_pq_next=None):
        pq, idx, gamma, beta, mean, invstd, ysel, arg, ssum = ctx.saved_tensors
        B, N, k, Co, training, slope = ctx.meta
        dev = pq.device
        g = _f32c(g) if g is not None else None
        g_pm, ld1 = _pm_grad(g_pm, B, N, Co)
        g_pm2, ld2 = _pm_grad(g_pm2, B, N, Co)
        rowptr, col = reverse_graph(idx)
        gpq = torch.empty_like(pq)
        dgamma = torch.empty(Co, dtype=torch.float32, device=dev)
        dbeta = torch.empty(Co, dtype=torch.float32, device=dev)
        h = torch.empty(B, N, Co, dtype=torch.float32, device=dev)
        ws = torch.empty(B * ((N + 63) // 64) * 2 * Co, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.call("fsg_edgeconv1_bwd_f32", _p(g), _p(g_pm), ld1, _p(g_pm2), ld2, _p(pq), _p(rowptr), _p(col), _p(gamma),
                      _p(beta), _p(mean),
                      _p(invstd), _p(ysel), _p(arg), _p(ssum), B, N, k, Co, int(training), slope, _p(gpq), _p(dgamma),
                      _p(dbeta), _p(h), _p(ws), _stream())
        return gpq, None, dgamma, dbeta, None, None, None, None, None, None, None, None


def edgeconv1_supported(out_channels, k):
    return out_channels % 64 == 0 and k <= 64


def edgeconv1(x, idx, conv_weight, bn, slope, x_pm=None, both=False, w_cat=None, knn_ws=None, w_next=None, pq_given=None):
    """Fused single-layer EdgeConv: x (B,C,N), idx (B,N,k) int32, conv_weight (Co,2C,1,1), bn a BatchNorm2d module
    (its running statistics are updated in place like nn.BatchNorm2d does) -> (B,Co,N); with both=True also the
    point-major copy (B,N,Co).  x_pm: optional point-major (B,N,C) copy of x (saves the transpose for the GEMM)."""
    _need_gpu(x, idx, conv_weight)
    Co, C2 = conv_weight.shape[0], conv_weight.shape[1]
    C = C2 // 2
    if w_cat is None:        # (2Co, C): [W_rel ; W_ctr - W_rel]; models hand in the batch of edge_weights_many instead
        w_cat = _EdgeWeights.apply(conv_weight.reshape(Co, C2).to(torch.float32))
    if x_pm is None:
        x_pm = x.transpose(1, 2)
    if pq_given is not None:    # emitted by the previous block's apply pass (w_next): no GEMM launch, ordinary backward
        pq = _LinearPMGiven.apply(x_pm.to(torch.float32), w_cat, pq_given)
    else:
        pq = linear_pm(x_pm.to(torch.float32).contiguous(), w_cat)               # (B,N,2Co): one plain GEMM
    training = bn.training or bn.running_mean is None
    momentum = 0.0
    if training and bn.track_running_stats and bn.num_batches_tracked is not None:
        momentum = bump_bn_counter(bn)
    track = training and bn.track_running_stats
    if idx.dtype != torch.int32:
        idx = idx.to(torch.int32)
    idx = idx.contiguous()
    if torch.is_grad_enabled() and pq.requires_grad and _ASYNC_CSR:
        prefetch_reverse_graph(idx)
    res = _EdgeConv1.apply(pq, idx, bn.weight, bn.bias,
                           bn.running_mean if (track or not training) else None,
                           bn.running_var if (track or not training) else None, training, float(momentum),
                           float(bn.eps), float(slope), knn_ws, w_next if knn_ws is not None else None)
    return _edgeconv_outputs(res, both, w_next is not None)


class _EdgeConv2(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, pq, idx, w2, g1, b1, rm1, rv1, g2, b2, rm2, rv2, training, mom1, mom2, eps1, eps2, slope, knn_ws=None,
                w_next=None):
        pq, w2 = _f32c(pq), _f32c(w2)
        g1, b1, g2, b2 = _f32c(g1), _f32c(b1), _f32c(g2), _f32c(b2)
        B, N, _ = pq.shape
        C2, k, dev = w2.shape[0], idx.shape[2], pq.device

        def new(*shape, dtype=torch.float32):
            return torch.empty(*shape, dtype=dtype, device=dev)
        out, out_pm = new(B, C2, N), new(B, N, C2)
        ysel2, arg2 = new(B, N, C2), new(B, N, C2, dtype=torch.uint8)
        if training:
            ssum1, ssum2 = new(B, N, 64), new(B, N, C2)
            mean1, invstd1, mean2, invstd2 = new(64), new(64), new(C2), new(C2)
        else:
            ssum1 = ssum2 = None
            mean1, invstd1 = rm1.detach().float().contiguous(), torch.rsqrt(rv1.detach().float() + eps1).contiguous()
            mean2, invstd2 = rm2.detach().float().contiguous(), torch.rsqrt(rv2.detach().float() + eps2).contiguous()
        ws = new(_lib.lib.fsg_edgeconv2_workspace_bytes(B, N, k, C2), dtype=torch.uint8)
        t = bool(training)
        ctx.bf16 = bf16_operands()
        with torch.cuda.device(dev):
            _lib.call("fsg_edgeconv2_fwd_bf16" if ctx.bf16 else "fsg_edgeconv2_fwd_f32", _p(pq), _p(idx), _p(w2), _p(g1), _p(b1),
                      _p(rm1 if t else None),
                      _p(rv1 if t else None), _p(g2), _p(b2), _p(rm2 if t else None), _p(rv2 if t else None), B, N, k, C2,
                      int(t), mom1, mom2, eps1, eps2, slope, _p(None if knn_ws is not None else out), _p(out_pm), _p(ssum1), _p(mean1),
                      _p(invstd1), _p(ysel2), _p(arg2), _p(ssum2), _p(mean2), _p(invstd2), _p(ws), _stream())
            pq_next = _apply_pass(ysel2, g2, b2, mean2, invstd2, B, N, C2, slope, out, out_pm, knn_ws, w_next)
        ctx.save_for_backward(pq, idx, w2, g1, b1, mean1, invstd1, ssum1, g2, b2, mean2, invstd2, ysel2, arg2)
        ctx.meta = (B, N, k, C2, t, slope)
        ctx.set_materialize_grads(False)
        if pq_next is None:
            return out, out_pm, out_pm.view(B, N, C2)    # alias for a second consumer, see _EdgeConv1
        ctx.mark_non_differentiable(pq_next)
        return out, out_pm, out_pm.view(B, N, C2), pq_next

    @staticmethod
    @_amp_bwd
    def backward(ctx, g, g_pm, g_pm2, _g_pq_next=None):
        pq, idx, w2, g1, b1, mean1, invstd1, ssum1, g2, b2, mean2, invstd2, ysel2, arg2 = ctx.saved_tensors
        B, N, k, C2, training, slope = ctx.meta
        dev = pq.device
        g = _f32c(g) if g is not None else None
        g_pm, ld1 = _pm_grad(g_pm, B, N, C2)
        g_pm2, ld2 = _pm_grad(g_pm2, B, N, C2)
        rowptr, col = reverse_graph(idx)
        gpq, gw2 = torch.empty_like(pq), torch.empty_like(w2)
        dg1, db1 = torch.empty_like(g1), torch.empty_like(b1)
        dg2, db2 = torch.empty_like(g2), torch.empty_like(b2)
        ws = torch.empty(_lib.lib.fsg_edgeconv2_bwd_workspace_bytes(B, N, k, C2), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.call("fsg_edgeconv2_bwd_bf16" if ctx.bf16 else "fsg_edgeconv2_bwd_f32", _p(g), _p(g_pm), ld1, _p(g_pm2), ld2,
                      _p(pq), _p(idx), _p(rowptr), _p(col),
                      _p(w2), _p(g1),
                      _p(b1), _p(mean1), _p(invstd1), _p(ssum1), _p(g2), _p(b2), _p(mean2), _p(invstd2), _p(ysel2),
                      _p(arg2), B, N, k, C2, int(training), slope, _p(gpq), _p(gw2), _p(dg1), _p(db1), _p(dg2), _p(db2),
                      _p(ws), _stream())
        return (gpq, None, gw2, dg1, db1, None, None, dg2, db2) + (None,) * 10


def edgeconv2_supported(c_mid, c_out, k):
    return c_mid == 64 and c_out in (64, 128) and k <= 64


def _bn_step(bn):
    """training flag + momentum of one BatchNorm module for this call (bumps num_batches_tracked like torch)."""
    training = bn.training or bn.running_mean is None
    momentum = 0.0
    if training and bn.track_running_stats and bn.num_batches_tracked is not None:
        momentum = bump_bn_counter(bn)
    return training, float(momentum)


def _edgeconv_outputs(res, both, want_pq_next):
    """(out, out_pm, alias[, pq_next]) of a fused EdgeConv Function -> what the module interface hands out; with w_next given the
    next block's rows come last (None when the pass could not emit them: no knn workspace, other widths)"""
    out, out_pm, out_pm2 = res[:3]
    pq_next = res[3] if len(res) > 3 else None
    if both == "twice":      # (B,Co,N), (B,N,Co) and an alias of the latter for a second consumer (see _EdgeConv1)
        r = (out, out_pm, out_pm2)
    elif both:
        r = (out, out_pm)
    else:
        r = (out,)
    if want_pq_next:
        return r + (pq_next,)
    return r if len(r) > 1 else r[0]


def edgeconv2(x, idx, conv1_weight, bn1, conv2_weight, bn2, slope, x_pm=None, both=False, w_cat=None, knn_ws=None, w_next=None,
              pq_given=None):
    """Fused two-layer EdgeConv (2C -> 64 -> 64|128): see csrc/edgeconv2.hip."""
    _need_gpu(x, idx, conv1_weight, conv2_weight)
    C1, CC = conv1_weight.shape[0], conv1_weight.shape[1]
    C = CC // 2
    if w_cat is None:
        w_cat = _EdgeWeights.apply(conv1_weight.reshape(C1, CC).to(torch.float32))
    if x_pm is None:
        x_pm = x.transpose(1, 2)
    if pq_given is not None:
        pq = _LinearPMGiven.apply(x_pm.to(torch.float32), w_cat, pq_given)
    else:
        pq = linear_pm(x_pm.to(torch.float32).contiguous(), w_cat)               # (B,N,128)
    w2 = conv2_weight.reshape(conv2_weight.shape[0], C1)
    t1, m1 = _bn_step(bn1)
    t2, m2 = _bn_step(bn2)
    if t1 != t2:
        raise RuntimeError("edgeconv2: both BatchNorm layers must be in the same mode")
    if idx.dtype != torch.int32:
        idx = idx.to(torch.int32)
    idx = idx.contiguous()
    if torch.is_grad_enabled() and pq.requires_grad and _ASYNC_CSR:
        prefetch_reverse_graph(idx)
    res = _EdgeConv2.apply(pq, idx, w2, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var,
                           bn2.weight, bn2.bias, bn2.running_mean, bn2.running_var, t1, m1, m2,
                           float(bn1.eps), float(bn2.eps), float(slope), knn_ws, w_next if knn_ws is not None else None)
    return _edgeconv_outputs(res, both, w_next is not None)


# ------------------------------------------------------------------ point-wise layers on the bf16 matrix pipe, fp32-grade
def pw_weight_image(w, scale=1.0, out=None, ks0=0, KS=None):
    """(N, K) fp32 weight (any strides: a view of a wider matrix, or a transposed view) -> the three-piece bf16 operand image of
    csrc/pointwise.hip (include/fsg_hip.h: fsg_pw_weight_image_f32).  `out` / `ks0` / `KS`: fill k-steps [ks0, ks0 + K/16) of
    an existing image that concatenates several matrices along k."""
    _need_gpu(w)
    N, K = w.shape
    ks = (K + 15) // 16
    KS = ks if KS is None else KS
    if out is None:
        out = torch.empty(((N + 31) // 32) * KS * 3 * 1024, dtype=torch.uint8, device=w.device)
    with torch.cuda.device(w.device):
        _lib.call("fsg_pw_weight_image_f32", _p(w), w.stride(0), w.stride(1), N, K, float(scale), ks0, KS, _p(out), _stream())
    return out


def pw_weight_images(specs):
    """several weight images in ONE launch: specs = [(w (N, K) view, scale, out uint8 tensor or None, ks0, KS or None), ...];
    returns the image tensors (include/fsg_hip.h: fsg_pw_weight_images_f32)"""
    jobs = _lib.PWImageJobs()
    outs = []
    assert 1 <= len(specs) <= _lib.PW_MAX_IMAGE_JOBS
    for j, (w, scale, out, ks0, KS) in enumerate(specs):
        N, K = w.shape
        KS = (K + 15) // 16 if KS is None else KS
        if out is None:
            out = torch.empty(((N + 31) // 32) * KS * 3 * 1024, dtype=torch.uint8, device=w.device)
        jobs.W[j], jobs.stride_n[j], jobs.stride_k[j], jobs.N[j], jobs.K[j] = w.data_ptr(), w.stride(0), w.stride(1), N, K
        jobs.ks0[j], jobs.KS[j], jobs.scale[j], jobs.image[j] = ks0, KS, float(scale), out.data_ptr()
        outs.append(out)
    jobs.n = len(specs)
    with torch.cuda.device(specs[0][0].device):
        _lib.call("fsg_pw_weight_images_f32", ctypes.byref(jobs), _stream())
    return outs


# bf16 operands for the nn.Linear products (forward, dX, dW on fsg_pw_linear_bf16 / fsg_pw_tn_bf16) are a SEPARATE opt-in of the
# bf16 mode: measured on the PointTransformer (BASELINE config 3, 2 x 2048 points against the fp32 oracle) they cost accuracy --
# mean |logit error| 0.095, max 0.86, parameter-gradient cosine 0.36 through its 60 BatchNorms and 18 softmax layers -- and time
# (9.0 vs 8.3 ms per step: three more small launches per Linear).  FSG_BF16_LINEAR=1 / set_bf16_linear(True) /
# bench.py --workload c3 --dtype bf16 switch them on.
_bf16_linear = _os.environ.get("FSG_BF16_LINEAR", "0") == "1"


def set_bf16_linear(flag):
    """bf16 operands for nn.Linear products while the bf16 operand mode is on (see above); returns the old value"""
    global _bf16_linear
    old, _bf16_linear = _bf16_linear, bool(flag)
    return old


def _pw_bf16_ok(x2, w):
    """bf16 Linear products requested, and the product inside the envelope of fsg_pw_linear_bf16 (fp32 GPU rows, K % 32 == 0)"""
    return (_bf16_linear and bf16_operands() and x2.is_cuda and x2.dtype == torch.float32 and w.dtype == torch.float32 and x2.dim() == 2 and
            x2.stride(1) == 1 and x2.stride(0) % 4 == 0 and x2.data_ptr() % 16 == 0 and x2.shape[1] % 32 == 0 and x2.shape[0] > 0)


def pw_linear_bf16(x, w, bias=None, tile=0, out=None):
    """y (M, N) = bf16(x) bf16(w)^T (+ bias), fp32 accumulation: one-piece mode of csrc/pointwise.hip (include/fsg_hip.h:
    fsg_pw_weight_image_bf16 + fsg_pw_linear_bf16).  w (N, K): any strides (a transposed view gives dX = dY W)."""
    M, K = x.shape
    N = w.shape[0]
    dev = x.device
    img = torch.empty(((N + 31) // 32) * ((K + 15) // 16) * 1024, dtype=torch.uint8, device=dev)
    y = torch.empty(M, N, dtype=torch.float32, device=dev) if out is None else out
    with torch.cuda.device(dev):
        _lib.call("fsg_pw_weight_image_bf16", _p(w), w.stride(0), w.stride(1), N, K, _p(img), _stream())
        _lib.call("fsg_pw_linear_bf16", _p(x), x.stride(0), _p(img), _p(bias.contiguous() if bias is not None else None), _p(y), N,
                  M, N, K, tile, _stream())
    return y


def pw_tn_bf16(g, x, rows_per_slice=None):
    """dW (N, K) = bf16(g)^T bf16(x) for g (M, N), x (M, K) contiguous rows, fp32 accumulation over the M rows in fixed slice
    order (include/fsg_hip.h: fsg_pw_tn_bf16)"""
    M, N = g.shape
    K = x.shape[1]
    dev = g.device
    if rows_per_slice is None:      # enough slices to fill the chip, never below 64 rows
        tiles = ((N + 63) // 64) * ((K + 63) // 64)
        rows_per_slice = max(64, min(1024, ((M * tiles // 512) // 32) * 32)) if M * tiles >= 512 * 64 else 64
    a = _lib.PWTnArgs()
    a.L1, a.ldl1, a.N1a, a.N1b, a.lpro = g.data_ptr(), g.stride(0), N, 0, 0
    a.R, a.ldr, a.N2, a.rpro = x.data_ptr(), x.stride(0), K, 0
    a.M, a.rows_per_cloud, a.rows_per_slice = M, 0, rows_per_slice
    nbytes = _lib.lib.fsg_pw_tn_workspace_bytes(N, K, M, rows_per_slice)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    out = torch.empty(N, K, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.call("fsg_pw_tn_bf16", ctypes.byref(a), 3, _p(ws), nbytes, _p(out), K, _stream())
    return out


def pw_linear(x, image, N, bias=None, tile=0):
    """y (M, N) = x (M, K) W^T (+ bias) with W given as pw_weight_image(W): six bf16 MFMA products per fp32 product, fp32
    accumulation (include/fsg_hip.h: fsg_pw_linear_f32)"""
    _need_gpu(x)
    M, K = x.shape
    assert x.stride(1) == 1 and x.dtype == torch.float32
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.call("fsg_pw_linear_f32", _p(x), x.stride(0), _p(image), _p(bias), _p(y), N, M, N, K, tile, _stream())
    return y

PW_STORE, PW_STATS, PW_SEL, PW_BWDSTATS, PW_BIAS = 1, 2, 4, 8, 16
PRO_NONE, PRO_BNACT, PRO_BNBWD = 0, 1, 2


def _ptr(t):
    return t.data_ptr() if t is not None else None


def pw_rowgemm(pro, epi, tile, **kw):
    """include/fsg_hip.h: fsg_pw_rowgemm_f32 -- keyword arguments are the fields of fsg_pw_rowgemm_args (tensors or numbers)"""
    a = _lib.PWRowGemmArgs()
    keep = []
    for k, v in kw.items():
        if torch.is_tensor(v):
            keep.append(v)
            v = v.data_ptr()
        setattr(a, k, v)
    dev = keep[0].device
    with torch.cuda.device(dev):
        _lib.call("fsg_pw_rowgemm_f32", ctypes.byref(a), pro, epi, tile, _stream())


def pw_tn(tile, C1, ldc1, C2=None, ldc2=0, defer=None, **kw):
    """include/fsg_hip.h: fsg_pw_tn_f32.  `defer`: a list -- the slices stay in their workspace and the job is appended to the
    list, to be folded by `pw_tn_reduce(list)` together with the other products' (one launch instead of one per product)"""
    a = _lib.PWTnArgs()
    keep = []
    for k, v in kw.items():
        if torch.is_tensor(v):
            keep.append(v)
            v = v.data_ptr()
        setattr(a, k, v)
    dev = C1.device
    N1 = a.N1a + a.N1b + (1 if a.ones else 0)
    nbytes = _lib.lib.fsg_pw_tn_workspace_bytes(N1, a.N2, a.M, a.rows_per_slice)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        if defer is None:
            _lib.call("fsg_pw_tn_f32", ctypes.byref(a), tile, _p(ws), nbytes, _p(C1), ldc1, _p(C2), ldc2, _stream())
        else:
            _lib.call("fsg_pw_tn_f32", ctypes.byref(a), tile, _p(ws), nbytes, None, 0, None, 0, _stream())
            defer.append((ws, C1, ldc1, C2, ldc2, (a.M + a.rows_per_slice - 1) // a.rows_per_slice, N1, a.N2, a.N1a))


def pw_tn_reduce(jobs):
    """include/fsg_hip.h: fsg_pw_tn_reduce_f32 -- fold the slices of the deferred products of pw_tn, one launch"""
    j = _lib.PWTnReduceJobs()
    assert 1 <= len(jobs) <= _lib.PW_MAX_REDUCE_JOBS
    for i, (ws, C1, ldc1, C2, ldc2, S, N1, N2, N1a) in enumerate(jobs):
        j.workspace[i], j.C1[i], j.ldc1[i], j.C2[i], j.ldc2[i] = ws.data_ptr(), C1.data_ptr(), ldc1, _ptr(C2), ldc2
        j.S[i], j.N1[i], j.N2[i], j.N1a[i] = S, N1, N2, N1a
    j.n = len(jobs)
    with torch.cuda.device(jobs[0][0].device):
        _lib.call("fsg_pw_tn_reduce_f32", ctypes.byref(j), _stream())


def _pw_bn_finalize(rec, R, ldn, c0, C, shift, B, bn, training, momentum, with_emu=False, with_cloud_mean=False, gfeat=None,
                    wglob=None):
    """statistics + consumer tables of one BatchNorm of the fused head; returns (mean, invstd, alpha, delta, emu, cloud_mean).
    `gfeat` (B, CG) + `wglob` (C, CG) view: the per-cloud shift gfeat wglob^T is formed inside the kernel (written to `shift`)"""
    dev = bn.weight.device
    nb = B if shift is not None else 1
    alpha = torch.empty(C, dtype=torch.float32, device=dev)
    delta = torch.empty(nb, C, dtype=torch.float32, device=dev)
    emu = torch.empty(nb, C, dtype=torch.float32, device=dev) if with_emu else None
    cm = torch.empty(B, C, dtype=torch.float32, device=dev) if (with_cloud_mean and training) else None
    if training:
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        invstd = torch.empty(C, dtype=torch.float32, device=dev)
        track = bn.track_running_stats and bn.running_mean is not None
        rm, rv = (bn.running_mean, bn.running_var) if track else (None, None)
    else:
        mean = bn.running_mean.detach().float().contiguous()
        invstd = torch.rsqrt(bn.running_var.detach().float() + bn.eps).contiguous()
        rm = rv = None
    with torch.cuda.device(dev):
        _lib.call("fsg_pw_bn_finalize_f32", _p(rec), R, ldn, c0, C, _p(shift), B, int(training), _p(bn.weight), _p(bn.bias),
                  float(bn.eps), float(momentum), _p(rm), _p(rv), _p(mean), _p(invstd), _p(alpha), _p(delta), _p(emu), _p(cm),
                  _p(gfeat), _p(wglob), wglob.stride(0) if wglob is not None else 0, gfeat.shape[1] if gfeat is not None else 0,
                  _p(shift) if gfeat is not None else None, _stream())
    return mean, invstd, alpha, delta, emu, cm


class _SegHead(torch.autograd.Function):
    """The whole point-wise head of DGCNNSeg behind the three EdgeConvs (models/dgcnn.py:123-162 of the reference) as ONE
    autograd node on the fused kernels of csrc/pointwise.hip:
      levels (M, 192) -> [Wg ; W0_levels] product: BatchNorm statistics + per-cloud max of the 1024 global-feature channels in
      the epilogue (the (M, 1024) activation never exists), y0 stored -> g (B, 1024) -> c = g W0_global^T per cloud ->
      BN0 (statistics of y0 + c[cloud] from the per-cloud records) -> layer 1, layer 2 with BatchNorm + LeakyReLU of the
      producer applied in the prologue and statistics in the epilogue -> class logits.
    Backward: BatchNorm backward formed in the prologues (dy never stored), its sums in the epilogue of the product that makes
    the activation gradient, weight gradients by the row-contraction kernel, the global-feature layer in its Gram form."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, levels, B, Npts, slope, bns, steps, Wg, gg, bg, W0, g0, b0, W1, g1, b1, W2, g2, b2, W3, b3):
        M, KL = levels.shape
        dev = levels.device
        CG, C0, C1, C2, CLS = Wg.shape[0], W0.shape[0], W1.shape[0], W2.shape[0], W3.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        (tr_g, mom_g), (tr_0, mom_0), (tr_1, mom_1), (tr_2, mom_2) = steps
        bn_g, bn_0, bn_1, bn_2 = bns
        # weight images (three bf16 pieces in MFMA operand layout)
        ksl = KL // 16
        img0 = torch.empty(((CG + C0) // 32) * ksl * 3 * 1024, dtype=torch.uint8, device=dev)
        ks_a, ks_b = C0 // 16, KL // 16
        img_lv = torch.empty((KL // 32) * (ks_a + ks_b) * 3 * 1024, dtype=torch.uint8, device=dev)   # [W0_levels^T ; -M1]: backward
        specs = [(Wg, 1.0, img0, 0, None), (W0[:, :KL], 1.0, img0[(CG // 32) * ksl * 3 * 1024:], 0, None), (W1, 1.0, None, 0, None),
                 (W2, 1.0, None, 0, None), (W3, 1.0, None, 0, None)]
        # (forward runs with grad mode off: whether a backward can follow is what the tensors require)
        need_bwd = any(t.requires_grad for t in (levels, Wg, gg, bg, W0, g0, b0, W1, g1, b1, W2, g2, b2, W3) if t is not None) or \
            (b3 is not None and b3.requires_grad)
        if need_bwd:    # transposed images for the backward products ride in the same launch
            specs += [(W2.t(), 1.0, None, 0, None), (W1.t(), 1.0, None, 0, None), (W0[:, :KL].t(), 1.0, img_lv, 0, ks_a + ks_b)]
        imgs = pw_weight_images(specs)
        img1, img2, img3 = imgs[2], imgs[3], imgs[4]
        img2t, img1t = (imgs[5], imgs[6]) if need_bwd else (None, None)
        # levels -> global-feature statistics / selection + y0
        R0 = M // 128
        rec0 = torch.empty(R0, 3, CG + C0, **f32)
        sel_val = torch.empty(R0, CG, **f32)
        sel_arg = torch.empty(R0, CG, dtype=torch.int32, device=dev)
        y0 = torch.empty(M, C0, **f32)
        sgn = gg                     # only the sign of the BatchNorm weight is used (max of sgn * y through the monotone BN + LeakyReLU)
        pw_rowgemm(PRO_NONE, PW_STORE | PW_STATS | PW_SEL, 1, A1=levels, lda1=levels.stride(0), K1=KL, K2=0, Bimg=img0, M=M,
                   N=CG + C0, rows_per_cloud=Npts, C=y0, ldc=C0, store_n0=CG, rec=rec0, sgn=sgn, sel_val=sel_val,
                   sel_arg=sel_arg, sel_n=CG)
        # BatchNorm statistics of the global-feature layer AND the finish of its max-pool (one launch: fsg_pw_bn_finalize_max_f32)
        g = torch.empty(B, CG, **f32)
        ysel = torch.empty(B, CG, **f32)
        arg = torch.empty(B, CG, dtype=torch.int32, device=dev)
        al_g, de_g = torch.empty(CG, **f32), torch.empty(1, CG, **f32)
        if tr_g:
            mean_g, inv_g = torch.empty(CG, **f32), torch.empty(CG, **f32)
            track = bn_g.track_running_stats and bn_g.running_mean is not None
            rm_g, rv_g = (bn_g.running_mean, bn_g.running_var) if track else (None, None)
        else:
            mean_g = bn_g.running_mean.detach().float().contiguous()
            inv_g = torch.rsqrt(bn_g.running_var.detach().float() + bn_g.eps).contiguous()
            rm_g = rv_g = None
        with torch.cuda.device(dev):
            _lib.call("fsg_pw_bn_finalize_max_f32", _p(rec0), R0, CG + C0, 0, CG, B, int(tr_g), _p(bn_g.weight), _p(bn_g.bias),
                      float(bn_g.eps), float(mom_g), _p(rm_g), _p(rv_g), _p(mean_g), _p(inv_g), _p(al_g), _p(de_g), _p(sel_val),
                      _p(sel_arg), _p(sgn), Npts // 128, slope, _p(g), _p(ysel), _p(arg), _stream())
        c = torch.empty(B, C0, **f32)     # g W0_global^T: the global part of the first head layer, one wave per output
        with torch.cuda.device(dev):
            _lib.call("fsg_pw_cloud_linear_f32", _p(g), _p(W0[:, KL:]), W0.stride(0), B, C0, CG, _p(c), _stream())
        mean_0, inv_0, al_0, de_0, emu_0, cm_0 = _pw_bn_finalize(rec0, R0, CG + C0, CG, C0, c, B, bn_0, tr_0, mom_0,
                                                                 with_emu=True, with_cloud_mean=True)
        # layer 1, layer 2, logits
        R1 = M // 64
        y1 = torch.empty(M, C1, **f32)
        rec1 = torch.empty(R1, 3, C1, **f32)
        pw_rowgemm(PRO_BNACT, PW_STORE | PW_STATS, 2, A1=y0, lda1=C0, K1=C0, K2=0, Bimg=img1, M=M, N=C1, rows_per_cloud=Npts,
                   alpha=al_0, delta=de_0, tstride=C0, slope=slope, C=y1, ldc=C1, store_n0=0, rec=rec1)
        mean_1, inv_1, al_1, de_1, _, _ = _pw_bn_finalize(rec1, R1, C1, 0, C1, None, B, bn_1, tr_1, mom_1)
        y2 = torch.empty(M, C2, **f32)
        rec2 = torch.empty(R1, 3, C2, **f32)
        pw_rowgemm(PRO_BNACT, PW_STORE | PW_STATS, 3, A1=y1, lda1=C1, K1=C1, K2=0, Bimg=img2, M=M, N=C2, rows_per_cloud=Npts,
                   alpha=al_1, delta=de_1, tstride=0, slope=slope, C=y2, ldc=C2, store_n0=0, rec=rec2)
        mean_2, inv_2, al_2, de_2, _, _ = _pw_bn_finalize(rec2, R1, C2, 0, C2, None, B, bn_2, tr_2, mom_2)
        out = torch.empty(M, CLS, **f32)
        pw_rowgemm(PRO_BNACT, PW_STORE | PW_BIAS, 3, A1=y2, lda1=C2, K1=C2, K2=0, Bimg=img3, M=M, N=CLS, rows_per_cloud=Npts,
                   alpha=al_2, delta=de_2, tstride=0, slope=slope, C=out, ldc=CLS, store_n0=0, bias=b3)
        if not need_bwd:       # inference: nothing kept
            return out
        ctx.images = (img2t, img1t, img_lv)
        ctx.save_for_backward(levels, y0, y1, y2, Wg, W0, W1, W2, W3, g, ysel, arg, c,
                              mean_g, inv_g, al_g, de_g, mean_0, inv_0, al_0, de_0, emu_0,
                              cm_0 if cm_0 is not None else torch.zeros(B, C0, **f32), mean_1, inv_1, al_1, de_1, mean_2, inv_2, al_2, de_2)
        ctx.meta = (B, Npts, slope, (tr_g, tr_0, tr_1, tr_2))
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, gout):
        (levels, y0, y1, y2, Wg, W0, W1, W2, W3, g, ysel, arg, c, mean_g, inv_g, al_g, de_g, mean_0, inv_0, al_0, de_0, emu_0,
         cm_0, mean_1, inv_1, al_1, de_1, mean_2, inv_2, al_2, de_2) = ctx.saved_tensors
        B, Npts, slope, (tr_g, tr_0, tr_1, tr_2) = ctx.meta
        img2t, img1t, img_lv = ctx.images
        M, KL = levels.shape
        dev = levels.device
        CG, C0, C1, C2, CLS = Wg.shape[0], W0.shape[0], W1.shape[0], W2.shape[0], W3.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        gout = gout if gout.is_contiguous() else gout.contiguous()

        def call(name, *a):
            with torch.cuda.device(dev):
                _lib.call(name, *a, _stream())
        # ---- logits layer
        db3 = _bias_grad(gout)
        dW3 = torch.empty(CLS, C2, **f32)
        folds = []                      # the four weight-gradient products leave their row slices; one launch folds them at the end
        pw_tn(3, dW3, C2, defer=folds, L1=gout, ldl1=CLS, N1a=CLS, N1b=0, lpro=PRO_NONE, R=y2, ldr=C2, N2=C2, rpro=PRO_BNACT, ralpha=al_2,
              rdelta=de_2, rts=0, slope=slope, M=M, rows_per_cloud=Npts, rows_per_slice=128)
        da2 = torch.empty(M, C2, **f32)
        Rb = (M + 31) // 32
        r2b = torch.empty(Rb, 2, C2, **f32)
        call("fsg_pw_logits_bwd_f32", _p(gout), CLS, _p(W3), _p(y2), _p(al_2), _p(de_2), _p(mean_2), _p(inv_2), M, C2, slope,
             _p(da2), _p(r2b))

        def bwd_fin(rec, R, C, training, alpha, inv, emu, per_cloud, cm=None, want_dc=False):
            nb = B if per_cloud else 1
            dbeta, dgamma = torch.empty(C, **f32), torch.empty(C, **f32)
            P, Q = torch.empty(nb, C, **f32), torch.empty(C, **f32)
            dc = torch.empty(B, C, **f32) if want_dc else None
            call("fsg_pw_bnbwd_finalize_f32", _p(rec), R, C, B, M, int(training), _p(alpha), _p(inv), _p(emu), int(per_cloud),
                 _p(cm), _p(dbeta), _p(dgamma), _p(P), _p(Q), _p(dc))
            return dbeta, dgamma, P, Q, dc
        db2, dg2, P2, Q2, _ = bwd_fin(r2b, Rb, C2, tr_2, al_2, inv_2, mean_2, False)
        # ---- layer 2
        dW2 = torch.empty(C2, C1, **f32)
        pw_tn(2, dW2, C1, defer=folds, L1=da2, LY1=y2, ldl1=C2, N1a=C2, N1b=0, lpro=PRO_BNBWD, lalpha=al_2, ldelta=de_2, lP=P2, lQ=Q2, lts=0,
              R=y1, ldr=C1, N2=C1, rpro=PRO_BNACT, ralpha=al_1, rdelta=de_1, rts=0, slope=slope, M=M, rows_per_cloud=Npts,
              rows_per_slice=_tn_rps(M))
        R1 = M // 64
        da1 = torch.empty(M, C1, **f32)
        r1b = torch.empty(R1, 2, C1, **f32)
        pw_rowgemm(PRO_BNBWD, PW_STORE | PW_BWDSTATS, 2, A1=da2, Y1=y2, lda1=C2, K1=C2, K2=0, Bimg=img2t, M=M,
                   N=C1, rows_per_cloud=Npts, alpha=al_2, delta=de_2, P=P2, Q=Q2, tstride=0, slope=slope, C=da1, ldc=C1,
                   store_n0=0, Yp=y1, ldyp=C1, ealpha=al_1, edelta=de_1, emu=mean_1, er=inv_1, etstride=0, rec2=r1b)
        db1, dg1, P1, Q1, _ = bwd_fin(r1b, R1, C1, tr_1, al_1, inv_1, mean_1, False)
        # ---- layer 1
        dW1 = torch.empty(C1, C0, **f32)
        pw_tn(1, dW1, C0, defer=folds, L1=da1, LY1=y1, ldl1=C1, N1a=C1, N1b=0, lpro=PRO_BNBWD, lalpha=al_1, ldelta=de_1, lP=P1, lQ=Q1, lts=0,
              R=y0, ldr=C0, N2=C0, rpro=PRO_BNACT, ralpha=al_0, rdelta=de_0, rts=C0, slope=slope, M=M, rows_per_cloud=Npts,
              rows_per_slice=_tn_rps(M))
        da0 = torch.empty(M, C0, **f32)
        r0b = torch.empty(R1, 2, C0, **f32)
        pw_rowgemm(PRO_BNBWD, PW_STORE | PW_BWDSTATS, 2, A1=da1, Y1=y1, lda1=C1, K1=C1, K2=0, Bimg=img1t, M=M,
                   N=C0, rows_per_cloud=Npts, alpha=al_1, delta=de_1, P=P1, Q=Q1, tstride=0, slope=slope, C=da0, ldc=C0,
                   store_n0=0, Yp=y0, ldyp=C0, ealpha=al_0, edelta=de_0, emu=emu_0, er=inv_0, etstride=C0, rec2=r0b)
        db0, dg0, P0, Q0, dc = bwd_fin(r0b, R1, C0, tr_0, al_0, inv_0, emu_0, True, cm=cm_0, want_dc=True)
        # ---- first head layer (levels part + per-cloud global part) and the global-feature layer in its Gram form
        dW0 = torch.empty(C0, KL + CG, **f32)
        dbg, dgg = torch.empty(CG, **f32), torch.empty(CG, **f32)
        Pg, Qg, coef = torch.empty(CG, **f32), torch.empty(CG, **f32), torch.empty(B, CG, **f32)
        W0G, dW0G = W0[:, KL:], dW0[:, KL:]
        ldq = (KL + 4) & ~3
        Wq = torch.empty(CG, ldq, **f32)                                   # rows [Q o Wg | -P]
        call("fsg_pw_gf_prep_f32", _p(dc), _p(W0G), W0.stride(0), C0, _p(g), _p(dW0G), dW0.stride(0), None, _p(ysel), _p(al_g),
             _p(de_g), _p(mean_g), _p(inv_g), B, CG, M, int(tr_g), slope, _p(dbg), _p(dgg), _p(Pg), _p(Qg), _p(coef), _p(Wg),
             Wg.stride(0), KL, _p(Wq), ldq)
        # [M1 ; npvec] = [Q o Wg | -P]^T Wg with M1 = Wg^T diag(Q) Wg: one row contraction over the CG channel rows
        m1n = torch.empty(KL + 1, KL, **f32)
        pw_tn(3, m1n, KL, L1=Wq, ldl1=ldq, N1a=KL + 1, N1b=0, lpro=PRO_NONE, R=Wg, ldr=Wg.stride(0), N2=KL, rpro=PRO_NONE, slope=slope,
              M=CG, rows_per_cloud=0, rows_per_slice=64)
        M1, npvec = m1n[:KL], m1n[KL]
        ks_a, ks_b = C0 // 16, KL // 16
        pw_weight_image(M1, scale=-1.0, out=img_lv, ks0=ks_a, KS=ks_a + ks_b)
        Gs = torch.empty(KL + 1, KL, **f32)          # [X^T X ; column sums of X] (the all-ones column behind the left operand)
        G, s = Gs[:KL], Gs[KL]
        pw_tn(5, dW0, KL + CG, Gs, KL, defer=folds, ones=1, L1=da0, LY1=y0, L2=levels, ldl1=C0, ldl2=levels.stride(0), N1a=C0, N1b=KL, lpro=PRO_BNBWD,
              lalpha=al_0, ldelta=de_0, lP=P0, lQ=Q0, lts=C0, R=levels, ldr=levels.stride(0), N2=KL, rpro=PRO_NONE, slope=slope,
              M=M, rows_per_cloud=Npts, rows_per_slice=_tn_rps(M))
        pw_tn_reduce(folds)             # dW3, dW2, dW1, [dW0_levels ; G]
        dlv = torch.empty(M, KL, **f32)
        pw_rowgemm(PRO_BNBWD, PW_STORE | PW_BIAS, 5 if (_PW_WIDE and KL == 192) else 4, A1=da0, Y1=y0, A2=levels, lda1=C0, lda2=levels.stride(0), K1=C0, K2=KL,
                   Bimg=img_lv, M=M, N=KL, rows_per_cloud=Npts, alpha=al_0, delta=de_0, P=P0, Q=Q0, tstride=C0, slope=slope,
                   C=dlv, ldc=KL, store_n0=0, bias=npvec)
        sws = torch.empty(_lib.lib.fsg_pw_scatter_rows_workspace_bytes(B, CG) // 4, dtype=torch.int32, device=dev)
        call("fsg_pw_scatter_rows_f32", _p(coef), _p(arg), _p(Wg), Wg.stride(0), B, CG, KL, Npts, _p(dlv), KL, _p(sws))
        dWg = torch.empty(CG, KL, **f32)
        call("fsg_pw_gf_dw_f32", _p(coef), _p(arg), _p(levels), levels.stride(0), _p(s), _p(Wg), Wg.stride(0), _p(G), _p(Pg), _p(Qg),
             B, CG, KL, Npts, _p(dWg), KL)
        return (dlv, None, None, None, None, None, dWg, dgg, dbg, dW0, dg0, db0, dW1, dg1, db1, dW2, dg2, db2, dW3, db3)


_fused_head = _os.environ.get("FSG_FUSED_HEAD", "1") != "0"
_TN_RPS = int(_os.environ.get("FSG_TN_RPS", "0"))        # rows per slice of the weight-gradient contractions (0: by size; tuning knob)


def _tn_rps(M):
    """rows per slice of the head's weight-gradient contractions: 128 up to 16384 rows (512 workgroups per product, two resident per
    CU: config 2 0.956 vs 0.960 ms per step at 256, 1.005 at 64), 256 above (config 4: 2.07 vs 2.10 ms at 128; 32 x 2048 static:
    3.19 vs 3.24 -- twice the partial products to write and fold)"""
    return _TN_RPS if _TN_RPS > 0 else (128 if M <= 16384 else 256)


# the 64 x 192 tile for the (B N, 448) x (448, 192) input-gradient product of the first head layer: its BatchNorm-backward
# prologue + split is then done once per row instead of once per column tile (38.8 -> 33.1 us).  The 64 x 256 tile for the
# 256-wide products was measured SLOWER (24.9 vs 20.7 us, 25.5 vs 21.3 us: 120 KB of LDS = one workgroup per CU)
_PW_WIDE = _os.environ.get("FSG_PW_WIDE", "1") != "0"


def set_fused_head(flag):
    """switch the fused point-wise head of DGCNNSeg (csrc/pointwise.hip) on / off; off = vendor GEMMs + fsg_bn_act_* stages
    (the round-2 path, kept for shapes outside the fused kernels' envelope and as their cross-check); returns the old value"""
    global _fused_head
    old, _fused_head = _fused_head, bool(flag)
    return old


def seg_head_supported(levels, B, Npts, Wg, W0, W1, W2, W3, blocks=None):
    """the fused head needs: fp32 GPU rows, clouds of a multiple of 256 points, channel counts on the 32 / 64 grid, at most 32
    clouds per rank (the reference's experiment scripts train with 32); `blocks` = (global block, *segmentation blocks): the
    four BatchNorm-backed blocks must be what models/dgcnn.py:282-323 builds -- conv without bias, AFFINE BatchNorm1d, LeakyReLU
    (the fused node has no place for a conv bias in front of a BatchNorm and reads gamma / beta unconditionally)"""
    if not _fused_head:     # (bf16 operand mode keeps the fused head: its products are fp32-grade, above what the mode asks for)
        return False
    if blocks is not None:
        for blk in blocks[:4]:
            conv, bn, act = blk.layers[0], blk.layers[1], blk.layers[2]
            if (conv.bias is not None or not isinstance(bn, torch.nn.BatchNorm1d) or not bn.affine or
                    not isinstance(act, torch.nn.LeakyReLU)):
                return False
    KL = levels.shape[1]
    return (levels.is_cuda and levels.dtype == torch.float32 and levels.stride(1) == 1 and levels.stride(0) % 4 == 0 and
            Npts % 256 == 0 and levels.shape[0] == B * Npts and KL % 64 == 0 and Wg.shape[0] % 128 == 0 and
            W0.shape[0] % 64 == 0 and W0.shape[1] == KL + Wg.shape[0] and W1.shape[0] % 64 == 0 and W1.shape[1] == W0.shape[0] and
            W2.shape[0] % 64 == 0 and W2.shape[1] == W1.shape[0] and W3.shape[1] == W2.shape[0] and W3.shape[0] <= 8 and
            W2.shape[0] % 32 == 0 and Wg.shape[0] <= 4096 and B <= 32 and (8 if B <= 8 else 32) * W0.shape[0] <= 8192)


def seg_head(levels, B, Npts, global_block, seg_blocks):
    """DGCNNSeg's head on point-major `levels` (B * Npts, 192): global feature block (conv, BN, LeakyReLU) + the four
    segmentation blocks -> logits (B * Npts, classes).  See _SegHead."""
    gconv, gbn, gact = global_block.layers
    (c0, bn0, _), (c1, bn1, _), (c2, bn2, _) = (blk.layers for blk in seg_blocks[:3])
    c3 = seg_blocks[3].layers[0]
    bns = (gbn, bn0, bn1, bn2)
    steps = tuple(_bn_step(bn) for bn in bns)
    w = lambda conv: conv.weight.view(conv.out_channels, conv.in_channels)
    return _SegHead.apply(levels, B, Npts, float(gact.negative_slope), bns, steps, w(gconv), gbn.weight, gbn.bias, w(c0),
                          bn0.weight, bn0.bias, w(c1), bn1.weight, bn1.bias, w(c2), bn2.weight, bn2.bias, w(c3), c3.bias)


# ------------------------------------------------------------------ BatchNorm + LeakyReLU on (M, C) rows
class _BNAct(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, y, gamma, beta, rm, rv, training, momentum, eps, slope):
        y, gamma, beta = _f32c(y), _f32c(gamma), _f32c(beta)
        M, C = y.shape
        dev = y.device
        out = torch.empty_like(y)
        if training:
            mean = torch.empty(C, dtype=torch.float32, device=dev)
            invstd = torch.empty(C, dtype=torch.float32, device=dev)
            ws = torch.empty(_lib.lib.fsg_bn_act_workspace_bytes(M, C) // 4, dtype=torch.float32, device=dev)
        else:
            mean, invstd, ws = rm.detach().float().contiguous(), torch.rsqrt(rv.detach().float() + eps).contiguous(), None
        with torch.cuda.device(dev):
            _lib.call("fsg_bn_act_fwd_f32", _p(y), _p(gamma), _p(beta), _p(rm if training else None),
                      _p(rv if training else None), M, C, int(training), momentum, eps, slope, _p(out), _p(mean),
                      _p(invstd), _p(ws), _stream())
        ctx.save_for_backward(y, gamma, beta, mean, invstd)
        ctx.meta = (M, C, bool(training), slope)
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        y, gamma, beta, mean, invstd = ctx.saved_tensors
        M, C, training, slope = ctx.meta
        g = _f32c(g)
        gy = torch.empty_like(y)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
        ws = torch.empty(_lib.lib.fsg_bn_act_workspace_bytes(M, C) // 4, dtype=torch.float32, device=y.device)
        with torch.cuda.device(y.device):
            _lib.call("fsg_bn_act_bwd_f32", _p(g), _p(y), _p(gamma), _p(beta), _p(mean), _p(invstd), M, C, int(training),
                      slope, _p(gy), _p(dgamma), _p(dbeta), _p(ws), _stream())
        return gy, dgamma, dbeta, None, None, None, None, None, None


class _BNActMax(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, y, gamma, beta, rm, rv, training, momentum, eps, slope):
        y, gamma, beta = _f32c(y), _f32c(gamma), _f32c(beta)
        B, N, C = y.shape
        dev = y.device
        out = torch.empty(B, C, dtype=torch.float32, device=dev)
        ysel = torch.empty(B, C, dtype=torch.float32, device=dev)
        arg = torch.empty(B, C, dtype=torch.int32, device=dev)
        if training:
            mean = torch.empty(C, dtype=torch.float32, device=dev)
            invstd = torch.empty(C, dtype=torch.float32, device=dev)
        else:
            mean, invstd = rm.detach().float().contiguous(), torch.rsqrt(rv.detach().float() + eps).contiguous()
        ws = torch.empty(_lib.lib.fsg_bn_act_max_workspace_bytes(B, N, C) // 4, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.call("fsg_bn_act_max_fwd_f32", _p(y), _p(gamma), _p(beta), _p(rm if training else None),
                      _p(rv if training else None), B, N, C, int(training), momentum, eps, slope, _p(out), _p(ysel),
                      _p(arg), _p(mean), _p(invstd), _p(ws), _stream())
        ctx.save_for_backward(y, gamma, beta, mean, invstd, ysel, arg)
        ctx.meta = (B, N, C, bool(training), slope)
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        y, gamma, beta, mean, invstd, ysel, arg = ctx.saved_tensors
        B, N, C, training, slope = ctx.meta
        g = _f32c(g)
        gy = torch.empty_like(y)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
        with torch.cuda.device(y.device):
            _lib.call("fsg_bn_act_max_bwd_f32", _p(g), _p(y), _p(ysel), _p(arg), _p(gamma), _p(beta), _p(mean), _p(invstd),
                      B, N, C, int(training), slope, _p(gy), _p(dgamma), _p(dbeta), _stream())
        return gy, dgamma, dbeta, None, None, None, None, None, None


def bn_act_max(y, bn, slope):
    """max over dim 1 of LeakyReLU(slope)(BatchNorm(y)) for y (B, N, C) -> (B, C), without the (B,N,C) activation."""
    _need_gpu(y)
    training, momentum = _bn_step(bn)
    track = training and bn.track_running_stats
    return _BNActMax.apply(y, bn.weight, bn.bias, bn.running_mean if (track or not training) else None,
                           bn.running_var if (track or not training) else None, training, momentum, float(bn.eps),
                           float(slope))


def bn_act_supported(y, bn):
    return y.is_cuda and y.dim() == 2 and y.shape[1] % 64 == 0 and bn.affine


def bn_act(y, bn, slope):
    """LeakyReLU(slope)(BatchNorm1d(y)) for point-major rows y (M, C): one fused HIP stage (slope 1: BN only)."""
    _need_gpu(y)
    training, momentum = _bn_step(bn)
    track = training and bn.track_running_stats
    return _BNAct.apply(y, bn.weight, bn.bias, bn.running_mean if (track or not training) else None,
                        bn.running_var if (track or not training) else None, training, momentum, float(bn.eps),
                        float(slope))


# ------------------------------------------------------------------ BatchNorm (+ residual) (+ ReLU) over packed rows
BN_ROWS_WIDTHS = (32, 64, 128, 256, 512)


class _BNRows(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, x, res, gamma, beta, rm, rv, training, momentum, eps, relu):
        M, C = x.shape
        dev = x.device
        out = torch.empty_like(x)
        if training:
            mean = torch.empty(C, dtype=torch.float32, device=dev)
            rstd = torch.empty(C, dtype=torch.float32, device=dev)
            ws = torch.empty(_lib.lib.fsg_bn_rows_workspace_bytes(M, C) // 8, dtype=torch.float64, device=dev)
        else:
            mean, rstd, ws = rm.detach().float().contiguous(), torch.rsqrt(rv.detach().float() + eps).contiguous(), None
        with torch.cuda.device(dev):
            _lib.call("fsg_bn_rows_fwd_f32", _p(x), _p(res), _p(gamma), _p(beta), _p(rm if training else None),
                      _p(rv if training else None), M, C, int(training), momentum, eps, int(relu), _p(out), _p(mean),
                      _p(rstd), _p(ws), _stream())
        ctx.save_for_backward(x, out, gamma, mean, rstd)
        ctx.meta = (M, C, bool(training), bool(relu), res is not None)
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        x, out, gamma, mean, rstd = ctx.saved_tensors
        M, C, training, relu, has_res = ctx.meta
        g = g if (g.dtype == torch.float32 and g.is_contiguous()) else g.float().contiguous()
        gx = torch.empty_like(x)
        gres = torch.empty_like(x) if has_res else None
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        ws = torch.empty(_lib.lib.fsg_bn_rows_workspace_bytes(M, C) // 8, dtype=torch.float64, device=x.device)
        with torch.cuda.device(x.device):
            _lib.call("fsg_bn_rows_bwd_f32", _p(g), _p(x), _p(out), _p(gamma), _p(mean), _p(rstd), M, C, int(training),
                      int(relu), _p(gx), _p(gres), _p(dgamma), _p(dbeta), _p(ws), _stream())
        return gx, gres, dgamma, dbeta, None, None, None, None, None, None


def bn_rows_supported(x, bn):
    return (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[1] in BN_ROWS_WIDTHS and x.shape[0] > 0
            and bn.affine and (bn.training or bn.running_mean is not None))


def bn_rows(x, bn, relu=True, residual=None):
    """[relu]( BatchNorm1d(x) [+ residual] ) for packed point rows x (M, C): one fused HIP stage (3 launches forward,
    3 backward) instead of ATen's 4-5 + 3-5.  Same running-statistics semantics as torch.nn.BatchNorm1d."""
    _need_gpu(x)
    training, momentum = _bn_step(bn)
    track = training and bn.track_running_stats
    x = x if x.is_contiguous() else x.contiguous()
    if residual is not None and not residual.is_contiguous():
        residual = residual.contiguous()
    return _BNRows.apply(x, residual, bn.weight, bn.bias, bn.running_mean if (track or not training) else None,
                         bn.running_var if (track or not training) else None, training, momentum, float(bn.eps), relu)


# ------------------------------------------------------------------ Chamfer (losses/chamfer_loss.py:19)
class _ChamferNN(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, x, y):
        xc, yc = _f32c(x), _f32c(y)
        B, N, _ = xc.shape
        M = yc.shape[1]
        d = torch.empty(B, N, dtype=torch.float32, device=x.device)
        a = torch.empty(B, N, dtype=torch.int32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.call("fsg_chamfer_nn_f32", _p(xc), _p(yc), B, N, M, _p(d), _p(a), _stream())
        ctx.save_for_backward(xc, yc, a)
        ctx.mark_non_differentiable(a)
        return d, a

    @staticmethod
    @_amp_bwd
    def backward(ctx, gd, _ga):
        xc, yc, a = ctx.saved_tensors
        B, N, _ = xc.shape
        M = yc.shape[1]
        gx, gy = torch.zeros_like(xc), torch.zeros_like(yc)
        ws = None
        if deterministic():
            ws = torch.empty(_lib.lib.fsg_chamfer_nn_bwd_workspace_bytes(B, N, M) // 4, dtype=torch.int32, device=xc.device)
        with torch.cuda.device(xc.device):
            _lib.call("fsg_chamfer_nn_bwd_f32", _p(xc), _p(yc), _p(a), _p(_f32c(gd)), B, N, M, _p(gx), _p(gy), _p(ws),
                      _stream())
        return gx, gy


def chamfer_nn(x, y):
    """x (B,N,3), y (B,M,3) -> (min squared distance (B,N), argmin (B,N) int32); differentiable in x, y."""
    _need_gpu(x, y)
    if x.dim() != 3 or y.dim() != 3 or x.shape[2] != 3 or y.shape[2] != 3 or x.shape[0] != y.shape[0]:
        raise ValueError(f"expected (B,N,3) and (B,M,3), got {tuple(x.shape)} and {tuple(y.shape)}")
    return _ChamferNN.apply(x, y)


# ------------------------------------------------------------------ segmentation loss (losses/nnu_loss.py:6-19)
class _NNULoss(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, logits, target, class_weights, w_ce, w_dice, smooth):
        B, C, N = logits.shape
        dev = logits.device
        vals = torch.empty(4, dtype=torch.float32, device=dev)
        need_grad = ctx.needs_input_grad[0]
        grad = torch.empty_like(logits) if need_grad else None      # keeps the (possibly point-major) strides
        if grad is not None and grad.stride() != logits.stride():
            grad = torch.empty_strided(logits.shape, logits.stride(), dtype=torch.float32, device=dev)
        ws = torch.empty(_lib.lib.fsg_nnu_loss_workspace_bytes(C) // 8, dtype=torch.float64, device=dev)
        gs = grad.stride() if grad is not None else (0, 0, 0)
        with torch.cuda.device(dev):
            _lib.call("fsg_nnu_loss_f32", _p(logits), logits.stride(0), logits.stride(1), logits.stride(2), _p(target),
                      _p(class_weights) if class_weights is not None else None, B, C, N, float(w_ce), float(w_dice),
                      float(smooth), _p(vals), _p(grad) if grad is not None else None, gs[0], gs[1], gs[2], _p(ws),
                      _stream())
        ctx.grad = grad
        total, ce, gdl = vals[0], vals[1], vals[2]
        ctx.mark_non_differentiable(ce, gdl)
        ctx.set_materialize_grads(False)      # no zero tensors for the two logging outputs
        return total, ce, gdl

    @staticmethod
    @_amp_bwd
    def backward(ctx, g, _gce, _ggdl):
        grad, ctx.grad = ctx.grad, None
        if g is None:
            return None, None, None, None, None, None
        return grad * g, None, None, None, None, None


def nnu_loss(logits, target, class_weights=None, w_ce=1.0, w_dice=1.0, smooth=1.0):
    """CrossEntropyLoss(class_weights) + generalised Dice (softmax, batch_dice) of logits (B,C,N) against labels (B,N).

    Returns 0-d tensors (w_ce*ce + w_dice*gdl, ce, gdl); the first is differentiable in `logits` (any strides), the
    other two are for logging.  The gradient is produced by the same two launches that evaluate the loss."""
    _need_gpu(logits, target)
    if logits.dim() != 3 or target.shape != (logits.shape[0], logits.shape[2]):
        raise ValueError(f"expected logits (B,C,N) and labels (B,N), got {tuple(logits.shape)} and {tuple(target.shape)}")
    if not 2 <= logits.shape[1] <= 32:
        raise ValueError(f"number of classes must be in [2, 32], got {logits.shape[1]}")
    if logits.dtype != torch.float32:
        logits = logits.float()
    target = target.to(torch.int64).contiguous()
    if class_weights is not None:
        if class_weights.numel() != logits.shape[1]:
            raise ValueError("class_weights must have one entry per class")
        class_weights = _f32c(class_weights.to(logits.device))
    return _NNULoss.apply(logits, target, class_weights, w_ce, w_dice, smooth)


# ------------------------------------------------------------------ packed clouds (pointops.py)
def knn_segment(nsample, xyz, new_xyz, offset, new_offset):
    """-> idx (m,nsample) int32, dist2 (m,nsample) squared distances."""
    _need_gpu(xyz, new_xyz, offset, new_offset)
    xyz, new_xyz = _f32c(xyz), _f32c(new_xyz)
    offset, new_offset = offset.to(torch.int32).contiguous(), new_offset.to(torch.int32).contiguous()
    m = new_xyz.shape[0]
    idx = torch.empty(m, nsample, dtype=torch.int32, device=xyz.device)
    d2 = torch.empty(m, nsample, dtype=torch.float32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        _lib.call("fsg_knn_segment_f32", _p(xyz), _p(new_xyz), _p(offset), _p(new_offset), offset.shape[0],
                  xyz.shape[0], m, nsample, _p(idx), _p(d2), _stream())
    return idx, d2


def fps(xyz, offset, new_offset, m):
    """m = total number of samples (new_offset[-1], known on the host) -> idx (m) int32."""
    _need_gpu(xyz, offset, new_offset)
    xyz = _f32c(xyz)
    offset, new_offset = offset.to(torch.int32).contiguous(), new_offset.to(torch.int32).contiguous()
    n = xyz.shape[0]
    idx = torch.empty(m, dtype=torch.int32, device=xyz.device)     # every entry is written (empty segments: zeros, by the kernel)
    tmp = torch.empty(n, dtype=torch.float32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        _lib.call("fsg_fps_f32", _p(xyz), _p(offset), _p(new_offset), offset.shape[0], n, _p(tmp), _p(idx), _stream())
    return idx


# ------------------------------------------------------------------ zeroed scratch of a backward pass, one fill
class ZeroArena:
    """Backward kernels that accumulate with atomics (grouping, interpolation, the attention layer's dk / dv / dp) need zeroed
    buffers: 30 fill launches per PointTransformer step.  Inside `zero_arena()` the forward of each such Function RESERVES its
    size; the first backward that asks gets ONE torch.zeros over all reservations and every Function its slice of it."""

    def __init__(self):
        self.total, self.buf, self.taken = 0, None, set()

    def reserve(self, numel):
        if self.buf is not None:          # a forward running after a backward of the same arena has started (checkpointing)
            return None
        off = self.total
        self.total += (int(numel) + 63) // 64 * 64      # 256-byte aligned slices
        return off

    def take(self, off, numel, device):
        if off is None or off in self.taken:            # second backward through the same graph: its own fresh zeros
            return torch.zeros(numel, dtype=torch.float32, device=device)
        if self.buf is None:
            self.buf = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.taken.add(off)
        return self.buf[off:off + numel]


_zero_arena = None


@_contextlib.contextmanager
def zero_arena():
    global _zero_arena
    outer, _zero_arena = _zero_arena, ZeroArena()
    try:
        yield _zero_arena
    finally:
        _zero_arena = outer


def _reserve_zeros(numel):
    """forward side: -> token for _take_zeros (None outside zero_arena())"""
    return (_zero_arena, _zero_arena.reserve(numel)) if _zero_arena is not None else None


def _take_zeros(token, numel, device):
    """backward side: a zeroed fp32 buffer of `numel` elements"""
    if token is None:
        return torch.zeros(numel, dtype=torch.float32, device=device)
    return token[0].take(token[1], numel, device)


class _GroupGather(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, feat, idx):
        f = _f32c(feat)
        n, c = f.shape
        m, ns = idx.shape
        out = torch.empty(m, ns, c, dtype=torch.float32, device=f.device)
        with torch.cuda.device(f.device):
            _lib.call("fsg_group_gather_fwd_f32", _p(f), _p(idx), _p(out), n, c, m, ns, _stream())
        ctx.save_for_backward(idx)
        ctx.nc = (n, c)
        ctx.zeros = _reserve_zeros(n * c) if ctx.needs_input_grad[0] else None
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        n, c = ctx.nc
        m, ns = idx.shape
        g = _f32c(g)
        gf = _take_zeros(ctx.zeros, n * c, g.device).view(n, c)
        with torch.cuda.device(g.device):
            _lib.call("fsg_group_gather_bwd_f32", _p(g), _p(idx), _p(gf), n, c, m, ns, _stream())
        return gf, None


def group_gather(feat, idx):
    """feat (n,c), idx (m,ns) int32 -> (m,ns,c) = feat[idx]."""
    _need_gpu(feat, idx)
    return _GroupGather.apply(feat, idx.to(torch.int32).contiguous())


class _GroupXyzFeat(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, xyz, new_xyz, feat, idx):
        xyz, new_xyz, f = _f32c(xyz), _f32c(new_xyz), _f32c(feat)
        n, c = f.shape
        m, ns = idx.shape
        out = torch.empty(m, ns, 3 + c, dtype=torch.float32, device=f.device)
        with torch.cuda.device(f.device):
            _lib.call("fsg_group_xyz_feat_fwd_f32", _p(xyz), _p(new_xyz), _p(f), _p(idx), _p(out), n, c, m, ns, _stream())
        ctx.save_for_backward(idx)
        ctx.nc = (n, c)
        ctx.zeros = _reserve_zeros(n * c) if ctx.needs_input_grad[2] else None
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        n, c = ctx.nc
        m, ns = idx.shape
        g = _f32c(g)
        gf = _take_zeros(ctx.zeros, n * c, g.device).view(n, c)
        with torch.cuda.device(g.device):
            _lib.call("fsg_group_xyz_feat_bwd_f32", _p(g), _p(idx), _p(gf), n, c, m, ns, _stream())
        return None, None, gf, None


def group_xyz_feat(xyz, new_xyz, feat, idx):
    """pointops.queryandgroup(use_xyz=True) behind the kNN query: (m, ns, 3 + c) = [xyz[idx] - new_xyz | feat[idx]], one launch
    each way; gradient for `feat` only (the caller checks that the coordinates need none)."""
    _need_gpu(xyz, new_xyz, feat, idx)
    return _GroupXyzFeat.apply(xyz, new_xyz, feat, idx.to(torch.int32).contiguous())


class _RowsMax(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, x):
        x = _f32c(x)
        m, ns, c = x.shape
        out = torch.empty(m, c, dtype=torch.float32, device=x.device)
        arg = torch.empty(m, c, dtype=torch.int32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.call("fsg_rows_max_fwd_f32", _p(x), _p(out), _p(arg), m, ns, c, _stream())
        ctx.save_for_backward(arg)
        ctx.ns = ns
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        m, c = arg.shape
        g = _f32c(g)
        gx = torch.empty(m, ctx.ns, c, dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            _lib.call("fsg_rows_max_bwd_f32", _p(g), _p(arg), _p(gx), m, ctx.ns, c, _stream())
        return gx


def rows_max(x):
    """x (m, ns, c) -> (m, c): max over the ns neighbour rows (nn.MaxPool1d(ns) of TransitionDown), one launch each way"""
    _need_gpu(x)
    return _RowsMax.apply(x)


class _Interp(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, feat, idx, d2):
        f = _f32c(feat)
        n, c = f.shape
        m, k = idx.shape
        out = torch.empty(m, c, dtype=torch.float32, device=f.device)
        with torch.cuda.device(f.device):
            _lib.call("fsg_interp_fwd_f32", _p(f), _p(idx), _p(d2), _p(out), n, c, m, k, _stream())
        ctx.save_for_backward(idx, d2)
        ctx.nc = (n, c)
        ctx.zeros = _reserve_zeros(n * c)
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        idx, d2 = ctx.saved_tensors
        n, c = ctx.nc
        m, k = idx.shape
        g = _f32c(g)
        gf = _take_zeros(ctx.zeros, n * c, g.device).view(n, c)
        with torch.cuda.device(g.device):
            _lib.call("fsg_interp_bwd_f32", _p(g), _p(idx), _p(d2), _p(gf), n, c, m, k, _stream())
        return gf, None, None


def interpolate(feat, idx, dist2):
    """pointops.interpolation's arithmetic as one launch: feat (n, c), idx / dist2 (m, k <= 8) of the k nearest coarse points
    (squared distances, as fsg_knn_segment_f32 returns them) -> (m, c) inverse-distance weighted mean of feat[idx]."""
    _need_gpu(feat, idx, dist2)
    if idx.shape != dist2.shape or idx.shape[1] > 8:
        raise ValueError(f"interpolate: idx {tuple(idx.shape)} / dist2 {tuple(dist2.shape)}: equal shapes, k <= 8")
    return _Interp.apply(feat, idx.to(torch.int32).contiguous(), _f32c(dist2))


class _VecAttn(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, v, pos, w, idx):
        v, pos, w = _f32c(v), _f32c(pos), _f32c(w)
        n, c = v.shape
        ns, cw = w.shape[1], w.shape[2]
        out = torch.empty(n, c, dtype=torch.float32, device=v.device)
        with torch.cuda.device(v.device):
            _lib.call("fsg_vec_attn_fwd_f32", _p(v), _p(pos), _p(w), _p(idx), _p(out), n, ns, c, cw, _stream())
        ctx.save_for_backward(v, pos, w, idx)
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        v, pos, w, idx = ctx.saved_tensors
        n, c = v.shape
        ns, cw = w.shape[1], w.shape[2]
        g = _f32c(g)
        gv = torch.zeros_like(v)
        gpos = torch.empty_like(pos)
        gw = torch.empty_like(w)
        with torch.cuda.device(v.device):
            _lib.call("fsg_vec_attn_bwd_f32", _p(v), _p(pos), _p(w), _p(idx), _p(g), _p(gv), _p(gpos), _p(gw), n, ns, c,
                      cw, _stream())
        return gv, gpos, gw, None


def vec_attn(v, pos, w, idx):
    """out[i,ch] = sum_j (v[idx[i,j],ch] + pos[i,j,ch]) * w[i,j,ch % cw]  (seg_model.py:50-52)."""
    _need_gpu(v, pos, w, idx)
    if pos.shape != (v.shape[0], idx.shape[1], v.shape[1]) or w.shape[:2] != pos.shape[:2] or v.shape[1] % w.shape[2]:
        raise ValueError(f"vec_attn: inconsistent shapes v{tuple(v.shape)} pos{tuple(pos.shape)} w{tuple(w.shape)}")
    return _VecAttn.apply(v, pos, w, idx.to(torch.int32).contiguous())


# ------------------------------------------------------------------ [Wq ; Wk ; Wv] of every PointTransformerLayer, one launch
class _PackQKV(torch.autograd.Function):
    """inputs: (Wq, Wk, Wv, bq, bk, bv) per layer -> outputs: (W_qkv (3c, c), b_qkv (3c)) per layer, all views of ONE buffer that
    one torch.cat fills (ATen copies up to 128 pieces per launch).  The backward hands views of the incoming gradients on: no
    launch, and it does not read them -- which is what lets the products' weight gradients be reduced late (gemm_small defer)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, *ts):
        L = len(ts) // 6
        ctx.cs = [ts[6 * i].shape for i in range(L)]
        arena = torch.cat([t.reshape(-1) for t in ts])
        outs, off = [], 0
        for i in range(L):
            co, ci = ctx.cs[i]
            outs.append(arena[off:off + 3 * co * ci].view(3 * co, ci))
            off += 3 * co * ci
            outs.append(arena[off:off + 3 * co])
            off += 3 * co
        return tuple(outs)

    @staticmethod
    @_amp_bwd
    def backward(ctx, *gs):
        res = []
        for i, (co, ci) in enumerate(ctx.cs):
            gw, gb = gs[2 * i], gs[2 * i + 1]
            res += [None] * 3 if gw is None else [gw[0:co], gw[co:2 * co], gw[2 * co:3 * co]]
            res += [None] * 3 if gb is None else [gb[0:co], gb[co:2 * co], gb[2 * co:3 * co]]
        return tuple(res)


_qkv_packs = None      # {id(layer): (W_qkv, b_qkv)} inside qkv_pack()


@_contextlib.contextmanager
def qkv_pack(layers):
    """packs the q / k / v weights and biases of `layers` (modules with linear_q / linear_k / linear_v of equal output width) for
    the duration of one forward; PointTransformerLayer.forward picks its pair up through packed_qkv(layer)"""
    global _qkv_packs
    outer = _qkv_packs
    ok = [m for m in layers if m.linear_q.weight.is_cuda and m.linear_q.bias is not None and
          m.linear_q.weight.shape == m.linear_k.weight.shape == m.linear_v.weight.shape]
    ts = [t for m in ok for t in (m.linear_q.weight, m.linear_k.weight, m.linear_v.weight, m.linear_q.bias, m.linear_k.bias,
                                  m.linear_v.bias)]
    packs = {}
    if ts:
        outs = _PackQKV.apply(*ts)
        for i, m in enumerate(ok):
            w, b = outs[2 * i], outs[2 * i + 1]
            w._fsg_grad_unread = (m.linear_q.weight, m.linear_k.weight, m.linear_v.weight)
            b._fsg_grad_unread = (m.linear_q.bias, m.linear_k.bias, m.linear_v.bias)
            packs[id(m)] = (w, b)
    _qkv_packs = packs
    try:
        yield
    finally:
        _qkv_packs = outer


def packed_qkv(layer):
    return _qkv_packs.get(id(layer)) if _qkv_packs is not None else None


# ------------------------------------------------------------------ fused PointTransformerLayer body (seg_model.py:38-53)
_PT_PARAM_NAMES = ("lp1_w", "lp1_b", "bnp_g", "bnp_b", "lp2_w", "lp2_b", "bn1_g", "bn1_b", "lw1_w", "lw1_b", "bn2_g", "bn2_b",
                   "lw2_w", "lw2_b")


class _PTAttn(torch.autograd.Function):
    """out = fused(grouping, linear_p, linear_w, softmax, aggregate)(p, idx, qkv; 14 parameters).  Only the (n,ns,c/8)
    tensors u1 and softmax weights are kept for the backward; gradients: qkv and the 14 parameters."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, p, idx, qkv, bns, *params):
        n, c3 = qkv.shape
        c, ns = c3 // 3, idx.shape[1]
        cs = c // 8
        dev = qkv.device
        bnp, bn1, bn2 = bns
        training = bnp.training or bnp.running_mean is None
        moms = [0.0, 0.0, 0.0]
        if training:
            for i, bn in enumerate(bns):
                if bn.track_running_stats and bn.num_batches_tracked is not None:
                    moms[i] = float(bump_bn_counter(bn))
        prm = _lib.PTLayerParams()
        for name, t in zip(_PT_PARAM_NAMES, params):
            setattr(prm, name, t.data_ptr())
        for tag, bn in (("bnp", bnp), ("bn1", bn1), ("bn2", bn2)):
            track = training and bn.track_running_stats and bn.running_mean is not None
            setattr(prm, tag + "_rm", bn.running_mean.data_ptr() if track else None)
            setattr(prm, tag + "_rv", bn.running_var.data_ptr() if track else None)
        prm.eps_p, prm.eps_1, prm.eps_2 = bnp.eps, bn1.eps, bn2.eps
        prm.mom_p, prm.mom_1, prm.mom_2 = moms
        if training:
            stats = torch.empty(2 * (3 + c + cs), dtype=torch.float32, device=dev)
        else:
            stats = torch.cat([t for bn in bns for t in (bn.running_mean.float(), torch.rsqrt(bn.running_var.float() + bn.eps))])
        out = torch.empty(n, c, dtype=torch.float32, device=dev)
        u1 = torch.empty(n, ns, cs, dtype=torch.float32, device=dev)
        sm = torch.empty(n, ns, cs, dtype=torch.float32, device=dev)
        ws = torch.empty(_lib.lib.fsg_pt_attn_workspace_bytes(n, ns, c) // 8 + 1, dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            _lib.call("fsg_pt_attn_fwd_f32", _p(p), _p(idx), _p(qkv), ctypes.c_void_p(qkv.data_ptr() + 4 * c),
                      ctypes.c_void_p(qkv.data_ptr() + 8 * c), c3, ctypes.byref(prm), n, ns, c, int(training), _p(out),
                      _p(stats), _p(u1), _p(sm), _p(ws), _stream())
        ctx.save_for_backward(p, idx, qkv, stats, u1, sm, *params)
        ctx.meta = (n, ns, c, training, (bnp.eps, bn1.eps, bn2.eps))
        ctx.zeros = _reserve_zeros(n * 3 * c + (n * 3 if ctx.needs_input_grad[0] else 0))
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        p, idx, qkv, stats, u1, sm, *params = ctx.saved_tensors
        n, ns, c, training, eps = ctx.meta
        dev = qkv.device
        prm = _lib.PTLayerParams()
        for name, t in zip(_PT_PARAM_NAMES, params):
            setattr(prm, name, t.data_ptr())
        prm.eps_p, prm.eps_1, prm.eps_2 = eps
        sizes = [t.numel() for t in params]
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        grads, gstruct, off = [], _lib.PTLayerGrads(), 0
        for name, t, sz in zip(_PT_PARAM_NAMES, params, sizes):
            gt = flat[off:off + sz].view(t.shape)
            setattr(gstruct, name, gt.data_ptr())
            grads.append(gt)
            off += sz
        # dk / dv / dp are accumulated with atomics: ONE zero fill for both buffers (two fill launches per layer and step before)
        need_dp = ctx.needs_input_grad[0]
        zbuf = _take_zeros(ctx.zeros, n * 3 * c + (n * 3 if need_dp else 0), dev)
        dqkv = zbuf[:n * 3 * c].view(n, 3 * c)
        dp = zbuf[n * 3 * c:].view(n, 3) if need_dp else None
        ws = torch.empty(_lib.lib.fsg_pt_attn_workspace_bytes(n, ns, c) // 8 + 1, dtype=torch.float64, device=dev)
        g = _f32c(g)
        with torch.cuda.device(dev):
            _lib.call("fsg_pt_attn_bwd_f32", _p(p), _p(idx), _p(qkv), ctypes.c_void_p(qkv.data_ptr() + 4 * c),
                      ctypes.c_void_p(qkv.data_ptr() + 8 * c), 3 * c, ctypes.byref(prm), n, ns, c, int(training), _p(g),
                      _p(stats), _p(u1), _p(sm), _p(dqkv), ctypes.c_void_p(dqkv.data_ptr() + 4 * c),
                      ctypes.c_void_p(dqkv.data_ptr() + 8 * c), 3 * c, _p(dp), ctypes.byref(gstruct), _p(ws), _stream())
        return (dp, None, dqkv, None, *grads)


PT_ATTN_PLANES = (32, 64, 128, 256, 512)


def pt_attn(p, idx, qkv, linear_p, linear_w):
    """Fused body of PointTransformerLayer.  p (n,3), idx (n,ns) int32, qkv (n,3c) = [q | k | v] rows;
    linear_p = Sequential(Linear(3,3), BatchNorm1d(3), ReLU, Linear(3,c)), linear_w = Sequential(BatchNorm1d(c), ReLU,
    Linear(c,c/8), BatchNorm1d(c/8), ReLU, Linear(c/8,c/8)) -- the reference's module layout (seg_model.py:27-33)."""
    _need_gpu(p, idx, qkv)
    c = qkv.shape[1] // 3
    if c not in PT_ATTN_PLANES or not 1 <= idx.shape[1] <= 16:
        raise ValueError(f"fused PointTransformer layer supports planes {PT_ATTN_PLANES} and nsample <= 16, got {c}, {idx.shape[1]}")
    lp1, bnp, _, lp2 = linear_p
    bn1, _, lw1, bn2, _, lw2 = linear_w
    params = (lp1.weight, lp1.bias, bnp.weight, bnp.bias, lp2.weight, lp2.bias, bn1.weight, bn1.bias, lw1.weight, lw1.bias,
              bn2.weight, bn2.bias, lw2.weight, lw2.bias)
    params = tuple(t if t.is_contiguous() else t.contiguous() for t in params)
    p = p if p.dtype == torch.float32 and p.is_contiguous() else p.float().contiguous()
    return _PTAttn.apply(p, idx.to(torch.int32).contiguous(), qkv.contiguous(), (bnp, bn1, bn2), *params)
