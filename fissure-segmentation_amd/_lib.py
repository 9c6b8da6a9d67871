"""ctypes binding of libfsg_hip.so (C ABI: include/fsg_hip.h).  No fallback: a missing library is an
ImportError, a failing call is a RuntimeError carrying fsg_last_error()."""
import ctypes
import os

import torch  # noqa: F401  -- must come first: brings torch's libamdhip64 into the process, which the
#                              library's DT_NEEDED libamdhip64.so.7 then resolves to (one HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
# (FSG_HIP_LIB: another BUILD of the same library -- tools/ use it to A/B compile-time variants on one box; still no fallback)
LIB_PATH = os.environ.get("FSG_HIP_LIB") or os.path.join(_HERE, "libfsg_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      f"or `make -C {os.path.join(_HERE, 'csrc')}`; there is no CPU fallback.")

lib = ctypes.CDLL(LIB_PATH)

_P, _I, _L, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
SIGNATURES = {
    "fsg_version": ([], _I),
    "fsg_last_error": ([], ctypes.c_char_p),
    "fsg_knn_dense_f32": ([_P, _I, _I, _L, _L, _I, _I, _I, _P, _P, _P, _P], _I),
    "fsg_knn_dense_workspace_bytes": ([_I, _I, _I], ctypes.c_size_t),
    "fsg_knn_dense_ws_f32": ([_P, _I, _I, _L, _L, _I, _I, _I, _P, _P, _P, ctypes.c_size_t, _P], _I),
    "fsg_knn_dense_prepared_f32": ([_P, _I, _I, _I, _I, _I, _P, _P, _P, ctypes.c_size_t, _P], _I),
    "fsg_knn_dense_ws_pq_f32": ([_P, _I, _I, _L, _L, _I, _I, _I, _P, _P, _P, ctypes.c_size_t, _P, _I, _P, _P], _I),
    "fsg_edgeconv_apply_f32": ([_P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P, ctypes.c_size_t, _P], _I),
    "fsg_edgeconv_apply_pq_f32": ([_P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P, ctypes.c_size_t, _P, _I, _P, _P], _I),
    "fsg_edge_gather_fwd_f32": ([_P, _P, _P, _I, _I, _I, _I, _P], _I),
    "fsg_edge_gather_bwd_f32": ([_P, _P, _P, _I, _I, _I, _I, _P], _I),
    "fsg_knn_gather_fused_f32": ([_P, _I, _I, _I, _I, _I, _P, _P, _P, _P], _I),
    "fsg_knn_gather_fused_ws_f32": ([_P, _I, _I, _I, _I, _I, _P, _P, _P, ctypes.c_size_t, _P], _I),
    "fsg_edge_gather_fwd_bf16": ([_P, _P, _P, _I, _I, _I, _I, _P], _I),
    "fsg_edge_gather_bwd_bf16": ([_P, _P, _P, _I, _I, _I, _I, _P], _I),
    "fsg_graph_reverse_csr_workspace_bytes": ([_I, _I, _I], ctypes.c_size_t),
    "fsg_graph_reverse_csr": ([_P, _I, _I, _I, _P, _P, _P, _P], _I),
    "fsg_edge_weights_fwd_f32": ([_P, _I, _I, _P, _P], _I),
    "fsg_edge_weights_bwd_f32": ([_P, _I, _I, _P, _P], _I),
    "fsg_edge_weights_many_f32": ([_P, _I, _P], _I),
    "fsg_edgeconv1_workspace_bytes": ([_I, _I, _I], ctypes.c_size_t),
    "fsg_edgeconv1_fwd_f32": ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P],
                              _I),
    "fsg_edgeconv1_bwd_f32": ([_P, _P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _P,
                               _P, _P, _P, _P], _I),
    "fsg_edgeconv2_workspace_bytes": ([_I, _I, _I, _I], ctypes.c_size_t),
    "fsg_edgeconv2_bwd_workspace_bytes": ([_I, _I, _I, _I], ctypes.c_size_t),
    "fsg_edgeconv2_fwd_f32": ([_P] * 11 + [_I] * 5 + [_F] * 5 + [_P] * 12, _I),
    "fsg_edgeconv2_bwd_f32": ([_P, _P, _L, _P, _L] + [_P] * 16 + [_I] * 5 + [_F] + [_P] * 8, _I),
    "fsg_edgeconv2_fwd_bf16": ([_P] * 11 + [_I] * 5 + [_F] * 5 + [_P] * 12, _I),
    "fsg_edgeconv2_bwd_bf16": ([_P, _P, _L, _P, _L] + [_P] * 16 + [_I] * 5 + [_F] + [_P] * 8, _I),
    "fsg_bn_act_workspace_bytes": ([ctypes.c_long, _I], ctypes.c_size_t),
    "fsg_bn_act_fwd_f32": ([_P, _P, _P, _P, _P, ctypes.c_long, _I, _I, _F, _F, _F, _P, _P, _P, _P, _P], _I),
    "fsg_bn_act_bwd_f32": ([_P, _P, _P, _P, _P, _P, ctypes.c_long, _I, _I, _F, _P, _P, _P, _P, _P], _I),
    "fsg_bn_act_max_workspace_bytes": ([_I, _I, _I], ctypes.c_size_t),
    "fsg_bn_act_max_fwd_f32": ([_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _F, _P, _P, _P, _P, _P, _P, _P], _I),
    "fsg_bn_act_max_bwd_f32": ([_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _P], _I),
    "fsg_chamfer_nn_f32": ([_P, _P, _I, _I, _I, _P, _P, _P], _I),
    "fsg_chamfer_nn_bwd_workspace_bytes": ([_I, _I, _I], ctypes.c_size_t),
    "fsg_chamfer_nn_bwd_f32": ([_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P], _I),
    "fsg_ensemble_accumulate_workspace_bytes": ([_I, _L], ctypes.c_size_t),
    "fsg_ensemble_accumulate_f32": ([_P, _I, _I, _I, _I, _P, _L, _P, _P, _P], _I),
    "fsg_sample_transform_f32": ([_P, _I, _I, _L, _P, _I, _P, _P, _P], _I),
    "fsg_colsum_narrow_f32": ([_P, _L, _I, _P, _P], _I),
    "fsg_fold_layer1_f32": ([_P, _I, _P, _L, _P, _I, _I, _I, _I, _P, _P], _I),
    "fsg_adam_flat_f32": ([_P, _P, _P, _P, _P, _L, _F, _P, _F, _F, _F, _F, _P], _I),
    "fsg_nnu_loss_workspace_bytes": ([_I], ctypes.c_size_t),
    "fsg_nnu_loss_f32": ([_P, _L, _L, _L, _P, _P, _I, _I, _I, _F, _F, _F, _P, _P, _L, _L, _L, _P, _P], _I),
    "fsg_bn_rows_workspace_bytes": ([ctypes.c_long, _I], ctypes.c_size_t),
    "fsg_bn_rows_fwd_f32": ([_P, _P, _P, _P, _P, _P, ctypes.c_long, _I, _I, _F, _F, _I, _P, _P, _P, _P, _P], _I),
    "fsg_bn_rows_bwd_f32": ([_P, _P, _P, _P, _P, _P, ctypes.c_long, _I, _I, _I, _P, _P, _P, _P, _P, _P], _I),
    "fsg_gemm_small_workspace_bytes": ([_I, _I, _I], ctypes.c_size_t),
    "fsg_gemm_small_f32": ([_P, _L, _L, _P, _L, _L, _P, _P, _L, _I, _I, _I, _P, _P], _I),
    "fsg_gemm_small_rowsum_f32": ([_P, _L, _L, _P, _L, _L, _P, _P, _L, _I, _I, _I, _P, _P, _P], _I),
    "fsg_gemm_small_deferred_f32": ([_P, _L, _L, _P, _L, _L, _P, _L, _I, _I, _I, _P, _P, _P, _P], _I),
    "fsg_gemm_small_reduce_many_f32": ([_P, _P], _I),
    "fsg_gemm_small_bf16": ([_P, _L, _L, _P, _L, _L, _P, _P, _L, _I, _I, _I, _P, _P, _P, _P], _I),
    "fsg_pt_attn_workspace_bytes": ([_I, _I, _I], ctypes.c_size_t),
    "fsg_pt_attn_fwd_f32": ([_P, _P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P], _I),
    "fsg_pt_attn_bwd_f32": ([_P, _P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P], _I),
    "fsg_knn_segment_f32": ([_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P], _I),
    "fsg_fps_f32": ([_P, _P, _P, _I, _I, _P, _P, _P], _I),
    "fsg_group_gather_fwd_f32": ([_P, _P, _P, _I, _I, _I, _I, _P], _I),
    "fsg_group_gather_bwd_f32": ([_P, _P, _P, _I, _I, _I, _I, _P], _I),
    "fsg_group_xyz_feat_fwd_f32": ([_P, _P, _P, _P, _P, _I, _I, _I, _I, _P], _I),
    "fsg_group_xyz_feat_bwd_f32": ([_P, _P, _P, _I, _I, _I, _I, _P], _I),
    "fsg_rows_max_fwd_f32": ([_P, _P, _P, _I, _I, _I, _P], _I),
    "fsg_rows_max_bwd_f32": ([_P, _P, _P, _I, _I, _I, _P], _I),
    "fsg_interp_fwd_f32": ([_P, _P, _P, _P, _I, _I, _I, _I, _P], _I),
    "fsg_interp_bwd_f32": ([_P, _P, _P, _P, _I, _I, _I, _I, _P], _I),
    "fsg_vec_attn_fwd_f32": ([_P, _P, _P, _P, _P, _I, _I, _I, _I, _P], _I),
    "fsg_vec_attn_bwd_f32": ([_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P], _I),
    "fsg_pw_weight_image_bytes": ([_I, _I], ctypes.c_size_t),
    "fsg_pw_weight_image_f32": ([_P, _L, _L, _I, _I, _F, _I, _I, _P, _P], _I),
    "fsg_pw_weight_images_f32": ([_P, _P], _I),
    "fsg_pw_linear_f32": ([_P, _L, _P, _P, _P, _L, _I, _I, _I, _I, _P], _I),
    "fsg_pw_weight_image_bf16": ([_P, _L, _L, _I, _I, _P, _P], _I),
    "fsg_pw_linear_bf16": ([_P, _L, _P, _P, _P, _L, _I, _I, _I, _I, _P], _I),
    "fsg_pw_tn_bf16": ([_P, _I, _P, ctypes.c_size_t, _P, _L, _P], _I),
    "fsg_pw_tile_rows": ([_I], _I),
    "fsg_pw_rowgemm_f32": ([_P, _I, _I, _I, _P], _I),
    "fsg_pw_tn_workspace_bytes": ([_I, _I, _I, _I], ctypes.c_size_t),
    "fsg_pw_tn_f32": ([_P, _I, _P, ctypes.c_size_t, _P, _L, _P, _L, _P], _I),
    "fsg_pw_tn_reduce_f32": ([_P, _P], _I),
    "fsg_pw_bn_finalize_f32": ([_P, _I, _I, _I, _I, _P, _I, _I, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P, _P], _I),
    "fsg_pw_cloud_linear_f32": ([_P, _P, _L, _I, _I, _I, _P, _P], _I),
    "fsg_pw_max_finish_f32": ([_P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P, _P], _I),
    "fsg_pw_bn_finalize_max_f32": ([_P, _I, _I, _I, _I, _I, _I, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _P, _P, _P,
                                   _P], _I),
    "fsg_pw_bnbwd_finalize_f32": ([_P, _I, _I, _I, _L, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P], _I),
    "fsg_pw_logits_bwd_f32": ([_P, _I, _P, _P, _P, _P, _P, _P, _L, _I, _F, _P, _P, _P], _I),
    "fsg_pw_gf_prep_f32": ([_P, _P, _L, _I, _P, _P, _L, _P, _P, _P, _P, _P, _P, _I, _I, _L, _I, _F, _P, _P, _P, _P, _P, _P, _L, _I, _P, _L,
                            _P], _I),
    "fsg_pw_scatter_rows_workspace_bytes": ([_I, _I], ctypes.c_size_t),
    "fsg_pw_scatter_rows_f32": ([_P, _P, _P, _L, _I, _I, _I, _I, _P, _L, _P, _P], _I),
    "fsg_pw_gf_dw_f32": ([_P, _P, _P, _L, _P, _P, _L, _P, _P, _P, _I, _I, _I, _I, _P, _L, _P], _I),
}
for _name, (_args, _res) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = header and library out of sync
    _fn.argtypes, _fn.restype = _args, _res

KNN_FIX_DIAG, KNN_DROP_FIRST, KNN_FORCE_ROWS, KNN_FORCE_MFMA, KNN_MAX_K = 1, 2, 4, 8, 64


_timing = None  # {entry point: [(start_event, end_event), ...]} while bench.py measures kernel durations


def start_timing():
    """Bracket every C-ABI call with HIP events on the stream it is launched on (torch's current stream)."""
    global _timing
    _timing = {}


def stop_timing():
    """-> {entry point: [milliseconds per call]} (synchronises)."""
    global _timing
    rec, _timing = _timing, None
    torch.cuda.synchronize()
    return {name: [s.elapsed_time(e) for s, e in evs] for name, evs in (rec or {}).items()}


class PTLayerParams(ctypes.Structure):
    """include/fsg_hip.h: fsg_pt_layer_params"""
    _fields_ = [(n, _P) for n in ("lp1_w", "lp1_b", "bnp_g", "bnp_b", "bnp_rm", "bnp_rv", "lp2_w", "lp2_b", "bn1_g", "bn1_b",
                                  "bn1_rm", "bn1_rv", "lw1_w", "lw1_b", "bn2_g", "bn2_b", "bn2_rm", "bn2_rv", "lw2_w",
                                  "lw2_b")] + [(n, _F) for n in ("eps_p", "eps_1", "eps_2", "mom_p", "mom_1", "mom_2")]


GEMM_REDUCE_MAX_JOBS = 48


class GemmReduceJobs(ctypes.Structure):
    """include/fsg_hip.h: fsg_gemm_reduce_jobs"""
    _fields_ = [("part", _P * GEMM_REDUCE_MAX_JOBS), ("C", _P * GEMM_REDUCE_MAX_JOBS), ("rowsum", _P * GEMM_REDUCE_MAX_JOBS),
                ("ldc", _L * GEMM_REDUCE_MAX_JOBS), ("S", _I * GEMM_REDUCE_MAX_JOBS), ("I", _I * GEMM_REDUCE_MAX_JOBS),
                ("J", _I * GEMM_REDUCE_MAX_JOBS), ("blocks", _I * GEMM_REDUCE_MAX_JOBS), ("n", _I)]


class EdgeWeightJobs(ctypes.Structure):
    """include/fsg_hip.h: fsg_edge_weight_jobs"""
    _fields_ = [("src", _P * 8), ("dst", _P * 8), ("Co", _I * 8), ("C", _I * 8), ("n", _I)]


class PWRowGemmArgs(ctypes.Structure):
    """include/fsg_hip.h: fsg_pw_rowgemm_args"""
    _fields_ = [("A1", _P), ("Y1", _P), ("A2", _P), ("lda1", _L), ("lda2", _L), ("K1", _I), ("K2", _I), ("Bimg", _P),
                ("M", _I), ("N", _I), ("rows_per_cloud", _I), ("alpha", _P), ("delta", _P), ("P", _P), ("Q", _P),
                ("tstride", _I), ("slope", _F), ("C", _P), ("ldc", _L), ("store_n0", _I), ("bias", _P), ("rec", _P),
                ("sgn", _P), ("sel_val", _P), ("sel_arg", _P), ("sel_n", _I), ("Yp", _P), ("ldyp", _L), ("ealpha", _P),
                ("edelta", _P), ("emu", _P), ("er", _P), ("etstride", _I), ("rec2", _P)]


PW_MAX_IMAGE_JOBS = 10


class PWImageJobs(ctypes.Structure):
    """include/fsg_hip.h: fsg_pw_image_jobs"""
    _fields_ = [("W", _P * PW_MAX_IMAGE_JOBS), ("stride_n", _L * PW_MAX_IMAGE_JOBS), ("stride_k", _L * PW_MAX_IMAGE_JOBS),
                ("N", _I * PW_MAX_IMAGE_JOBS), ("K", _I * PW_MAX_IMAGE_JOBS), ("ks0", _I * PW_MAX_IMAGE_JOBS),
                ("KS", _I * PW_MAX_IMAGE_JOBS), ("scale", _F * PW_MAX_IMAGE_JOBS), ("image", _P * PW_MAX_IMAGE_JOBS), ("n", _I)]


PW_MAX_REDUCE_JOBS = 6


class PWTnReduceJobs(ctypes.Structure):
    """include/fsg_hip.h: fsg_pw_tn_reduce_jobs"""
    _fields_ = [("workspace", _P * PW_MAX_REDUCE_JOBS), ("C1", _P * PW_MAX_REDUCE_JOBS), ("C2", _P * PW_MAX_REDUCE_JOBS),
                ("ldc1", _L * PW_MAX_REDUCE_JOBS), ("ldc2", _L * PW_MAX_REDUCE_JOBS), ("S", _I * PW_MAX_REDUCE_JOBS),
                ("N1", _I * PW_MAX_REDUCE_JOBS), ("N2", _I * PW_MAX_REDUCE_JOBS), ("N1a", _I * PW_MAX_REDUCE_JOBS), ("n", _I)]


class PWTnArgs(ctypes.Structure):
    """include/fsg_hip.h: fsg_pw_tn_args"""
    _fields_ = [("L1", _P), ("LY1", _P), ("L2", _P), ("ldl1", _L), ("ldl2", _L), ("N1a", _I), ("N1b", _I), ("lpro", _I),
                ("lalpha", _P), ("ldelta", _P), ("lP", _P), ("lQ", _P), ("lts", _I), ("R", _P), ("ldr", _L), ("N2", _I),
                ("rpro", _I), ("ralpha", _P), ("rdelta", _P), ("rts", _I), ("slope", _F), ("M", _I), ("rows_per_cloud", _I),
                ("rows_per_slice", _I), ("ones", _I)]


class PTLayerGrads(ctypes.Structure):
    """include/fsg_hip.h: fsg_pt_layer_grads"""
    _fields_ = [(n, _P) for n in ("lp1_w", "lp1_b", "bnp_g", "bnp_b", "lp2_w", "lp2_b", "bn1_g", "bn1_b", "lw1_w", "lw1_b",
                                  "bn2_g", "bn2_b", "lw2_w", "lw2_b")]


_experiments = None


def experiments():
    """libfsg_hip_experiments.so: the superseded kNN designs (cross-checks for tests/, baselines for tools/) -- test
    infrastructure, loaded on first use, never by the product path"""
    global _experiments
    if _experiments is None:
        path = os.path.join(_HERE, "libfsg_hip_experiments.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} not found: `make -C {os.path.join(_HERE, 'csrc')} all` builds it")
        x = ctypes.CDLL(path)
        x.fsg_knn_experiment_f32.argtypes, x.fsg_knn_experiment_f32.restype = SIGNATURES["fsg_knn_dense_f32"][0], _I
        x.fsg_last_error.restype = ctypes.c_char_p
        _experiments = x
    return _experiments


def call(name, *args):
    if _timing is None:
        rc = getattr(lib, name)(*args)
    else:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = getattr(lib, name)(*args)
        e.record()
        _timing.setdefault(name, []).append((s, e))
    if rc != 0:
        raise RuntimeError(f"{name} failed (code {rc}): {lib.fsg_last_error().decode()}")
