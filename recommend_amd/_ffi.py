"""ctypes binding of librecommend_amd.so (the C ABI in include/recommend_amd.h).

Only plain pointers and sizes cross the boundary.  Tensors are passed as
``tensor.data_ptr()``; the caller keeps them alive until the stream has run.
There is no CPU implementation behind these calls: if the library is missing,
or a device entry point is used without a GPU, the call raises.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int32,
                    c_int64, c_size_t, c_uint32, c_uint64, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "librecommend_amd.so")

RK_OK = 0
RK_ERR_ARG = -1
RK_ERR_HIP = -2
RK_ERR_UNSUPPORTED = -3
RK_ERR_NODEV = -4


class RkError(RuntimeError):
    """A librecommend_amd call returned a negative code."""

    def __init__(self, code, message):
        super().__init__(f"librecommend_amd error {code}: {message}")
        self.code = code
        self.message = message


class AdamTensor(Structure):
    _fields_ = [("param", c_void_p), ("grad", c_void_p), ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p),
                ("n", c_int64)]


class SpmmPlan(Structure):
    _fields_ = [
        ("long_thresh", c_int32),
        ("n_chunks", c_int32),
        ("chunk_row", c_void_p),
        ("chunk_begin", c_void_p),
        ("chunk_end", c_void_p),
        ("chunk_slot", c_void_p),
        ("n_multi", c_int32),
        ("multi_rows", c_void_p),
        ("multi_slot_ptr", c_void_p),
        ("n_slots", c_int32),
        ("partial", c_void_p),
        ("arrive", c_void_p),
    ]


DENSE_REDUCE_MAX_SLABS = 8


class DenseReduceJob(Structure):
    _fields_ = [
        ("slabs", c_void_p * DENSE_REDUCE_MAX_SLABS),
        ("n_slabs", c_int32),
        ("n_part", c_int32),
        ("dW", c_void_p),
        ("db", c_void_p),
        ("dalpha", c_void_p),
    ]


P = c_void_p
# name -> (restype, argtypes); mirrors include/recommend_amd.h declaration by declaration
SIGNATURES = {
    "rk_abi_version": (c_int32, []),
    "rk_last_error": (c_char_p, []),
    "rk_device_count": (c_int32, []),
    "rk_build_coo_host": (c_int32, [P, P, P, c_int64, c_int32, c_int32, c_int32, P, P, P]),
    "rk_csr_from_coo_host": (c_int32, [P, P, c_int64, c_int32, P, P, P]),
    "rk_spmm_plan_host": (c_int32, [P, c_int32, c_int32, POINTER(c_int32), POINTER(c_int32),
                                    POINTER(c_int32), P, P, P, P, P, P]),
    "rk_sym_normalize_f32": (c_int32, [P, P, P, P, c_int32, P, P, P]),
    "rk_in_weight_sqrt_f32": (c_int32, [P, P, P, c_int32, P, P]),
    "rk_sym_scale_rows_f32": (c_int32, [P, P, P, P, c_int32, c_int32, P, P, P]),
    "rk_deg_normalize_f32": (c_int32, [P, P, P, c_int32, P, P]),
    "rk_edge_feature_sim_f32": (c_int32, [P, P, c_int64, P, c_int32, c_int32, c_int32, c_int32, P, P]),
    "rk_gather_f32": (c_int32, [P, P, c_int64, P, P]),
    "rk_spmm_csr_f32": (c_int32, [P, P, P, c_int32, P, c_int32, P, P, c_float, c_float,
                                  POINTER(SpmmPlan), P]),
    "rk_proj_fwd_f32": (c_int32, [P, c_int32, c_int32, P, P, c_int32, c_int32, P, P]),
    "rk_proj_fwd_ws_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "rk_proj_fwd_ws_f32": (c_int32, [P, c_int32, c_int32, P, P, c_int32, c_int32, P, P, c_size_t, P]),
    "rk_proj_bwd_ws_bytes": (c_size_t, [c_int32, c_int32, c_int32, c_int32]),
    "rk_proj_bwd_f32": (c_int32, [P, c_int32, c_int32, P, c_int32, c_int32, P, P, P, c_size_t, P]),
    "rk_dense_fwd_f32": (c_int32, [P, P, P, P, P, P, P, c_float, c_uint64, P, P, c_int64, c_int32,
                                   c_int32, P, P]),
    "rk_dense_bwd_ws_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "rk_dense_bwd_f32": (c_int32, [P, P, P, P, c_float, c_uint64, P, P, c_int64, c_int32, c_int32,
                                   P, P, P, P, P, c_size_t, P]),
    "rk_dense_bwd_accum_f32": (c_int32, [P, P, P, P, c_float, c_uint64, P, P, c_int64, c_int32, c_int32,
                                         P, P, P, P, c_int32, P, c_size_t, P]),
    "rk_dense_bwd_slab_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "rk_dense_bwd_partials_f32": (c_int32, [P, P, P, P, c_float, c_uint64, P, P, c_int64, c_int32, c_int32,
                                            P, P, c_size_t, POINTER(c_int32), P]),
    "rk_dense_reduce_jobs_f32": (c_int32, [P, c_int32, P]),
    "rk_dense_reduce_jobs_d_f32": (c_int32, [P, c_int32, c_int32, P]),
    "rk_modal_sum_fwd_f32": (c_int32, [P, c_int32, P, c_int64, P, P]),
    "rk_modal_sum_bwd_ws_bytes": (c_size_t, []),
    "rk_modal_sum_bwd_f32": (c_int32, [P, P, c_int32, P, P, c_int64, P, P, c_size_t, P]),
    "rk_adam_ws_bytes": (c_size_t, [P, c_int32]),
    "rk_adam_clip_step_f32": (c_int32, [P, c_int32, P, P, c_float, c_float, c_float, c_float, c_float, P, P,
                                        c_size_t, P]),
    "rk_adam_sqnorm_f32": (c_int32, [P, c_int32, c_int32, P, P, P, c_size_t, P]),
    "rk_adam_update_f32": (c_int32, [P, c_int32, c_int32, P, P, c_float, c_float, c_float, c_float, c_float, P, P, P,
                                     c_size_t, P]),
    "rk_mt_seed_host": (c_int32, [P, c_uint64]),
    "rk_neg_sample_host": (c_int32, [P, P, c_int64, c_int32, P, P, c_int32, P, P]),
    "rk_neg_order_host": (c_int32, [P, c_int32, c_int32, P]),
    "rk_neg_sample_cached_host": (c_int32, [P, P, c_int64, c_int32, P, P, c_int32, P, P, P, c_int32, P, P]),
    "rk_bpr_ws_bytes": (c_size_t, [c_int64, c_int32]),
    "rk_bpr_fwd_f32": (c_int32, [P, P, c_int32, P, P, P, c_int64, c_int32, P, P, P, P, c_int32,
                                 P, P, P, c_size_t, P]),
    "rk_bpr_fwd_noalloc_max_batch": (c_int64, []),
    "rk_bpr_fwd_noalloc_f32": (c_int32, [P, P, c_int32, P, P, P, c_int64, c_int32, P, P, P, P, c_int32,
                                         P, P, P]),
    "rk_bpr_bwd_f32": (c_int32, [P, P, c_int32, P, P, P, c_int64, c_int32, P, P, P, P, P, P, P]),
    "rk_rank_loss_fwd_f32": (c_int32, [P, P, c_int32, P, P, P, c_int64, c_int32, P, P, P, P, c_int32, c_int32,
                                       P, P, P, c_size_t, P]),
    "rk_rank_loss_bwd_f32": (c_int32, [P, P, c_int32, P, P, P, c_int64, c_int32, P, P, P, P, c_int32, P, P, P]),
    "rk_bpr_scores_fwd_f32": (c_int32, [P, P, c_int64, c_int32, P, P, P, P, c_size_t, P]),
    "rk_bpr_scores_bwd_f32": (c_int32, [P, P, P, P, c_int64, c_int32, P, P, P]),
    "rk_score_topk_ws_bytes": (c_size_t, [c_int32, c_int32, c_int32, c_int32]),
    "rk_score_topk_f32": (c_int32, [P, c_int32, P, c_int32, c_int32, P, P, c_int32, c_int32,
                                    P, P, P, c_size_t, P]),
    "rk_score_topk_exact_f32": (c_int32, [P, c_int32, P, c_int32, c_int32, P, P, c_int32, c_int32,
                                          P, P, P, c_size_t, P]),
    "rk_score_topk_filtered_f32": (c_int32, [P, c_int32, P, c_int32, c_int32, P, P, c_int32, c_int32,
                                             P, P, P, c_size_t, P]),
    "rk_score_topk_open_users": (c_int32, [c_int32, c_int32, c_int32, c_int32, P, c_size_t, P, P]),
    "rk_topk_merge_f32": (c_int32, [P, P, c_int32, c_int32, c_int32, c_int32, P, P, P]),
    "rk_rows_topk_ws_bytes": (c_size_t, [c_int32, c_int32]),
    "rk_score_topk_wide_ws_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "rk_score_topk_wide_f32": (c_int32, [P, c_int32, P, c_int32, c_int32, c_int32, P, P, P, c_size_t, P]),
    "rk_rows_topk_f32": (c_int32, [P, c_int32, c_int32, c_int64, c_int32, P, P, P, c_size_t, P]),
    "rk_rank_metrics_ws_bytes": (c_size_t, [c_int32, c_int32]),
    "rk_rank_metrics_f32": (c_int32, [P, c_int32, c_int32, P, P, P, c_int32, P, P, P, c_int64,
                                      P, c_size_t, P]),
}

_lib = None


def load():
    """Load the shared library once; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # RECOMMEND_AMD_LIB names another build of the same C ABI (kernel experiments)
    path = os.environ.get("RECOMMEND_AMD_LIB", LIB_PATH)
    if not os.path.exists(path):
        raise RkError(RK_ERR_NODEV, f"{path} is missing: run `python -m recommend_amd.build` "
                      "(or __graft_entry__.build()) first; there is no CPU fallback")
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.rk_abi_version() != 1:
        raise RkError(RK_ERR_UNSUPPORTED, "librecommend_amd ABI version mismatch")
    _lib = lib
    return lib


def last_error():
    msg = load().rk_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(code):
    if code != RK_OK:
        raise RkError(code, last_error())


def call(name, *args):
    """Call an int-returning entry point and raise RkError on failure."""
    check(getattr(load(), name)(*args))


def device_count():
    return int(load().rk_device_count())


def require_device():
    """Device entry points need a GPU; fail loudly instead of computing elsewhere."""
    if device_count() < 1:
        raise RkError(RK_ERR_NODEV, "no HIP device visible: the recommend_amd device path "
                      "runs only on a GPU and has no CPU fallback")


def ptr(t):
    """Device or host address of a tensor / numpy array (None -> NULL)."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return c_void_p(t.data_ptr())
    return c_void_p(t.ctypes.data)


def stream_ptr(device=None):
    import torch
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


__all__ = ["load", "call", "check", "ptr", "stream_ptr", "SpmmPlan", "DenseReduceJob", "RkError", "SIGNATURES",
           "LIB_PATH", "device_count", "require_device", "last_error",
           "c_int32", "c_int64", "c_float", "c_double", "c_uint32", "c_size_t"]
