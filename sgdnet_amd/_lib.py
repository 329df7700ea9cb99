"""ctypes binding of libsgdnet_hip.so (include/sgdnet_hip.h).

The library is the product; this module only loads it.  There is no Python or
CPU fallback: if the shared object is missing, loading raises.
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# SGDNET_LIB_PATH: development aid (A/B of kernel variants built side by side); the product path is lib/
LIB_PATH = os.environ.get("SGDNET_LIB_PATH") or os.path.join(_HERE, "lib", "libsgdnet_hip.so")

# every symbol include/sgdnet_hip.h declares
EXPORTS = [
    "sgdnet_abi_version", "sgdnet_last_error", "sgdnet_device_count", "sgdnet_set_option", "sgdnet_get_option",
    "sgdnet_fit_sparse", "sgdnet_fit_dense",
    "sgdnet_rng_seed", "sgdnet_rng_unif", "sgdnet_rng_fill", "sgdnet_rng_jump_poly", "sgdnet_rng_jump",
    "sgdnet_solver_create", "sgdnet_solver_destroy", "sgdnet_solver_set_penalty",
    "sgdnet_solver_get_state", "sgdnet_solver_set_state", "sgdnet_solver_upload_stream",
    "sgdnet_solver_generate_stream", "sgdnet_solver_get_stream",
    "sgdnet_solver_run", "sgdnet_solver_enqueue_epochs", "sgdnet_solver_sync",
    "sgdnet_solver_profile_epoch", "sgdnet_solver_deviance", "sgdnet_solver_snapshot",
    "sgdnet_solver_export_delta", "sgdnet_solver_apply_merged", "sgdnet_solver_delta_len",
    "sgdnet_solver_convergence", "sgdnet_solver_last_change", "sgdnet_auto_batch", "sgdnet_shard_window",
    "sgdnet_solver_gather_form", "sgdnet_solver_stream", "sgdnet_solver_export_delta_async",
    "sgdnet_solver_apply_merged_async", "sgdnet_solver_sync_buffer_len", "sgdnet_solver_sync_bind",
    "sgdnet_solver_sync_begin", "sgdnet_solver_sync_gather", "sgdnet_solver_sync_sweep",
    "sgdnet_solver_sync_end", "sgdnet_solver_set_n_total",
    "sgdnet_solver_export_delta_weighted_async", "sgdnet_solver_set_virtual_shards", "sgdnet_solver_set_merge_period",
    "sgdnet_score_sparse", "sgdnet_score_dense", "sgdnet_predict_sparse", "sgdnet_predict_dense",
    "sgdnet_auc_sparse", "sgdnet_auc_dense", "sgdnet_auc_sparse_rng", "sgdnet_auc_dense_rng",
    "sgdnet_solver_link_peers", "sgdnet_solver_set_cu_budget", "sgdnet_solver_epoch_timing",
    "sgdnet_solver_peer_info_bytes", "sgdnet_solver_peer_info", "sgdnet_solver_link_ipc",
    "sgdnet_solver_rng_layout", "sgdnet_solver_rng_open", "sgdnet_solver_rng_next", "sgdnet_solver_rng_done", "sgdnet_solver_rng_close",
]
ABI_VERSION = 4   # include/sgdnet_hip.h: SGDNET_ABI_VERSION
MEASURES = {"deviance": 0, "mse": 1, "mae": 2, "class": 3, "auc": 4}

FAMILIES = {"gaussian": 0, "binomial": 1, "multinomial": 2, "mgaussian": 3}
PENALTIES = {"ridge": 0, "elasticnet": 1, "grouplasso": 2}
MODES = {"exact": 0, "batched": 1, "auto": 2}

UNIF_FN = C.CFUNCTYPE(C.c_double, C.c_void_p)
LOSSES_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_int)


class Csc(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("n_cols", C.c_int64),
                ("colptr", C.POINTER(C.c_int32)), ("rowidx", C.POINTER(C.c_int32)),
                ("values", C.POINTER(C.c_double))]


class Rng(C.Structure):
    _fields_ = [("mti", C.c_uint32), ("mt", C.c_uint32 * 624)]


class Control(C.Structure):
    _fields_ = [("debug", C.c_int), ("elasticnet_mix", C.c_double), ("family", C.c_int),
                ("intercept", C.c_int), ("is_sparse", C.c_int),
                ("lambda_", C.POINTER(C.c_double)), ("n_lambda_user", C.c_int),
                ("lambda_min_ratio", C.c_double), ("max_iter", C.c_uint), ("n_lambda", C.c_int),
                ("n_classes", C.c_int), ("standardize", C.c_int),
                ("standardize_response", C.c_int), ("tol", C.c_double),
                ("type_multinomial", C.c_int),
                ("sample_stream", C.POINTER(C.c_uint32)), ("sample_stream_len", C.c_int64),
                ("unif", UNIF_FN), ("unif_ctx", C.c_void_p), ("seed", C.c_uint32),
                ("rng_state", C.POINTER(Rng)),
                ("mode", C.c_int), ("batch", C.c_int64), ("device", C.c_int),
                ("losses_sink", LOSSES_FN), ("losses_ctx", C.c_void_p),
                ("n_gpus", C.c_int), ("devices", C.POINTER(C.c_int))]


class Result(C.Structure):
    _fields_ = [("a0", C.POINTER(C.c_double)), ("beta", C.POINTER(C.c_double)),
                ("lambda_", C.POINTER(C.c_double)), ("dev_ratio", C.POINTER(C.c_double)),
                ("return_codes", C.POINTER(C.c_double)), ("losses", C.POINTER(C.c_double)),
                ("losses_len", C.POINTER(C.c_int32)), ("nulldev", C.c_double),
                ("npasses", C.c_double), ("draws_used", C.c_int64)]


class Problem(C.Structure):
    _fields_ = [("family", C.c_int), ("n_classes", C.c_int), ("n_samples", C.c_int64),
                ("n_total", C.c_int64), ("n_features", C.c_int64), ("fit_intercept", C.c_int),
                ("standardize", C.c_int),
                ("rowptr", C.POINTER(C.c_int64)), ("colidx", C.POINTER(C.c_int32)),
                ("values", C.POINTER(C.c_double)), ("x_dense", C.POINTER(C.c_double)),
                ("x_center_scaled", C.POINTER(C.c_double)), ("y", C.POINTER(C.c_double)),
                ("y_rows", C.c_int), ("device", C.c_int)]


class SgdnetError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libsgdnet_hip error {code}: {msg}")
        self.code = code


_lib = None
# PyTorch-ROCm wheels bundle their own HIP/HSA runtime under the system's sonames.  Whichever of
# torch and libsgdnet_hip.so is loaded first decides which runtime the process uses; loading the
# library first and torch afterwards ends with two HSA runtimes and torch seeing no GPU.  Only the
# multi-GPU plumbing (parallel.py) uses torch, and it checks this flag.
loaded_before_torch = False


def require_torch_first():
    if loaded_before_torch:
        raise RuntimeError("libsgdnet_hip.so was loaded before torch in this process: import torch "
                           "before sgdnet_amd touches the device when using sgdnet_amd.parallel")


def load():
    """Load libsgdnet_hip.so; raises if it has not been built (no fallback)."""
    global _lib, loaded_before_torch
    if _lib is not None:
        return _lib
    loaded_before_torch = "torch" not in sys.modules
    if not os.path.exists(LIB_PATH):
        raise OSError(f"{LIB_PATH} not found: run ./build.sh (or __graft_entry__.build()); "
                      "the SAGA backend has no CPU fallback")
    L = C.CDLL(LIB_PATH)
    if L.sgdnet_abi_version() != ABI_VERSION:          # the structs below are laid out for this version
        raise OSError(f"{LIB_PATH} has ABI version {L.sgdnet_abi_version()}, this binding was written for "
                      f"{ABI_VERSION} (include/sgdnet_hip.h: SGDNET_ABI_VERSION): rebuild with ./build.sh")
    L.sgdnet_last_error.restype = C.c_char_p
    L.sgdnet_set_option.argtypes = [C.c_char_p, C.c_int]
    L.sgdnet_get_option.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
    L.sgdnet_rng_unif.restype = C.c_double
    L.sgdnet_solver_delta_len.restype = C.c_int64
    L.sgdnet_solver_destroy.restype = None
    L.sgdnet_solver_create.argtypes = [C.POINTER(Problem), C.POINTER(C.c_void_p)]
    L.sgdnet_solver_destroy.argtypes = [C.c_void_p]
    L.sgdnet_solver_set_penalty.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double]
    L.sgdnet_solver_get_state.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
    L.sgdnet_solver_set_state.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
    L.sgdnet_solver_upload_stream.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_int64]
    L.sgdnet_solver_generate_stream.argtypes = [C.c_void_p, C.POINTER(Rng), C.c_int64]
    L.sgdnet_solver_get_stream.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_int64, C.c_int64]
    L.sgdnet_solver_run.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_uint,
                                    C.c_double, C.POINTER(C.c_uint), C.POINTER(C.c_int),
                                    C.POINTER(C.c_double)]
    L.sgdnet_solver_enqueue_epochs.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int]
    L.sgdnet_solver_sync.argtypes = [C.c_void_p]
    L.sgdnet_solver_profile_epoch.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                              C.POINTER(C.c_double), C.POINTER(C.c_int),
                                              C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.sgdnet_solver_deviance.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.sgdnet_solver_snapshot.argtypes = [C.c_void_p]
    L.sgdnet_solver_export_delta.argtypes = [C.c_void_p, C.c_void_p]
    L.sgdnet_solver_apply_merged.argtypes = [C.c_void_p, C.c_void_p, C.c_double]
    L.sgdnet_solver_delta_len.argtypes = [C.c_void_p]
    L.sgdnet_solver_stream.argtypes = [C.c_void_p]
    L.sgdnet_solver_stream.restype = C.c_void_p
    L.sgdnet_solver_export_delta_async.argtypes = [C.c_void_p, C.c_void_p]
    L.sgdnet_solver_apply_merged_async.argtypes = [C.c_void_p, C.c_void_p, C.c_double]
    L.sgdnet_solver_sync_buffer_len.argtypes = [C.c_void_p]
    L.sgdnet_solver_sync_buffer_len.restype = C.c_int64
    L.sgdnet_solver_sync_bind.argtypes = [C.c_void_p, C.c_void_p]
    L.sgdnet_solver_sync_begin.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
    L.sgdnet_solver_sync_gather.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int]
    L.sgdnet_solver_sync_sweep.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int]
    L.sgdnet_solver_sync_end.argtypes = [C.c_void_p, C.c_int]
    L.sgdnet_solver_set_n_total.argtypes = [C.c_void_p, C.c_int64]
    L.sgdnet_solver_set_virtual_shards.argtypes = [C.c_void_p, C.c_int]
    L.sgdnet_solver_set_merge_period.argtypes = [C.c_void_p, C.c_int64]
    L.sgdnet_solver_link_peers.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    L.sgdnet_solver_set_cu_budget.argtypes = [C.c_void_p, C.c_int]
    L.sgdnet_solver_peer_info.argtypes = [C.c_void_p, C.c_void_p]
    L.sgdnet_solver_link_ipc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.sgdnet_solver_epoch_timing.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.sgdnet_solver_export_delta_weighted_async.argtypes = [C.c_void_p, C.c_void_p, C.c_double]
    L.sgdnet_solver_convergence.argtypes = [C.c_void_p, C.c_double, C.POINTER(C.c_int)]
    L.sgdnet_solver_last_change.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.sgdnet_solver_gather_form.argtypes = [C.c_void_p, C.c_int64]
    L.sgdnet_auto_batch.argtypes = [C.c_double, C.c_double]
    L.sgdnet_auto_batch.restype = C.c_int64
    L.sgdnet_shard_window.argtypes = [C.c_int64, C.c_int64]
    L.sgdnet_shard_window.restype = C.c_int64
    L.sgdnet_fit_sparse.argtypes = [C.POINTER(Csc), C.POINTER(C.c_double), C.c_int,
                                    C.POINTER(Control), C.POINTER(Result)]
    L.sgdnet_fit_dense.argtypes = [C.POINTER(C.c_double), C.c_int64, C.c_int64,
                                   C.POINTER(C.c_double), C.c_int, C.POINTER(Control),
                                   C.POINTER(Result)]
    L.sgdnet_rng_fill.argtypes = [C.POINTER(Rng), C.c_uint32, C.POINTER(C.c_uint32), C.c_int64]
    L.sgdnet_solver_rng_open.argtypes = [C.c_void_p, C.POINTER(Rng), C.c_int64, C.c_int]
    L.sgdnet_solver_rng_layout.argtypes = [C.c_void_p, C.c_int64]
    L.sgdnet_solver_rng_next.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.sgdnet_solver_rng_done.argtypes = [C.c_void_p]
    L.sgdnet_solver_rng_close.argtypes = [C.c_void_p, C.POINTER(Rng)]
    L.sgdnet_rng_jump_poly.argtypes = [C.c_uint64, C.POINTER(C.c_uint32)]
    L.sgdnet_rng_jump.argtypes = [C.POINTER(Rng), C.POINTER(C.c_uint32), C.POINTER(Rng)]
    L.sgdnet_rng_jump.restype = None
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise SgdnetError(rc, load().sgdnet_last_error().decode())


def dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))
