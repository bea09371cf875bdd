"""ctypes binding of libfmhip.so — the same C ABI (include/fmhip.h, include/fmhip_experimental.h) a JNI shim would bind.

There is NO CPU fallback: if the HIP library is missing this module raises, loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FMHIP_LIB") or os.path.join(_HERE, "lib", "libfmhip.so")   # FMHIP_LIB: A/B builds

OK = 0
K_FORWARD, K_REDUCE, K_BACKWARD, K_FIXUP, K_APPLY, K_COUNT = 0, 1, 2, 3, 4, 5
KERNEL_NAMES = ("forward", "reduce", "backward", "fixup", "apply")
RANGE_LEN = 64

# every symbol include/fmhip.h declares — the PRODUCT surface (tests check the library exports all of them)
SYMBOLS = (
    "fmhip_version", "fmhip_last_error", "fmhip_device_count",
    "fmhip_model_create", "fmhip_model_destroy", "fmhip_model_info", "fmhip_model_init_normal",
    "fmhip_model_set_params", "fmhip_model_get_params", "fmhip_model_get_rows", "fmhip_model_set_params_f32", "fmhip_model_get_params_f32",
    "fmhip_synchronize",
    "fmhip_dataset_create", "fmhip_dataset_create_f32", "fmhip_dataset_create_opts", "fmhip_rows_create", "fmhip_rows_create_f32",
    "fmhip_dataset_destroy", "fmhip_dataset_info", "fmhip_dataset_batch_info", "fmhip_dataset_get_transpose",
    "fmhip_predict", "fmhip_predict_rows", "fmhip_rmse", "fmhip_residual", "fmhip_term_q",
    "fmhip_sgd_step", "fmhip_sgd_epoch", "fmhip_batch_grad", "fmhip_als_epoch",
    "fmhip_grad_floats", "fmhip_grad_bind", "fmhip_grad_ptr", "fmhip_grad_layout", "fmhip_step_compute",
    "fmhip_step_forward", "fmhip_step_backward", "fmhip_step_apply", "fmhip_step_stats",
    "fmhip_comm_unique_id", "fmhip_comm_create", "fmhip_comm_destroy", "fmhip_comm_info", "fmhip_comm_selftest",
    "fmhip_dp_exchange", "fmhip_dp_exchange_info", "fmhip_dp_plan", "fmhip_dp_plan_info",
    "fmhip_dp_step", "fmhip_dp_step_at", "fmhip_dp_steps", "fmhip_dp_epoch", "fmhip_dp_epoch_order", "fmhip_shard_rows",
    "fmhip_feature_counts", "fmhip_rank_from_counts", "fmhip_relabel_columns",
    "fmhip_feature_counts_gpu", "fmhip_rank_from_counts_gpu", "fmhip_relabel_columns_gpu",
)
# ... and include/fmhip_experimental.h — the measurement / experiment surface (tuning keys, profiling, emulation, layout
# queries, the two-pass forward on its own, a transport of the caller's own)
SYMBOLS_EXPERIMENTAL = (
    "fmhip_ablation_mask", "fmhip_tune", "fmhip_model_tune",
    "fmhip_profile_begin", "fmhip_profile_begin_rotating", "fmhip_profile_begin_sampled", "fmhip_profile_end",
    "fmhip_dataset_layout", "fmhip_dataset_hot_pages", "fmhip_dataset_band_plan", "fmhip_dataset_als_levels",
    "fmhip_dataset_partition_rows", "fmhip_step_forward_pass",
    "fmhip_comm_create_external", "fmhip_stream_wait", "fmhip_device_read", "fmhip_device_write",
    "fmhip_comm_profile_begin", "fmhip_comm_profile_end", "fmhip_comm_emulate", "fmhip_comm_emulate_load", "fmhip_comm_emulate_ranks",
)
# enum fmhip_tune_key (include/fmhip_experimental.h); TUNE maps the names without their prefix
(TUNE_FORWARD_KERNEL, TUNE_BACKWARD_KERNEL, TUNE_TILE_ROWS, TUNE_ROW_BLOCK, TUNE_XCD_PLACEMENT, TUNE_HOT_BLOCK, TUNE_FORWARD_OCCUPANCY,
 TUNE_ROW_ORDER, TUNE_FLAT_ADDRESS, TUNE_LAZY_DECAY, TUNE_FUSED_UPDATE, TUNE_MERGED_FINISH, TUNE_HOT_PAGES) = range(13)
TUNE = {name[5:]: value for name, value in list(globals().items()) if name.startswith("TUNE_")}
UNIQUE_ID_BYTES = 128


class Stats(C.Structure):
    _fields_ = [("sse", C.c_double), ("sum_e", C.c_double), ("rows", C.c_int64), ("nnz", C.c_int64),
                ("nonfinite", C.c_int64), ("steps", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Profile(C.Structure):
    _fields_ = [("ms", C.c_double * K_COUNT), ("launches", C.c_int64 * K_COUNT), ("nnz", C.c_int64 * K_COUNT),
                ("rows", C.c_int64 * K_COUNT), ("steps", C.c_int64 * K_COUNT)]

    def as_dict(self):
        return {KERNEL_NAMES[i]: dict(ms=self.ms[i], launches=self.launches[i], nnz=self.nnz[i], rows=self.rows[i], steps=self.steps[i])
                for i in range(K_COUNT)}


class DatasetOpts(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("hot_block", C.c_int32), ("batch_rows", C.c_int64), ("row_block_rows", C.c_int64)]


# int fn(void *ctx, void *device_buf, size_t count, int kind, void *hip_stream) — fmhip_comm_create_external
CollectiveFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
COLL_SUM_F32, COLL_MAX_I64, COLL_BCAST0_I64, COLL_ALLGATHER_I32, COLL_REDUCE_SCATTER_F32, COLL_ALLGATHER_F32 = 0, 1, 2, 3, 4, 5
EXCHANGE_DENSE, EXCHANGE_TOUCHED, EXCHANGE_SHARDED = 0, 1, 2
EXCHANGE_PIPELINED = 3
EXCHANGE_MODES = {"dense": EXCHANGE_DENSE, "touched": EXCHANGE_TOUCHED, "sharded": EXCHANGE_SHARDED, "pipelined": EXCHANGE_PIPELINED}


class CommProfile(C.Structure):
    _fields_ = [("exposed_ms", C.c_double), ("comm_ms", C.c_double), ("steps", C.c_int64), ("bytes", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class FmhipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("fmhip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load():
    """Loads libfmhip.so; raises if it has not been built (no fallback of any kind)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libfmhip.so not found at %s — build it first: python -c 'import __graft_entry__ as g; g.build()' "
            "(sparkfm_amd has no CPU fallback)" % LIB_PATH)
    try:
        # torch bundles its own HIP runtime (same soname); import it first so that one process
        # never ends up with two runtimes when torch tensors/streams are shared with the library
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    P = C.POINTER
    L.fmhip_version.restype = C.c_int
    L.fmhip_last_error.restype = C.c_char_p
    L.fmhip_device_count.argtypes = [P(C.c_int)]
    L.fmhip_tune.argtypes = [C.c_int, C.c_int]
    L.fmhip_model_create.argtypes = [C.c_int, i64, i32, vp, P(vp)]
    L.fmhip_model_destroy.argtypes = [vp]
    L.fmhip_model_tune.argtypes = [vp, C.c_int, C.c_int]
    L.fmhip_model_info.argtypes = [vp, P(i64), P(i32), P(i32)]
    L.fmhip_model_set_params.argtypes = [vp, dbl, vp, vp]
    L.fmhip_model_get_params.argtypes = [vp, P(dbl), vp, vp]
    L.fmhip_model_set_params_f32.argtypes = [vp, C.c_float, vp, vp]
    L.fmhip_model_get_params_f32.argtypes = [vp, P(C.c_float), vp, vp]
    L.fmhip_synchronize.argtypes = [vp]
    L.fmhip_dataset_create.argtypes = [C.c_int, i64, vp, vp, vp, vp, i64, P(vp)]
    L.fmhip_dataset_create_f32.argtypes = [C.c_int, i64, vp, vp, vp, vp, i64, P(vp)]
    L.fmhip_dataset_destroy.argtypes = [vp]
    L.fmhip_dataset_info.argtypes = [vp, P(i64), P(i64), P(i64), P(i64), P(i64)]
    L.fmhip_dataset_batch_info.argtypes = [vp, i64, P(i64), P(i64), P(i64), P(i64)]
    L.fmhip_dataset_get_transpose.argtypes = [vp, i64, vp, vp, vp, vp]
    L.fmhip_predict.argtypes = [vp, vp, vp]
    L.fmhip_rmse.argtypes = [vp, vp, P(dbl), P(Stats)]
    L.fmhip_residual.argtypes = [vp, vp, vp]
    L.fmhip_term_q.argtypes = [vp, vp, vp]
    L.fmhip_sgd_step.argtypes = [vp, vp, i64, dbl, dbl, dbl, dbl, P(Stats)]
    L.fmhip_sgd_epoch.argtypes = [vp, vp, dbl, dbl, dbl, dbl, vp, P(Stats)]
    L.fmhip_batch_grad.argtypes = [vp, vp, i64, vp, vp, P(dbl), P(Stats)]
    L.fmhip_als_epoch.argtypes = [vp, vp, dbl, dbl, dbl]
    L.fmhip_grad_floats.argtypes = [vp, P(i64)]
    L.fmhip_grad_bind.argtypes = [vp, vp]
    L.fmhip_grad_ptr.argtypes = [vp, P(vp)]
    L.fmhip_step_compute.argtypes = [vp, vp, i64]
    L.fmhip_step_forward.argtypes = [vp, vp, i64]
    L.fmhip_step_backward.argtypes = [vp, vp, i64, i64, i64, C.c_int]
    L.fmhip_grad_layout.argtypes = [vp, P(i64), P(i64)]
    L.fmhip_step_apply.argtypes = [vp, dbl, dbl, dbl, dbl]
    L.fmhip_step_stats.argtypes = [vp, P(Stats)]
    L.fmhip_profile_begin.argtypes = [vp]
    L.fmhip_profile_begin_rotating.argtypes = [vp]
    L.fmhip_profile_begin_sampled.argtypes = [vp, C.c_int]
    L.fmhip_profile_end.argtypes = [vp, P(Profile)]
    L.fmhip_dataset_create_opts.argtypes = [C.c_int, i64, vp, vp, vp, vp, P(DatasetOpts), P(vp)]
    L.fmhip_dataset_layout.argtypes = [vp, P(i32), vp, P(i64)]
    L.fmhip_model_get_rows.argtypes = [vp, i64, vp, vp, vp]
    L.fmhip_model_init_normal.argtypes = [vp, C.c_uint64, dbl, dbl]
    L.fmhip_rows_create.argtypes = [C.c_int, i64, vp, vp, vp, vp, P(vp)]
    L.fmhip_rows_create_f32.argtypes = [C.c_int, i64, vp, vp, vp, vp, P(vp)]
    L.fmhip_predict_rows.argtypes = [vp, i64, vp, vp, vp, vp]
    L.fmhip_comm_unique_id.argtypes = [vp]
    L.fmhip_comm_create.argtypes = [vp, vp, C.c_int, C.c_int, P(vp)]
    L.fmhip_comm_destroy.argtypes = [vp]
    L.fmhip_comm_info.argtypes = [vp, P(C.c_int), P(C.c_int)]
    L.fmhip_comm_selftest.argtypes = [vp, P(C.c_int)]
    L.fmhip_dataset_partition_rows.argtypes = [vp, i64]
    L.fmhip_step_forward_pass.argtypes = [vp, vp, i64, C.c_int]
    L.fmhip_dp_steps.argtypes = [vp, vp, vp, i64, vp, dbl, dbl, dbl, dbl]
    L.fmhip_dp_plan.argtypes = [vp, vp, vp, C.c_int, vp, vp]
    L.fmhip_dp_step.argtypes = [vp, vp, i64, vp, dbl, dbl, dbl, dbl]
    L.fmhip_dp_epoch.argtypes = [vp, vp, vp, dbl, dbl, dbl, dbl, P(Stats)]
    L.fmhip_dp_step_at.argtypes = [vp, vp, i64, vp, dbl, dbl, dbl, dbl]
    L.fmhip_dp_epoch_order.argtypes = [vp, vp, vp, dbl, dbl, dbl, dbl, vp, i64, P(Stats)]
    L.fmhip_dp_plan_info.argtypes = [vp, P(i64), P(i64)]
    L.fmhip_dataset_als_levels.argtypes = [vp, P(i64), P(i64), P(i64)]
    L.fmhip_comm_emulate.argtypes = [vp, dbl]
    L.fmhip_comm_emulate_ranks.argtypes = [vp, C.c_int]
    L.fmhip_comm_emulate_load.argtypes = [vp, C.c_int]
    L.fmhip_comm_profile_begin.argtypes = [vp]
    L.fmhip_comm_profile_end.argtypes = [vp, P(CommProfile)]
    L.fmhip_shard_rows.argtypes = [i64, vp, C.c_int, C.c_int, P(i64), P(i64)]
    L.fmhip_comm_create_external.argtypes = [vp, C.c_int, C.c_int, CollectiveFn, vp, P(vp)]
    L.fmhip_dp_exchange.argtypes = [vp, C.c_int]
    L.fmhip_dp_exchange_info.argtypes = [vp, P(C.c_int), P(i64), P(C.c_double)]
    L.fmhip_stream_wait.argtypes = [vp]
    L.fmhip_device_read.argtypes = [vp, vp, C.c_size_t, vp]
    L.fmhip_device_write.argtypes = [vp, vp, C.c_size_t, vp]
    L.fmhip_dataset_hot_pages.argtypes = [vp, P(C.c_int32), P(C.c_int32), vp, P(i64)]
    L.fmhip_dataset_band_plan.argtypes = [vp, P(i64), P(i64), P(i64)]
    L.fmhip_feature_counts.argtypes = [i64, vp, i64, vp]
    L.fmhip_rank_from_counts.argtypes = [i64, vp, vp, vp]
    L.fmhip_relabel_columns.argtypes = [i64, vp, i64, vp, vp]
    L.fmhip_feature_counts_gpu.argtypes = [C.c_int, i64, vp, i64, vp]
    L.fmhip_rank_from_counts_gpu.argtypes = [C.c_int, i64, vp, vp, vp]
    L.fmhip_relabel_columns_gpu.argtypes = [C.c_int, i64, vp, i64, vp, vp]
    for name in SYMBOLS + SYMBOLS_EXPERIMENTAL:
        fn = getattr(L, name)
        if name not in ("fmhip_version", "fmhip_last_error"):
            fn.restype = C.c_int
    # FMHIP_TUNE="key=value,key=value": experiment knobs (fmhip_tune) applied at load time; key = a number or a name of
    # enum fmhip_tune_key without its prefix (flat_address=1)
    for item in filter(None, os.environ.get("FMHIP_TUNE", "").split(",")):
        k, v = item.split("=")
        L.fmhip_tune(int(k) if k.strip().isdigit() else TUNE[k.strip().upper()], int(v))
    if L.fmhip_ablation_mask() != 0 and not os.environ.get("FMHIP_LIB"):
        raise ImportError("libfmhip.so was built with a timing-only kernel ablation (mask %d): rebuild it without FMHIP_EXP_* flags" % L.fmhip_ablation_mask())
    _lib = L
    return L


def check(code):
    if code != OK:
        raise FmhipError(code, load().fmhip_last_error().decode("utf-8", "replace"))
    return code


def ptr(a):
    """numpy array (or None) -> void* for the C ABI."""
    return None if a is None else a.ctypes.data_as(C.c_void_p)
