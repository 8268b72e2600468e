"""ctypes binding of libmdd_hip.so (include/mdd_hip.h).

There is no CPU fallback: if the library is missing or no gfx950 device is present the
product path raises.  (The CPU oracle under oracle/ is test infrastructure and is never
imported from here.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MDD_LIB_PATH") or os.path.join(_HERE, "libmdd_hip.so")   # (MDD_LIB_PATH: a diagnostic build of the same library, tests only)

MDD_ERR_EMPTY = -5   # include/mdd_hip.h: an empty sequence where the reference raises TypeError

EXPORTS = (
    "mdd_last_error", "mdd_version", "mdd_create", "mdd_destroy", "mdd_load_weight", "mdd_finalize_weights",
    "mdd_set_precision", "mdd_get_precision", "mdd_stack_len", "mdd_stack_skip", "mdd_len_frames", "mdd_forward", "mdd_forward_fused", "mdd_forward_raw", "mdd_forward_num_stages", "mdd_forward_profile", "mdd_tap", "mdd_tap_copy", "mdd_enable_taps", "mdd_sync",
    "mdd_greedy", "mdd_beam", "mdd_ctc_loss", "mdd_ctc_workspace_bytes", "mdd_align", "mdd_align_batch", "mdd_eval_batch", "mdd_fbank_num_frames", "mdd_fbank",
    "mdd_train_create", "mdd_train_destroy", "mdd_train_num_tensors", "mdd_train_tensor_info", "mdd_train_num_masks", "mdd_train_mask_bytes",
    "mdd_train_forward", "mdd_train_backward", "mdd_train_sync", "mdd_train_set_precision", "mdd_adam_step",
    "mdd_diag_gemm_ph8", "mdd_diag_gates", "mdd_diag_gemm", "mdd_diag_gemm_time",
)


class MddConfig(C.Structure):
    _fields_ = [("feat", C.c_int32), ("hidden", C.c_int32), ("layers", C.c_int32), ("num_class", C.c_int32),
                ("channels", C.c_int32), ("emb_rows", C.c_int32), ("emb_dim", C.c_int32), ("bn_eps", C.c_float)]


class MddError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libmdd_hip.so once; fail loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MddError("libmdd_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
                       "g.build()'` or `make -C ctc-attention-mispronunciation_amd/csrc` (no CPU fallback exists)" % LIB_PATH)
    # torch first: the process must end up with ONE HIP runtime (the one torch ships), otherwise torch streams
    # and device pointers would belong to a different runtime than the one libmdd_hip.so binds to.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, i32, i64p, f32p = C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_float)
    L.mdd_last_error.restype = C.c_char_p
    L.mdd_create.argtypes = [C.POINTER(MddConfig), C.c_int, C.POINTER(vp)]
    L.mdd_destroy.argtypes = [vp]
    L.mdd_destroy.restype = None
    L.mdd_load_weight.argtypes = [vp, C.c_char_p, vp, i64p, i32]
    L.mdd_finalize_weights.argtypes = [vp]
    L.mdd_set_precision.argtypes = [vp, i32]
    L.mdd_get_precision.argtypes = [vp]
    L.mdd_stack_len.argtypes = [i32, i32, i32]
    L.mdd_stack_len.restype = i32
    L.mdd_stack_skip.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, vp]
    L.mdd_len_frames.argtypes = [i32, i32, i32]
    L.mdd_len_frames.restype = i32
    L.mdd_forward.argtypes = [vp, vp, i32, i32, vp, i32, vp, vp]
    L.mdd_forward_raw.argtypes = [vp, vp, i32, i32, vp, i32, vp, vp]
    L.mdd_forward_fused.argtypes = [vp, vp, i32, i32, vp, i32, vp, vp, vp, vp]
    L.mdd_forward_num_stages.argtypes = [vp]
    L.mdd_forward_profile.argtypes = [vp, vp, i32, i32, vp, i32, vp, vp, C.c_char_p, i32, vp, vp, vp, i32]
    L.mdd_tap.argtypes = [vp, C.c_char_p, i64p]
    L.mdd_tap.restype = vp
    L.mdd_tap_copy.argtypes = [vp, C.c_char_p, vp, C.c_int64, vp]
    L.mdd_enable_taps.argtypes = [vp, i32]
    L.mdd_sync.argtypes = [vp, vp]
    L.mdd_greedy.argtypes = [vp, i32, i32, i32, vp, i32, vp, vp, vp]
    L.mdd_beam.argtypes = [vp, i32, i32, i32, vp, i32, i32, vp, C.c_double, vp, vp, vp, vp, vp]
    L.mdd_ctc_loss.argtypes = [vp, i32, i32, i32, vp, i32, vp, vp, i32, vp, vp, vp, C.c_int64, vp]
    L.mdd_ctc_workspace_bytes.argtypes = [i32, i32, i32, i32, i32]
    L.mdd_ctc_workspace_bytes.restype = C.c_int64
    L.mdd_align.argtypes = [vp, i32, vp, i32, C.POINTER(i32), vp, C.POINTER(i32)]
    L.mdd_align_batch.argtypes = [vp, vp, i32, vp, vp, i32, i32, vp, vp, i32, vp]
    L.mdd_eval_batch.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, vp]
    L.mdd_fbank_num_frames.argtypes = [C.c_int64]
    L.mdd_fbank_num_frames.restype = i32
    L.mdd_fbank.argtypes = [vp, C.c_int64, vp, vp, vp, vp]
    L.mdd_train_create.argtypes = [C.POINTER(MddConfig), C.c_int, C.POINTER(vp)]
    L.mdd_train_destroy.argtypes = [vp]
    L.mdd_train_destroy.restype = None
    L.mdd_train_num_tensors.argtypes = [vp]
    L.mdd_train_tensor_info.argtypes = [vp, i32, C.c_char_p, i32, i64p, C.POINTER(i32)]
    L.mdd_train_num_masks.argtypes = [vp]
    L.mdd_train_mask_bytes.argtypes = [vp, i32, i32, i32]
    L.mdd_train_mask_bytes.restype = C.c_int64
    L.mdd_train_forward.argtypes = [vp, vp, vp, i32, i32, vp, i32, vp, C.c_uint64, C.c_float, vp, vp]
    L.mdd_train_backward.argtypes = [vp, vp, vp, vp, vp]
    L.mdd_train_sync.argtypes = [vp, vp]
    L.mdd_train_set_precision.argtypes = [vp, C.c_int32]
    L.mdd_adam_step.argtypes = [vp, vp, vp, vp, vp, i32, i32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, vp]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise MddError("libmdd_hip: %s (status %d)" % (lib().mdd_last_error().decode(), rc))


def current_stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise MddError("no HIP device visible: the MI355X path has no CPU fallback")
