"""ctypes binding of libpq_amd.so (C ABI: include/pq_amd.h).

The library is the only compute path of this package: if it cannot be loaded, or a tensor is not
on a ROCm device, the ops raise -- there is no CPU or PyTorch fallback.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PQ_AMD_LIB", os.path.join(_HERE, "libpq_amd.so"))

PQ_ABI_VERSION = 1
PQ_MAX_PARTS = 8
PQ_DTYPE_FP16, PQ_DTYPE_BF16 = 0, 1

# every symbol include/pq_amd.h declares
EXPORTS = ("pq_version", "pq_caps", "pq_last_error", "pq_dequant", "pq_gemm_workspace_bytes", "pq_gemm",
           "pq_packed_codes_bytes", "pq_repack_codes", "pq_gemm_packed", "pq_gemm_kernel_name", "pq_silu_and_mul",
           "pq_rms_norm", "pq_fused_add_rms_norm", "pq_gemm_silu_mul_is_fused", "pq_gemm_packed_silu_mul", "pq_embedding",
           "pq_debug_reload_knobs", "pq_gemm_workspace_max_bytes", "pq_gemm_packed_noalloc", "pq_ar_last_error", "pq_ar_buffer_bytes",
           "pq_ar_max_message_bytes", "pq_ar_alloc", "pq_ar_free", "pq_ar_open", "pq_ar_close", "pq_ar_create",
           "pq_ar_destroy", "pq_ar_all_reduce", "pq_ar_status", "pq_gemm_packed_add_norm_supported",
           "pq_gemm_packed_normed_supported", "pq_gemm_packed_add_norm", "pq_gemm_packed_normed",
           "pq_gemm_packed_normed_silu_mul_supported", "pq_gemm_packed_normed_silu_mul", "pq_ar_add_norm_supported",
           "pq_ar_all_reduce_add_norm", "pq_ar_set_timeout_ms")

_lock = threading.Lock()
_lib = None


class PQLibraryError(RuntimeError):
    pass


def load():
    """Load libpq_amd.so once; raises PQLibraryError when it is missing or has the wrong ABI."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise PQLibraryError(
                f"{LIB_PATH} not found: build it with `make -C vllm_pq_amd/csrc` (or __graft_entry__.build()); "
                "vllm_pq_amd has no fallback path")
        try:
            L = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise PQLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        missing = [s for s in EXPORTS if not hasattr(L, s)]
        if missing:
            raise PQLibraryError(f"{LIB_PATH} lacks symbols {missing}")
        vp, i32, i64, sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t
        pi32 = ctypes.POINTER(ctypes.c_int32)
        L.pq_version.argtypes = []
        L.pq_version.restype = ctypes.c_int
        L.pq_caps.argtypes = []
        L.pq_caps.restype = ctypes.c_uint64
        L.pq_last_error.argtypes = []
        L.pq_last_error.restype = ctypes.c_char_p
        L.pq_dequant.argtypes = [vp, vp, vp, pi32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]
        L.pq_dequant.restype = ctypes.c_int
        L.pq_embedding.argtypes = [vp, i32, vp, vp, vp, pi32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i64, vp]
        L.pq_embedding.restype = ctypes.c_int
        L.pq_gemm_workspace_bytes.argtypes = [i32] * 8
        L.pq_gemm_workspace_bytes.restype = sz
        L.pq_gemm_workspace_max_bytes.argtypes = []
        L.pq_gemm_workspace_max_bytes.restype = sz
        L.pq_gemm.argtypes = [vp, i32, i64, vp, vp, vp, vp, pi32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i64,
                              vp, sz, vp]
        L.pq_gemm.restype = ctypes.c_int
        L.pq_gemm_packed.argtypes = L.pq_gemm.argtypes
        L.pq_gemm_packed.restype = ctypes.c_int
        L.pq_gemm_packed_noalloc.argtypes = L.pq_gemm.argtypes[:18] + [vp]
        L.pq_gemm_packed_noalloc.restype = ctypes.c_int
        L.pq_packed_codes_bytes.argtypes = [pi32, i32, i32, i32, i32, i32, i32, i32]
        L.pq_packed_codes_bytes.restype = sz
        L.pq_repack_codes.argtypes = [vp, pi32, i32, i32, i32, i32, i32, i32, i32, vp, vp]
        L.pq_repack_codes.restype = ctypes.c_int
        L.pq_gemm_kernel_name.argtypes = [i32] * 8 + [pi32, i32, i32, ctypes.c_char_p, sz]
        L.pq_gemm_kernel_name.restype = ctypes.c_int
        f32 = ctypes.c_float
        L.pq_silu_and_mul.argtypes = [vp, i64, i32, i32, i32, vp, i64, vp]
        L.pq_silu_and_mul.restype = ctypes.c_int
        L.pq_rms_norm.argtypes = [vp, i64, vp, f32, i32, i32, i32, vp, i64, vp]
        L.pq_rms_norm.restype = ctypes.c_int
        L.pq_fused_add_rms_norm.argtypes = [vp, i64, vp, i64, vp, f32, i32, i32, i32, vp]
        L.pq_fused_add_rms_norm.restype = ctypes.c_int
        L.pq_gemm_silu_mul_is_fused.argtypes = [i32] * 7
        L.pq_gemm_silu_mul_is_fused.restype = ctypes.c_int
        L.pq_gemm_packed_silu_mul.argtypes = [vp, i32, i64, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, i64, vp,
                                              vp, sz, vp]
        L.pq_gemm_packed_silu_mul.restype = ctypes.c_int
        for fn in (L.pq_gemm_packed_add_norm_supported, L.pq_gemm_packed_normed_supported):
            fn.argtypes = [i32, pi32, i32, i32, i32, i32, i32, i32, i32, i32]
            fn.restype = ctypes.c_int
        L.pq_gemm_packed_add_norm.argtypes = [vp, i32, i64, vp, vp, vp, vp, pi32, i32, i32, i32, i32, i32, i32, i32, i32,
                                              vp, i64, vp, vp, i64, vp, vp, sz, vp]
        L.pq_gemm_packed_add_norm.restype = ctypes.c_int
        L.pq_gemm_packed_normed.argtypes = [vp, i32, i64, vp, vp, vp, vp, pi32, i32, i32, i32, i32, i32, i32, i32, i32,
                                            vp, f32, vp, i64, vp, sz, vp]
        L.pq_gemm_packed_normed.restype = ctypes.c_int
        L.pq_gemm_packed_normed_silu_mul_supported.argtypes = [i32] * 7
        L.pq_gemm_packed_normed_silu_mul_supported.restype = ctypes.c_int
        L.pq_gemm_packed_normed_silu_mul.argtypes = [vp, i32, i64, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp,
                                                     ctypes.c_float, vp, i64, vp]
        L.pq_gemm_packed_normed_silu_mul.restype = ctypes.c_int
        L.pq_debug_reload_knobs.argtypes = []
        L.pq_debug_reload_knobs.restype = None
        pvp = ctypes.POINTER(vp)
        L.pq_ar_last_error.argtypes = []
        L.pq_ar_last_error.restype = ctypes.c_char_p
        L.pq_ar_buffer_bytes.argtypes = []
        L.pq_ar_buffer_bytes.restype = sz
        L.pq_ar_max_message_bytes.argtypes = []
        L.pq_ar_max_message_bytes.restype = sz
        L.pq_ar_alloc.argtypes = [pvp, ctypes.c_char_p]
        L.pq_ar_free.argtypes = [vp]
        L.pq_ar_open.argtypes = [ctypes.c_char_p, pvp]
        L.pq_ar_close.argtypes = [vp]
        L.pq_ar_create.argtypes = [i32, i32, pvp, pvp]
        L.pq_ar_destroy.argtypes = [vp]
        L.pq_ar_all_reduce.argtypes = [vp, vp, vp, i64, i32, vp]
        L.pq_ar_status.argtypes = [vp]
        L.pq_ar_add_norm_supported.argtypes = [i32, i32, i32]
        L.pq_ar_all_reduce_add_norm.argtypes = [vp, vp, i32, i32, i32, vp, vp, f32, vp, vp]
        L.pq_ar_set_timeout_ms.argtypes = [vp, i64]
        for fn in ("pq_ar_add_norm_supported", "pq_ar_all_reduce_add_norm", "pq_ar_set_timeout_ms", "pq_ar_alloc", "pq_ar_free", "pq_ar_open", "pq_ar_close", "pq_ar_create", "pq_ar_destroy",
                   "pq_ar_all_reduce", "pq_ar_status"):
            getattr(L, fn).restype = ctypes.c_int
        if L.pq_version() != PQ_ABI_VERSION:
            raise PQLibraryError(f"{LIB_PATH}: ABI version {L.pq_version()} != {PQ_ABI_VERSION}")
        _lib = L
    return _lib


def reload_knobs() -> None:
    """Diagnostics: have the library read the PQ_AMD_* dispatch knobs from the environment again."""
    load().pq_debug_reload_knobs()


def last_error() -> str:
    return load().pq_last_error().decode("utf-8", "replace")


_PARTS_ARRAYS = {}


def parts_array(sizes):
    """ctypes int32 array of the partition sizes (read-only for the library: one shared array per distinct list)."""
    key = tuple(sizes)
    arr = _PARTS_ARRAYS.get(key)
    if arr is None:
        arr = _PARTS_ARRAYS[key] = (ctypes.c_int32 * len(key))(*[int(s) for s in key])
    return arr
