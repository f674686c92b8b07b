"""numpy/ctypes front-end of the CPU oracle (oracle/pq_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py.  The product package (vllm_pq_amd) never imports this module.

fp16 / bf16 arrays are numpy uint16 bit patterns; ``dtype`` is "fp16" or "bf16".
The functions mirror vllm/model_executor/layers/quantization/aqlm.py:
  unpack(codes, nbits)                       aqlm.py:32-34
  dequant(codes, codebooks, scales, parts)   aqlm.py:37-74 applied per partition as in :105-126
  gemm(x, codes, codebooks, scales, parts, bias, order)   aqlm.py:77-127 (order 0), :132-156 (order 1)
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpq_oracle.so")
_DT = {"fp16": 0, "bf16": 1}
_lib = None


def build(force: bool = False) -> str:
    """Compile oracle/pq_oracle.c with gcc (a few seconds)."""
    src = os.path.join(_HERE, "pq_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpq_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        L.pq_oracle_unpack.argtypes = [vp, i32, i64, i32, vp]
        L.pq_oracle_unpack.restype = i32
        L.pq_oracle_dequant.argtypes = [vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp]
        L.pq_oracle_dequant.restype = i32
        L.pq_oracle_gemm.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32,
                                     vp, vp]
        L.pq_oracle_gemm.restype = i32
        L.pq_oracle_act_to_f32.argtypes = [ctypes.c_uint16, i32]
        L.pq_oracle_act_to_f32.restype = ctypes.c_float
        L.pq_oracle_f32_to_act.argtypes = [ctypes.c_float, i32]
        L.pq_oracle_f32_to_act.restype = ctypes.c_uint16
        L.pq_oracle_set_bf16_sum_rounding.argtypes = [i32]
        L.pq_oracle_set_bf16_sum_rounding.restype = None
        _lib = L
    return _lib


def set_bf16_sum_rounding(mode: str) -> None:
    """"rne" (default; what the HIP kernels do) or "torch_cpu" (add 0x8000 and truncate: the
    rounding torch's CPU embedding_bag applies to the fp32 sum over M >= 2 bf16 codebooks)."""
    lib().pq_oracle_set_bf16_sum_rounding({"rne": 0, "torch_cpu": 1}[mode])


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _codes(codes):
    codes = np.ascontiguousarray(codes)
    assert codes.dtype in (np.int8, np.int16, np.int32), codes.dtype
    assert codes.ndim == 3
    return codes


def _u16(a):
    if a is None:
        return None
    a = np.ascontiguousarray(a)
    assert a.dtype == np.uint16, a.dtype
    return a


def unpack(codes: np.ndarray, nbits: int) -> np.ndarray:
    codes = np.ascontiguousarray(codes)
    out = np.empty(codes.shape, np.int64)
    rc = lib().pq_oracle_unpack(_p(codes), codes.dtype.itemsize, codes.size, nbits, _p(out))
    if rc:
        raise ValueError(f"pq_oracle_unpack rc={rc}")
    return out


def dequant(codes, codebooks, scales, parts, dtype: str) -> np.ndarray:
    """-> uint16 [N, G*d]: the reference's dequantised weight (scaled iff scales is not None)."""
    codes = _codes(codes)
    codebooks = _u16(codebooks)
    scales = _u16(scales)
    N, G, M = codes.shape
    PM, K, d = codebooks.shape[0], codebooks.shape[1], codebooks.shape[-1]
    P = len(parts)
    assert PM == P * M, (PM, P, M)
    parr = np.asarray(parts, np.int32)
    W = np.empty((N, G * d), np.uint16)
    rc = lib().pq_oracle_dequant(_p(codes), codes.dtype.itemsize, _p(codebooks), _p(scales), _p(parr), P, N, G, M, K,
                                 d, _DT[dtype], _p(W), None)
    if rc:
        raise ValueError(f"pq_oracle_dequant rc={rc}")
    return W


def gemm(x, codes, codebooks, scales, parts, bias, dtype: str, order: int = 0, acc_double: bool = True,
         want_f32: bool = False):
    """-> uint16 [T, N] (and the un-rounded fp32 result when want_f32)."""
    codes = _codes(codes)
    x = _u16(x)
    codebooks = _u16(codebooks)
    scales = _u16(scales)
    bias = _u16(bias)
    N, G, M = codes.shape
    PM, K, d = codebooks.shape[0], codebooks.shape[1], codebooks.shape[-1]
    P = len(parts)
    assert PM == P * M
    T = x.shape[0]
    assert x.shape[1] == G * d
    parr = np.asarray(parts, np.int32)
    y = np.empty((T, N), np.uint16)
    y32 = np.empty((T, N), np.float32) if want_f32 else None
    rc = lib().pq_oracle_gemm(_p(x), T, _p(codes), codes.dtype.itemsize, _p(codebooks), _p(scales), _p(bias),
                              _p(parr), P, N, G, M, K, d, _DT[dtype], order, int(acc_double), _p(y), _p(y32))
    if rc:
        raise ValueError(f"pq_oracle_gemm rc={rc}")
    return (y, y32) if want_f32 else y


# ---- helpers shared by the tests ---------------------------------------------------------
def to_f32(bits: np.ndarray, dtype: str) -> np.ndarray:
    bits = np.asarray(bits, np.uint16)
    if dtype == "bf16":
        return (bits.astype(np.uint32) << 16).view(np.float32)
    return bits.view(np.float16).astype(np.float32)


def from_f32(a: np.ndarray, dtype: str) -> np.ndarray:
    a = np.ascontiguousarray(a, np.float32)
    if dtype == "fp16":
        return a.astype(np.float16).view(np.uint16)
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    nan = np.isnan(a)
    if nan.any():
        r[nan] = ((a.view(np.uint32)[nan] >> 16) | 0x40).astype(np.uint16)
    return r
