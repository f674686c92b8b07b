"""numpy restatement of the element-wise layers either side of the PQ linears.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the
product package (vllm_pq_amd) never imports this module.  Pinned against tests/golden/layers_golden.npz, the
outputs of the reference's own forward_native methods (tests/test_layers.py).

fp16 / bf16 arrays are numpy uint16 bit patterns; ``dtype`` is "fp16" or "bf16".
  silu_and_mul(x, dtype)                         SiluAndMul.forward_native, vllm/model_executor/layers/activation.py:73-76
  rms_norm(x, weight, eps, dtype, residual)      RMSNorm.forward_native, vllm/model_executor/layers/layernorm.py:38-74
"""
import numpy as np

from .pq_oracle import from_f32, to_f32


def silu_and_mul(x: np.ndarray, dtype: str) -> np.ndarray:
    """x [T, 2I] -> [T, I].  F.silu returns the act dtype (one rounding of x / (1 + exp(-x)), evaluated
    here in fp64 so that the only error is that rounding); the product is rounded once more."""
    I = x.shape[-1] // 2
    g = to_f32(x[..., :I], dtype).astype(np.float64)
    u = to_f32(x[..., I:], dtype)
    with np.errstate(over="ignore"):
        s = (g / (1.0 + np.exp(-g))).astype(np.float32)
    s = to_f32(from_f32(s, dtype), dtype)
    return from_f32(s * u, dtype)


def rms_norm(x: np.ndarray, weight: np.ndarray, eps: float, dtype: str, residual: np.ndarray = None):
    """-> y, or (y, residual_out) when residual is given.  The sum and the statistics are fp32 (layernorm.py:45-48,
    :65-67: the residual written back is the rounded sum, the variance uses the unrounded one); x * rsqrt is
    rounded to the act dtype before the multiplication by the weight (:68-70)."""
    z = to_f32(x, dtype)
    res_out = None
    if residual is not None:
        z = z + to_f32(residual, dtype)
        res_out = from_f32(z, dtype)
    var = (z.astype(np.float64) ** 2).mean(axis=-1, keepdims=True).astype(np.float32)
    rs = (np.float32(1.0) / np.sqrt(var + np.float32(eps))).astype(np.float32)
    n = to_f32(from_f32(z * rs, dtype), dtype)
    y = from_f32(n * to_f32(weight, dtype), dtype)
    return y if residual is None else (y, res_out)


def ulp_distance(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Distance of two act-dtype bit patterns in units in the last place (sign-magnitude order)."""
    def key(v):
        v = v.astype(np.int32)
        return np.where(v & 0x8000, -(v & 0x7FFF), v & 0x7FFF)
    return np.abs(key(a) - key(b))
