"""Helpers shared by the parity tests: seeded synthetic problems, numpy<->torch bit views, tolerances."""
import numpy as np
import torch

from oracle import pq_oracle as orc

TORCH_DT = {"fp16": torch.float16, "bf16": torch.bfloat16}
NP_CODE = {"int8": np.int8, "int16": np.int16}


def bits_of(t: torch.Tensor) -> np.ndarray:
    """fp16/bf16 tensor (any device) -> uint16 bit patterns."""
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def tensor_of(bits: np.ndarray, dtype: str, device="cpu") -> torch.Tensor:
    t = torch.from_numpy(np.ascontiguousarray(bits).view(np.int16).copy()).view(TORCH_DT[dtype])
    return t.to(device)


def get_int_dtype(nbits):
    return torch.int8 if nbits <= 8 else torch.int16


class Problem:
    """One synthetic PQ linear problem (codes/codebooks/scales/x/bias), generated as the reference's
    own benchmark draws them (benchmarks/kernels/benchmark_aqlm.py:264-279) but seeded."""

    def __init__(self, seed, K, d, M, parts, Kin, T, dtype="bf16", bias=False, scales="uniform"):
        g = torch.Generator().manual_seed(seed)
        self.K, self.d, self.M, self.parts, self.Kin, self.T, self.dtype = K, d, M, list(parts), Kin, T, dtype
        self.N = sum(parts)
        self.G = Kin // d
        nbits = K.bit_length() - 1
        idt = get_int_dtype(nbits)
        info = torch.iinfo(idt)
        dt = TORCH_DT[dtype]
        self.codes = torch.randint(info.min, info.max + 1, (self.N, self.G, M), generator=g, dtype=torch.int64).to(idt)
        self.codebooks = torch.randn(len(parts) * M, K, 1, d, generator=g).to(dt)
        if scales == "normal":
            self.scales = torch.randn(self.N, 1, 1, 1, generator=g).to(dt)
        else:
            self.scales = (torch.rand(self.N, 1, 1, 1, generator=g) + 0.5).to(dt)
        self.x = torch.randn(T, Kin, generator=g).to(dt)
        self.bias = torch.randn(self.N, generator=g).to(dt) if bias else None

    # ---- oracle side (numpy) ----
    def np_args(self):
        return dict(codes=self.codes.numpy(), codebooks=bits_of(self.codebooks).reshape(-1, self.K, self.d),
                    scales=bits_of(self.scales).reshape(-1), x=bits_of(self.x),
                    bias=None if self.bias is None else bits_of(self.bias))

    def oracle_gemm(self, order=2, want_f32=True):
        a = self.np_args()
        return orc.gemm(a["x"], a["codes"], a["codebooks"], a["scales"], self.parts, a["bias"], self.dtype,
                        order=order, acc_double=True, want_f32=want_f32)

    def oracle_dequant(self, scaled=False):
        a = self.np_args()
        return orc.dequant(a["codes"], a["codebooks"], a["scales"] if scaled else None, self.parts, self.dtype)

    # ---- device side ----
    def to(self, device):
        for n in ("codes", "codebooks", "scales", "x", "bias"):
            v = getattr(self, n)
            if v is not None:
                setattr(self, n + "_dev", v.to(device))
            else:
                setattr(self, n + "_dev", None)
        return self


def ulp(dtype: str) -> float:
    """Spacing of the act dtype relative to a value in [1, 2)."""
    return 2.0 ** -7 if dtype == "bf16" else 2.0 ** -10


def assert_close_to_exact(y_bits: np.ndarray, y32: np.ndarray, dtype: str, what=""):
    """Tolerance of the fused GEMM against the exact-arithmetic value y32 = s * (x . W) + b
    (fp64-accumulated oracle, order 2): one rounding to the act dtype (half an ulp) plus fp32
    accumulation noise, stated as  |y - y32| <= 0.51 ulp * |y32| + 2e-5 * rms(y32) + tiny."""
    y = orc.to_f32(y_bits, dtype).astype(np.float64)
    ref = y32.astype(np.float64)
    rms = float(np.sqrt(np.mean(ref * ref))) if ref.size else 0.0
    tol = 0.51 * ulp(dtype) * np.abs(ref) + 2e-5 * rms + 1e-30
    if dtype == "fp16":
        tol = tol + 2.0 ** -25   # subnormal spacing
    bad = np.abs(y - ref) > tol
    assert not bad.any(), (f"{what}: {bad.sum()} of {bad.size} outside tolerance; worst "
                           f"{np.abs(y - ref)[bad].max():.4g} vs tol {tol[bad].min():.4g}")


def assert_close_to_reference_order(y_bits: np.ndarray, ref_bits: np.ndarray, dtype: str, what=""):
    """Tolerance against a value computed in one of the reference's own orders (weight rounded to the
    act dtype after scaling, aqlm.py:70-72): the three reference orders differ from each other by
    this much.  mean|y - ref| / mean|ref| < 2 ulp, and no element off by more than 8 ulp of
    max(|ref|, rms)."""
    y = orc.to_f32(y_bits, dtype).astype(np.float64)
    ref = orc.to_f32(ref_bits, dtype).astype(np.float64)
    rms = float(np.sqrt(np.mean(ref * ref))) + 1e-30
    mean_rel = np.abs(y - ref).mean() / (np.abs(ref).mean() + 1e-30)
    assert mean_rel < 2 * ulp(dtype), f"{what}: mean relative error {mean_rel:.4g}"
    worst = (np.abs(y - ref) / np.maximum(np.abs(ref), rms)).max()
    assert worst < 8 * ulp(dtype), f"{what}: worst element {worst:.4g}"
