"""The element-wise layers either side of the PQ linears (SiluAndMul, RMSNorm): the oracle against golden
vectors from the reference's own forward_native methods, the host mirror's interface, and (gpu) the HIP
kernels through the C ABI against the oracle.

Tolerances.  Both are a handful of fp32 operations followed by roundings to the act dtype.  Two correct
implementations can disagree where an fp32 intermediate sits within an fp32 rounding error of an act-dtype
rounding boundary (exp / rsqrt are not correctly rounded on either side, sums run in different orders), and
then differ by one unit in the last place of that rounding; the multiplication that follows (by the up
projection, by the norm weight) can turn it into two units of the product's last place.
So: distance <= 2 ulp everywhere and at least 99% of the elements identical; the residual sum, a single
exact fp32 addition rounded once, is bit-exact."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import layers_oracle as lo
from pq_testlib import bits_of, tensor_of

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "layers_golden.json")) as f:
    CASES = json.load(f)["cases"]
ARR = np.load(os.path.join(HERE, "golden", "layers_golden.npz"), allow_pickle=False)
DEV = "cuda:0"


def _close(got, want, max_ulp, min_equal, what):
    d = lo.ulp_distance(got, want)
    # NaN payloads and the two zeros compare by value
    both_nan = np.isnan(_f(got, what)) & np.isnan(_f(want, what))
    d = np.where(both_nan, 0, d)
    assert d.max(initial=0) <= max_ulp, f"{what}: {int(d.max())} ulp"
    assert (d == 0).mean() >= min_equal, f"{what}: only {(d == 0).mean():.4f} identical"


def _f(b, what):
    return lo.to_f32(b, "bf16" if "bf16" in what else "fp16")


def _id(c):
    return f"{c['id']}-{c['dtype']}-T{c['T']}"


SILU = [c for c in CASES if c["kind"] == "silu_and_mul"]
NORM = [c for c in CASES if c["kind"] == "rms_norm"]


@pytest.mark.parametrize("c", SILU, ids=_id)
def test_oracle_silu_and_mul_matches_reference(c):
    y = lo.silu_and_mul(ARR[f"{c['id']}_x"], c["dtype"])
    _close(y, ARR[f"{c['id']}_y"], 2, 0.99, f"silu {c['dtype']}")


@pytest.mark.parametrize("c", NORM, ids=_id)
def test_oracle_rms_norm_matches_reference(c):
    i, dt = c["id"], c["dtype"]
    if c["residual"]:
        y, r = lo.rms_norm(ARR[f"{i}_x"], ARR[f"{i}_w"], c["eps"], dt, ARR[f"{i}_res"])
        assert np.array_equal(r, ARR[f"{i}_res_out"])
    else:
        y = lo.rms_norm(ARR[f"{i}_x"], ARR[f"{i}_w"], c["eps"], dt)
    _close(y, ARR[f"{i}_y"], 2, 0.99, f"rms_norm {dt}")


def test_host_layers_interface_and_cpu_refusal():
    from vllm_pq_amd.host.layers import RMSNorm, SiluAndMul
    n = RMSNorm(64, eps=1e-5)
    assert n.variance_epsilon == 1e-5 and n.weight.shape == (64,) and not n.weight.requires_grad
    with pytest.raises(ValueError, match="hidden_size"):
        n(torch.zeros(2, 32, dtype=torch.bfloat16))
    with pytest.raises(NotImplementedError):
        RMSNorm(64, var_hidden_size=32)
    for call in (lambda: n(torch.zeros(2, 64, dtype=torch.bfloat16)),
                 lambda: n(torch.zeros(2, 64, dtype=torch.bfloat16), torch.zeros(2, 64, dtype=torch.bfloat16)),
                 lambda: SiluAndMul()(torch.zeros(2, 64, dtype=torch.bfloat16))):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            call()


# ---- GPU: the HIP kernels through the C ABI ---------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("c", SILU, ids=_id)
def test_gpu_silu_and_mul_golden(c):
    from vllm_pq_amd import ops
    x = tensor_of(ARR[f"{c['id']}_x"], c["dtype"], DEV)
    out = torch.empty(c["T"], c["I"], dtype=x.dtype, device=DEV)
    ops.silu_and_mul(out, x)
    _close(bits_of(out), ARR[f"{c['id']}_y"], 2, 0.99, f"gpu silu {c['dtype']}")
    _close(bits_of(out), lo.silu_and_mul(ARR[f"{c['id']}_x"], c["dtype"]), 2, 0.99, f"gpu silu vs oracle {c['dtype']}")


@pytest.mark.gpu
@pytest.mark.parametrize("c", NORM, ids=_id)
def test_gpu_rms_norm_golden(c):
    from vllm_pq_amd import ops
    i, dt = c["id"], c["dtype"]
    x = tensor_of(ARR[f"{i}_x"], dt, DEV)
    w = tensor_of(ARR[f"{i}_w"], dt, DEV)
    if c["residual"]:
        r = tensor_of(ARR[f"{i}_res"], dt, DEV)
        ops.fused_add_rms_norm(x, r, w, c["eps"])
        assert np.array_equal(bits_of(r), ARR[f"{i}_res_out"])
        y = x
    else:
        y = torch.empty_like(x)
        ops.rms_norm(y, x, w, c["eps"])
        assert np.array_equal(bits_of(x), ARR[f"{i}_x"])          # input untouched
    _close(bits_of(y), ARR[f"{i}_y"], 2, 0.99, f"gpu rms_norm {dt}")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_gpu_layers_full_size_and_strided(dtype):
    """Llama-3-8B sizes at 4096 tokens against the oracle, the layer classes, rows inside a wider buffer."""
    from vllm_pq_amd.host.layers import RMSNorm, SiluAndMul
    from vllm_pq_amd import ops
    tdt = {"bf16": torch.bfloat16, "fp16": torch.float16}[dtype]
    g = torch.Generator().manual_seed(77)
    T, I, H = 4096, 14336, 4096
    x = (torch.randn(T, 2 * I, generator=g) * 2).to(tdt)
    y = SiluAndMul()(x.to(DEV))
    assert y.shape == (T, I)
    sample = slice(0, T, 97)
    _close(bits_of(y[sample]), lo.silu_and_mul(bits_of(x[sample]), dtype), 2, 0.99, f"silu full {dtype}")

    h = torch.randn(T, H, generator=g).to(tdt)
    res = torch.randn(T, H, generator=g).to(tdt)
    norm = RMSNorm(H, eps=1e-5).to(DEV)
    norm.weight.data = (1 + 0.1 * torch.randn(H, generator=g)).to(tdt).to(DEV)
    wb = bits_of(norm.weight.data)
    out = norm(h.to(DEV))
    _close(bits_of(out[sample]), lo.rms_norm(bits_of(h[sample]), wb, 1e-5, dtype), 2, 0.99, f"rms full {dtype}")
    hd, rd = h.to(DEV), res.to(DEV)
    o2, r2 = norm(hd, rd)
    assert o2.data_ptr() == hd.data_ptr() and r2.data_ptr() == rd.data_ptr()       # in place, as the reference's op
    yo, ro = lo.rms_norm(bits_of(h[sample]), wb, 1e-5, dtype, bits_of(res[sample]))
    assert np.array_equal(bits_of(r2[sample]), ro)
    _close(bits_of(o2[sample]), yo, 2, 0.99, f"fused full {dtype}")

    # rows inside wider buffers (strides), odd widths (element-wise kernels), empty batch
    for (t, i) in ((3, 40), (5, 37), (2, 8)):
        wide = (torch.randn(t, 2 * i + 24, generator=g)).to(tdt).to(DEV)
        xin = wide[:, 8:8 + 2 * i]
        obuf = torch.zeros(t, i + 16, dtype=tdt, device=DEV)
        ops.silu_and_mul(obuf[:, :i], xin)
        _close(bits_of(obuf[:, :i]), lo.silu_and_mul(bits_of(xin), dtype), 2, 0.9, f"silu strided {dtype}")
        assert not obuf[:, i:].any()
        hin = wide[:, 3:3 + i]
        w = torch.ones(i, dtype=tdt, device=DEV)
        o = torch.empty(t, i, dtype=tdt, device=DEV)
        ops.rms_norm(o, hin, w, 1e-6)
        _close(bits_of(o), lo.rms_norm(bits_of(hin), bits_of(w), 1e-6, dtype), 2, 0.9, f"rms strided {dtype}")
    ops.silu_and_mul(torch.empty(0, 8, dtype=tdt, device=DEV), torch.empty(0, 16, dtype=tdt, device=DEV))
    with pytest.raises(ValueError):
        ops.silu_and_mul(torch.empty(2, 9, dtype=tdt, device=DEV), torch.empty(2, 16, dtype=tdt, device=DEV))
    with pytest.raises(ValueError):
        ops.rms_norm(torch.empty(2, 16, dtype=tdt, device=DEV), torch.empty(2, 16, dtype=tdt, device=DEV),
                     torch.empty(8, dtype=tdt, device=DEV), 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("T,I,Kin,bias", [(1, 512, 1024, False), (16, 14336, 4096, False), (7, 176, 1280, True),
                                          (32, 1024, 512, True), (20, 48, 256, False), (100, 512, 1024, True),
                                          (3, 14336, 4096, True), (9, 5632, 2048, False), (16, 1200, 1536, True)])
def test_gpu_fused_gate_up_silu_mul(dtype, T, I, Kin, bias):
    """The fused launch equals the GEMM followed by silu_and_mul bit for bit (same sums, same roundings), for
    batches it fuses and for those it runs as two steps; ragged tiles, bias, both dtypes."""
    from pq_testlib import Problem
    from vllm_pq_amd import ops, _lib
    p = Problem(900 + T, 256, 8, 1, [I, I], Kin, T, dtype, bias=bias).to(DEV)
    packed = ops.pq_repack_codes(p.codes_dev, p.codebooks_dev, [I, I])
    y = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, [I, I], p.bias_dev)
    want = torch.empty(T, I, dtype=y.dtype, device=DEV)
    ops.silu_and_mul(want, y)
    got = ops.pq_gemm_packed_silu_mul(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, p.bias_dev)
    fused = _lib.load().pq_gemm_silu_mul_is_fused(T, I, Kin // 8, 256, 8, 1, 1 if dtype == "bf16" else 0)
    assert fused == (1 if T <= 16 else 0)
    assert got.shape == (T, I)
    # the fused launch runs on the decode kernel where that takes the layer (whole 512-feature chunks, at most 8), else
    # on the streaming kernel
    fused_kernel = "mfma_decode_packed" if Kin % 512 == 0 and Kin <= 4096 and (Kin > 1024 or 2 * I > 4096) else "mfma_stream_packed"
    if fused and ops.gemm_kernel_name(T, 2 * I, Kin // 8, 1, 256, 8, 1, y.dtype, [I, I], packed=True) != fused_kernel:
        # the unfused GEMM ran on another kernel (other summation order): compare within the layers' tolerance
        _close(bits_of(got), bits_of(want), 2, 0.9, f"fused vs tiled {dtype}")
    else:
        assert torch.equal(got, want)
    # and through the linear method
    from vllm_pq_amd import PQConfig
    from vllm_pq_amd.host.linear import MergedColumnParallelLinear
    layer = MergedColumnParallelLinear(Kin, [I, I], bias=bias, params_dtype=y.dtype, quant_config=PQConfig(8, 8, 1, 1))
    layer.codes.data, layer.codebooks.data, layer.scales.data = p.codes, p.codebooks, p.scales
    if bias:
        layer.bias.data = p.bias
    layer.to(DEV)
    layer.quant_method.process_weights_after_loading(layer)
    out = layer.quant_method.apply_silu_and_mul(layer, p.x_dev, layer.bias if bias else None)
    assert out.shape == (T, I)
    _close(bits_of(out), bits_of(want), 2, 0.9, f"method {dtype}")
