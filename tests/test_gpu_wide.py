"""GPU parity of the mid-batch kernel (pq_gemm_wide_impl.cuh) through the C ABI: every block geometry it can take
(waves along K x row groups, fragments per wave and round, rounds per block), token counts either side of the
32-token groups, against the oracle's exact arithmetic; full Llama-3-8B shapes by size-independent properties."""
import numpy as np
import pytest
import torch

from pq_testlib import Problem, assert_close_to_exact, bits_of
from vllm_pq_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from vllm_pq_amd import ops as _ops
    return _ops


def _run(ops, monkeypatch, p, parts, cfg):
    packed = ops.pq_repack_codes(p.codes_dev, p.codebooks_dev, parts)
    assert packed is not None
    monkeypatch.setenv("PQ_AMD_FORCE_KERNEL", "wide")
    if cfg:
        monkeypatch.setenv("PQ_AMD_WIDE_CFG", cfg)
    _lib.reload_knobs()
    name = ops.gemm_kernel_name(p.T, p.N, p.G, p.M, p.K, p.d, p.codes.element_size(), p.x.dtype, parts, packed=True)
    assert name == "mfma_wide_packed", name
    y = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, p.bias_dev)
    y2 = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, p.bias_dev)
    assert torch.equal(y, y2), "not deterministic"
    return y


# Kin (chunks of 512) decides the waves along K: 1 chunk -> 1 (8 row groups), 2 -> 2, 3..4 -> 4, 5..8 -> 8 (1 row group);
# cfg = fragments per wave and round x rounds per block
@pytest.mark.parametrize("Kin,cfg", [
    (512, "1x1"), (512, "2x3"), (512, "4x2"),
    (1024, "1x2"), (1024, "4x4"), (1024, "3x1"),
    (1536, "2x2"), (1536, "4x1"),                               # three chunks on four waves: one wave without work
    (2048, "1x1"), (2048, "4x5"),
    (3584, "3x2"), (3584, "4x16"),                              # seven chunks on eight waves
    (4096, "1x1"), (4096, "2x7"), (4096, "4x3"), (4096, "4x16"), (4096, None),
])
@pytest.mark.parametrize("T", [1, 17, 32, 33, 50, 64, 100])
def test_wide_kernel_every_geometry(ops, monkeypatch, Kin, cfg, T):
    parts = [208, 112, 16]          # partial tiles for every rows-per-block, a one-fragment partition
    p = Problem(3000 + Kin // 512 + T, 256, 8, 1, parts, Kin, T, "bf16", bias=(T % 2 == 0)).to(DEV)
    y = _run(ops, monkeypatch, p, parts, cfg)
    _, y32 = p.oracle_gemm(order=2)
    assert_close_to_exact(bits_of(y), y32, "bf16", f"wide Kin={Kin} cfg={cfg} T={T}")


@pytest.mark.parametrize("K", [16, 32, 64, 128, 256])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_wide_kernel_codebook_sizes(ops, monkeypatch, K, dtype):
    parts, Kin, T = [160, 96], 2048, 40
    p = Problem(70 + K, K, 8, 1, parts, Kin, T, dtype, bias=True).to(DEV)
    y = _run(ops, monkeypatch, p, parts, None)
    _, y32 = p.oracle_gemm(order=2)
    assert_close_to_exact(bits_of(y), y32, dtype, f"wide K={K} {dtype}")


def test_wide_kernel_strided_activations_and_output(ops, monkeypatch):
    """x and y as column slices of wider tensors (row strides larger than the rows)."""
    parts, Kin, T = [256, 64, 64], 1024, 70
    p = Problem(91, 256, 8, 1, parts, Kin, T, "bf16", bias=True).to(DEV)
    packed = ops.pq_repack_codes(p.codes_dev, p.codebooks_dev, parts)
    monkeypatch.setenv("PQ_AMD_FORCE_KERNEL", "wide")
    _lib.reload_knobs()
    wide_x = torch.zeros(T, Kin + 64, dtype=torch.bfloat16, device=DEV)
    wide_x[:, 32:32 + Kin] = p.x_dev
    y = ops.pq_gemm_packed(wide_x[:, 32:32 + Kin], packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, p.bias_dev)
    _, y32 = p.oracle_gemm(order=2)
    assert_close_to_exact(bits_of(y), y32, "bf16", "wide strided x")


@pytest.mark.parametrize("name,parts,Kin", [("qkv", [4096, 1024, 1024], 4096), ("o", [4096], 4096),
                                            ("gate_up", [14336, 14336], 4096), ("tp4.o", [4096], 1024),
                                            ("tp4.qkv", [1024, 256, 256], 4096)])
@pytest.mark.parametrize("T", [48, 128, 256, 512])
def test_wide_kernel_llama3_shapes_full_size(ops, monkeypatch, name, parts, Kin, T):
    """Full BASELINE sizes, checked on the device by size-independent properties: the same inputs through the tiled
    kernel (whose own parity is pinned against the oracle elsewhere) give the same values up to the summation order,
    and the exact fp32 matmul over the bit-exact dequantised weight is matched within the stated tolerance."""
    p = Problem(7000 + T, 256, 8, 1, parts, Kin, T, "bf16", bias=False).to(DEV)
    y = _run(ops, monkeypatch, p, parts, None)
    w = ops.pq_dequant(p.codes_dev, p.codebooks_dev, parts).double() * p.scales_dev.view(-1, 1).double()
    ref = (p.x_dev.double() @ w.t()).cpu().numpy()
    assert_close_to_exact(bits_of(y), ref, "bf16", f"wide {name} T={T}")
