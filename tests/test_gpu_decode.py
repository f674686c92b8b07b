"""GPU parity of the decode kernel (pq_gemm_decode_impl.cuh) through the C ABI: every block geometry it can take
(waves along K x row groups x fragments per wave x chunks per wave), against the oracle's exact arithmetic."""
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
    monkeypatch.setenv("PQ_AMD_FORCE_KERNEL", "decode")
    if cfg:
        monkeypatch.setenv("PQ_AMD_DECODE_CFG", cfg)
    _lib.reload_knobs()
    name = ops.gemm_kernel_name(p.T, p.N, p.G, p.M, p.K, p.d, p.codes.element_size(), p.x.dtype, parts, packed=True)
    assert name == "mfma_decode_packed", name
    y = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, p.bias_dev)
    y2 = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, p.bias_dev)
    assert torch.equal(y, y2), "not deterministic"
    return y


# Kin (chunks of 512) decides the waves along K: 1 chunk -> 1, 2 -> 2, 3..4 -> 4, 5..8 -> 8
@pytest.mark.parametrize("Kin,cfg", [
    (512, "1x1"), (512, "2x1"), (512, "8x1"),
    (1024, "2x1"), (1024, "2x2"), (1024, "8x2"), (1024, "4x1"),
    (1536, "4x1"), (1536, "4x3"), (1536, "8x4"),            # three chunks on four waves: one wave without work
    (2048, "4x4"), (2048, "8x2"),
    (3584, "8x1"), (3584, "8x7"),                              # seven chunks on eight waves
    (4096, "8x1"), (4096, "8x7"), (4096, "8x8"),
])
@pytest.mark.parametrize("T", [1, 5, 8, 9, 16])
def test_decode_kernel_every_geometry(ops, monkeypatch, Kin, cfg, T):
    parts = [208, 112, 16]          # partial tiles for every rows-per-block, a one-fragment partition
    p = Problem(1000 + Kin // 512, 256, 8, 1, parts, Kin, T, "bf16", bias=(T != 5)).to(DEV)
    y = _run(ops, monkeypatch, p, parts, cfg)
    _, y32 = p.oracle_gemm(order=2)
    assert_close_to_exact(bits_of(y), y32, "bf16", f"decode Kin={Kin} cfg={cfg} T={T}")


@pytest.mark.parametrize("K", [16, 32, 64, 128, 256])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_decode_kernel_codebook_sizes(ops, monkeypatch, K, dtype):
    parts, Kin, T = [160, 96], 2048, 7
    p = Problem(50 + K, K, 8, 1, parts, Kin, T, dtype, bias=True).to(DEV)
    y = _run(ops, monkeypatch, p, parts, None)
    _, y32 = p.oracle_gemm(order=2)
    assert_close_to_exact(bits_of(y), y32, dtype, f"decode K={K} {dtype}")


# K longer than the 8 chunks of one block: 9 chunks -> 2 blocks of 5 and 4, 16 -> 2 x 8, 28 -> 4 x 7, 64 -> 8 x 8
@pytest.mark.parametrize("Kin,cfg", [(4608, None), (4608, "8x3"), (8192, None), (8192, "8x8"), (14336, None),
                                     (14336, "8x2"), (32768, "8x1")])
@pytest.mark.parametrize("T", [1, 8, 13, 16])
def test_decode_kernel_k_split_over_blocks(ops, monkeypatch, Kin, cfg, T):
    parts = [208, 112, 16]
    ops.ensure_workspace(torch.device(DEV))
    p = Problem(2000 + Kin // 512 + T, 256, 8, 1, parts, Kin, T, "bf16", bias=(T != 8)).to(DEV)
    y = _run(ops, monkeypatch, p, parts, cfg)
    _, y32 = p.oracle_gemm(order=2)
    assert_close_to_exact(bits_of(y), y32, "bf16", f"decode K split Kin={Kin} cfg={cfg} T={T}")


# more than 16 tokens: groups of 16 over the grid, every group with its own rows of y and of the K-split partials
@pytest.mark.parametrize("Kin,cfg", [(1024, None), (4096, None), (4096, "8x3"), (8192, None), (14336, "8x2")])
@pytest.mark.parametrize("T", [17, 32, 100, 128])
def test_decode_kernel_token_groups(ops, monkeypatch, Kin, cfg, T):
    parts = [208, 112, 16]
    ops.ensure_workspace(torch.device(DEV))
    p = Problem(3000 + Kin // 512 + T, 256, 8, 1, parts, Kin, T, "bf16", bias=(T != 32)).to(DEV)
    y = _run(ops, monkeypatch, p, parts, cfg)
    _, y32 = p.oracle_gemm(order=2)
    assert_close_to_exact(bits_of(y), y32, "bf16", f"decode token groups Kin={Kin} cfg={cfg} T={T}")


def test_decode_kernel_long_k_without_workspace_is_the_stream_kernels(ops):
    """Without a workspace a K of more than 4096 cannot be split over blocks: the library says so through the
    kernel it names for a launch that has none, and still computes the layer."""
    parts, Kin, T = [256], 8192, 16
    p = Problem(77, 256, 8, 1, parts, Kin, T, "bf16").to(DEV)
    packed = ops.pq_repack_codes(p.codes_dev, p.codebooks_dev, parts)
    ops.ensure_workspace(torch.device(DEV))
    want = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, None)
    ops.drop_workspace()
    try:
        got = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, None)
    finally:
        ops.ensure_workspace(torch.device(DEV))
    _, y32 = p.oracle_gemm(order=2)
    assert_close_to_exact(bits_of(got), y32, "bf16", "long K, no workspace")
    assert_close_to_exact(bits_of(want), y32, "bf16", "long K, workspace")


def test_decode_kernel_edge_codes_and_one_hot(ops, monkeypatch):
    """Codes at the ends of the signed range, and one-hot activations: y[t] must be scale x one codebook element,
    exactly (a single product, no rounding in the sum)."""
    parts, Kin, T = [64], 1024, 16
    p = Problem(7, 256, 8, 1, parts, Kin, T, "bf16")
    p.codes[:, ::3, 0] = -128
    p.codes[:, 1::3, 0] = 127
    p.codes[:, 2::3, 0] = -1
    p.x.zero_()
    ks = [0, 7, 8, 511, 512, 513, 1023, 100, 200, 300, 400, 600, 700, 800, 900, 31]
    for t, k in enumerate(ks):
        p.x[t, k] = 1.0
    p.to(DEV)
    y = _run(ops, monkeypatch, p, parts, None)
    w = ops.pq_dequant(p.codes_dev, p.codebooks_dev, parts)                  # bit-exact gather
    want = (w[:, ks].float() * p.scales_dev.view(-1, 1).float()).t().to(torch.bfloat16)
    assert torch.equal(y, want)


def test_decode_kernel_strided_rows(ops, monkeypatch):
    parts, Kin, T = [256], 4096, 11
    p = Problem(9, 256, 8, 1, parts, Kin, T, "bf16").to(DEV)
    want = _run(ops, monkeypatch, p, parts, None)
    packed = ops.pq_repack_codes(p.codes_dev, p.codebooks_dev, parts)
    wide = torch.randn(T, 8192, device=DEV).to(torch.bfloat16)
    wide[:, 1024:5120] = p.x_dev
    got = ops.pq_gemm_packed(wide[:, 1024:5120], packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, None)
    assert torch.equal(got, want)


@pytest.mark.parametrize("name,parts,Kin", [
    ("8B.qkv", [4096, 1024, 1024], 4096), ("8B.o", [4096], 4096), ("8B.gate_up", [14336, 14336], 4096),
    ("8B.down", [4096], 14336), ("8B/tp4.o", [4096], 1024), ("8B/tp4.down", [4096], 3584),
    ("70B/tp8.qkv", [1024, 128, 128], 8192), ("70B/tp8.gate_up", [3584, 3584], 8192), ("70B/tp8.o", [8192], 1024)])
@pytest.mark.parametrize("T", [1, 16])
def test_decode_kernel_model_shapes_full_size(ops, name, parts, Kin, T):
    """Full-size per-rank shapes on the geometry the library picks: against an fp32 matmul over the bit-exact
    dequantised weight on the device, and linearity in x (exact in binary floating point)."""
    torch.manual_seed(T + Kin)
    N, G = sum(parts), Kin // 8
    codes = torch.randint(-128, 128, (N, G, 1), dtype=torch.int8, device=DEV)
    cb = torch.randn(len(parts), 256, 1, 8, device=DEV).to(torch.bfloat16)
    scales = (torch.rand(N, 1, 1, 1, device=DEV) + 0.5).to(torch.bfloat16)
    x = torch.randn(T, Kin, device=DEV).to(torch.bfloat16)
    packed = ops.pq_repack_codes(codes, cb, parts)
    name = ops.gemm_kernel_name(T, N, G, 1, 256, 8, 1, x.dtype, parts, packed=True)
    # the decode kernel where it is the faster one: three chunks or more, or more than 4096 rows (K longer than
    # 4096 is split over blocks through the workspace a loaded layer has)
    ops.ensure_workspace(torch.device(DEV))
    decode = Kin // 512 > 2 or N > 4096
    assert name == ("mfma_decode_packed" if decode else "mfma_stream_packed")
    y = ops.pq_gemm_packed(x, packed, codes, cb, scales, parts, None)
    w = ops.pq_dequant(codes, cb, parts)
    ref = (x.float() @ w.float().t()) * scales.view(1, -1).float()
    rms = ref.pow(2).mean().sqrt()
    tol = 0.51 * 2.0 ** -7 * ref.abs() + 1e-4 * rms
    bad = (y.float() - ref).abs() > tol
    assert not bad.any(), f"{name} T={T}: {int(bad.sum())} of {bad.numel()} outside tolerance"
    y2 = ops.pq_gemm_packed(x * 2, packed, codes, cb, scales, parts, None)
    assert torch.equal(y2, y * 2)


def test_release_reference_codes_keeps_apply_working(ops):
    """With release_reference_codes the reference-layout tensor keeps shape and dtype but not its storage; apply()
    (packed path) gives the same bits as before."""
    from vllm_pq_amd import PQConfig
    from vllm_pq_amd.host.linear import MergedColumnParallelLinear
    outs = []
    for release in (False, True):
        torch.manual_seed(11)
        layer = MergedColumnParallelLinear(1024, [256, 256], bias=False, params_dtype=torch.bfloat16,
                                           quant_config=PQConfig(8, 8, 1, 1, release_reference_codes=release))
        layer.codes.data = torch.randint(-128, 128, layer.codes.shape, dtype=torch.int8)
        layer.codebooks.data = torch.randn(layer.codebooks.shape).to(torch.bfloat16)
        layer.scales.data = (torch.rand(layer.scales.shape) + 0.5).to(torch.bfloat16)
        layer.to(DEV)
        layer.quant_method.process_weights_after_loading(layer)
        assert tuple(layer.codes.shape) == (512, 128, 1) and layer.codes.dtype == torch.int8
        assert layer.codes.untyped_storage().nbytes() == (1 if release else 512 * 128)
        x = torch.randn(40, 1024, generator=torch.Generator().manual_seed(5)).to(torch.bfloat16).to(DEV)
        outs.append((layer(x)[0], layer(x[:7])[0]))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
