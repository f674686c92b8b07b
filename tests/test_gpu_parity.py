"""Parity of the HIP path (through the C ABI, libpq_amd.so) with the CPU oracle and with the golden
vectors produced by the reference's own Python.  Runs on a real MI355X: `pytest -m gpu`."""
import numpy as np
import pytest
import torch

from conftest import golden
from vllm_pq_amd import _lib
from oracle import pq_oracle as orc
from pq_testlib import (Problem, assert_close_to_exact, assert_close_to_reference_order, bits_of, tensor_of)

pytestmark = pytest.mark.gpu
G = golden()
DEV = "cuda:0"


def _id(c):
    return f"{c['id']}-K{c['K']}d{c['d']}M{c['M']}-{c['dtype']}"


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from vllm_pq_amd import ops as o
    return o


def _golden_tensors(case):
    dt = case["dtype"]
    codes = torch.from_numpy(G.get(case, "codes")).to(DEV)
    cb = tensor_of(G.get(case, "codebooks"), dt, DEV)
    sc = G.get(case, "scales")
    sc = None if sc is None else tensor_of(sc, dt, DEV)
    return codes, cb, sc


# ---- gather / dequant: bit-exact --------------------------------------------------------------
@pytest.mark.parametrize("case", G.cases("gather"), ids=_id)
def test_gather_golden_bit_exact(ops, case):
    codes, cb, _ = _golden_tensors(case)
    out = ops.pq_dequant(codes, cb, case["parts"])
    assert np.array_equal(bits_of(out), G.get(case, "out"))


@pytest.mark.parametrize("case", G.cases("dequant"), ids=_id)
def test_scaled_dequant_golden(ops, case):
    codes, cb, sc = _golden_tensors(case)
    out = bits_of(ops.pq_dequant(codes, cb, case["parts"], sc))
    # the oracle in nearest-even mode is the bit-exact target (M >= 2 bf16: torch's CPU kernel rounds
    # exact ties of the codebook sum away from zero, the HIP kernel to even; see oracle/pq_oracle.c)
    want = orc.dequant(G.get(case, "codes"), G.get(case, "codebooks").reshape(-1, case["K"], case["d"]),
                       G.get(case, "scales").reshape(-1), case["parts"], case["dtype"])
    assert np.array_equal(out, want)
    if case["M"] == 1 or case["dtype"] == "fp16":
        assert np.array_equal(out, G.get(case, "out"))
    else:
        gold = G.get(case, "out")
        diff = np.abs(orc.to_f32(out, "bf16") - orc.to_f32(gold, "bf16"))
        assert (out != gold).mean() < 0.15 and (diff <= 2 * 2.0 ** -7 * np.abs(orc.to_f32(gold, "bf16")) * 1.01).all()


@pytest.mark.parametrize("K,d,M,parts,Kin,dtype", [
    (256, 8, 1, [1024, 256, 256], 4096, "bf16"),     # Llama-3-8B qkv at TP=4
    (256, 8, 1, [512], 14336, "bf16"),               # down_proj rows
    (256, 4, 1, [768, 768, 768], 768, "fp16"),       # OPT-125m qkv
    (16, 2, 1, [256], 512, "bf16"),
    (4096, 8, 1, [256], 1024, "bf16"),
    (4096, 4, 2, [64, 64], 256, "fp16"),             # codebooks too large for LDS -> L2 gather
    (256, 8, 2, [200, 56], 1000, "bf16"),            # ragged: Kin % 8 == 0, odd partition sizes
    (256, 3, 1, [33, 31], 99, "bf16"),               # d not a divisor of 8 -> element kernel
    (65536, 8, 1, [64], 512, "fp16"),                # AQLM 1x16
])
def test_dequant_seeded(ops, K, d, M, parts, Kin, dtype):
    p = Problem(7, K, d, M, parts, Kin, 1, dtype, scales="normal").to(DEV)
    for scaled in (False, True):
        out = ops.pq_dequant(p.codes_dev, p.codebooks_dev, parts, p.scales_dev if scaled else None)
        assert np.array_equal(bits_of(out), p.oracle_dequant(scaled)), f"scaled={scaled}"


# ---- fused GEMM ----------------------------------------------------------------------------------
@pytest.mark.parametrize("kernel", ["auto", "generic"])
@pytest.mark.parametrize("case", G.cases("gemm"), ids=_id)
def test_gemm_golden(ops, case, kernel, monkeypatch):
    monkeypatch.setenv("PQ_AMD_FORCE_KERNEL", "" if kernel == "auto" else kernel)
    _lib.reload_knobs()
    dt = case["dtype"]
    codes, cb, sc = _golden_tensors(case)
    x = tensor_of(G.get(case, "x"), dt, DEV)
    b = G.get(case, "bias")
    b = None if b is None else tensor_of(b, dt, DEV)
    y = bits_of(ops.pq_gemm(x, codes, cb, sc, case["parts"], b))
    assert y.shape == (case["T"], case["N"])
    _, y32 = orc.gemm(G.get(case, "x"), G.get(case, "codes"), G.get(case, "codebooks").reshape(-1, case["K"], case["d"]),
                      G.get(case, "scales").reshape(-1), case["parts"], G.get(case, "bias"), dt, order=2,
                      acc_double=True, want_f32=True)
    assert_close_to_exact(y, y32, dt, "vs exact")
    assert_close_to_reference_order(y, G.get(case, "out"), dt, "vs reference output")


SEEDED = [
    # (K, d, M, parts, Kin, T, dtype, bias)
    (256, 8, 1, [1024, 256, 256], 4096, 1, "bf16", False),    # Llama-3-8B qkv, TP=4
    (256, 8, 1, [1024, 256, 256], 4096, 16, "bf16", False),
    (256, 8, 1, [1024, 256, 256], 4096, 37, "bf16", False),
    (256, 8, 1, [512], 3584, 5, "bf16", False),               # down_proj K slice at TP=4 (7 chunks)
    (256, 8, 1, [896, 896], 4096, 8, "fp16", False),          # gate_up slice
    (256, 4, 1, [768, 768, 768], 768, 9, "fp16", True),       # OPT-125m qkv, bias live
    (256, 4, 1, [768], 3072, 3, "bf16", True),                # OPT-125m fc2
    (16, 8, 1, [512], 512, 4, "bf16", False),
    (16, 4, 1, [512], 512, 16, "fp16", True),
    (16, 2, 1, [256], 512, 2, "bf16", False),
    (256, 2, 1, [256], 512, 16, "bf16", False),
    (4096, 8, 1, [256], 1024, 6, "bf16", True),
    (4096, 4, 1, [256], 512, 1, "fp16", False),
    (4096, 2, 1, [128], 256, 16, "bf16", False),
    (256, 8, 2, [512, 512], 1024, 6, "fp16", True),           # AQLM 2x8
    (65536, 8, 1, [64], 512, 2, "fp16", False),               # AQLM 1x16 -> generic kernel
    (256, 8, 1, [100, 60], 136, 3, "bf16", True),             # ragged -> generic kernel
    (256, 8, 1, [40, 24], 128, 70, "bf16", False),            # partial row tiles, T > 64
    (256, 8, 1, [512, 128, 128], 1024, 300, "bf16", True),    # tiled kernel: 2 token tiles, 3 partitions
    (256, 8, 1, [272], 576, 130, "fp16", False),              # partial row tile, K not a multiple of 512
    (256, 4, 1, [256, 256], 768, 65, "bf16", True),
    (16, 2, 1, [256], 512, 100, "bf16", False),
    (4096, 8, 1, [256], 1024, 48, "fp16", True),
    (256, 8, 2, [256], 1024, 40, "bf16", False),
]


@pytest.mark.parametrize("K,d,M,parts,Kin,T,dtype,bias", SEEDED)
def test_gemm_seeded(ops, K, d, M, parts, Kin, T, dtype, bias):
    p = Problem(1234, K, d, M, parts, Kin, T, dtype, bias=bias).to(DEV)
    y = ops.pq_gemm(p.x_dev, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, p.bias_dev)
    assert y.dtype == p.x_dev.dtype and y.shape == (T, p.N)
    _, y32 = p.oracle_gemm(order=2)
    assert_close_to_exact(bits_of(y), y32, dtype)
    ref0 = p.oracle_gemm(order=0, want_f32=False)
    assert_close_to_reference_order(bits_of(y), ref0, dtype)


@pytest.mark.parametrize("K,d,M,parts,Kin,T,dtype,bias", [c for c in SEEDED if all(s % 16 == 0 for s in c[3])])
def test_gemm_packed_equals_reference_layout(ops, K, d, M, parts, Kin, T, dtype, bias):
    """pq_repack_codes + pq_gemm_packed give the bits pq_gemm gives on the reference layout (same sums
    in the same order), and stay within the stated tolerance of the oracle."""
    p = Problem(1234, K, d, M, parts, Kin, T, dtype, bias=bias).to(DEV)
    packed = ops.pq_repack_codes(p.codes_dev, p.codebooks_dev, parts)
    name = ops.gemm_kernel_name(T, p.N, p.G, M, K, d, p.codes.element_size(), p.x.dtype, parts)
    if K > 4096:                   # K = 65536: the codebook does not fit LDS, no MFMA kernel, no packed layout
        assert packed is None and name == "generic"
        return
    assert packed is not None and packed.dtype == torch.uint8
    pname = ops.gemm_kernel_name(T, p.N, p.G, M, K, d, p.codes.element_size(), p.x.dtype, parts, packed=True)
    # the tiled kernel takes T > 16 when in_features is a whole number of its code pieces; the decode kernel takes
    # batches of up to 128 tokens (as groups of 16) where its modelled launch is the shortest
    assert pname in (("mfma_tiled_packed", "mfma_stream_packed", "mfma_decode_packed") if T > 16
                     else ("mfma_stream_packed", "mfma_decode_packed"))
    y = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, p.bias_dev)
    _, y32 = p.oracle_gemm(order=2)
    assert_close_to_exact(bits_of(y), y32, dtype)
    if name == "mfma_stream":
        y_nat = ops.pq_gemm(p.x_dev, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, p.bias_dev)
        assert_close_to_exact(bits_of(y_nat), y32, dtype)


def test_repack_refuses_unaligned_partitions(ops):
    p = Problem(2, 256, 8, 1, [40, 24], 128, 1, "bf16").to(DEV)
    assert ops.pq_repack_codes(p.codes_dev, p.codebooks_dev, [40, 24]) is None


def test_gemm_input_shapes_and_strides(ops):
    p = Problem(5, 256, 8, 1, [256], 512, 12, "bf16", bias=True).to(DEV)
    want = ops.pq_gemm(p.x_dev, p.codes_dev, p.codebooks_dev, p.scales_dev, [256], p.bias_dev)
    # [batch, seq, K] input keeps its leading dims (aqlm.py:103)
    y3 = ops.pq_gemm(p.x_dev.view(3, 4, 512), p.codes_dev, p.codebooks_dev, p.scales_dev, [256], p.bias_dev)
    assert y3.shape == (3, 4, 256) and torch.equal(y3.view(12, 256), want)
    # row-strided view (a column slice of a wider activation)
    wide = torch.zeros(12, 1024, dtype=torch.bfloat16, device=DEV)
    wide[:, :512] = p.x_dev
    ys = ops.pq_gemm(wide[:, :512], p.codes_dev, p.codebooks_dev, p.scales_dev, [256], p.bias_dev)
    assert torch.equal(ys, want)
    # the input is not modified, an empty batch gives an empty result
    assert torch.equal(wide[:, :512], p.x_dev)
    y0 = ops.pq_gemm(p.x_dev[:0], p.codes_dev, p.codebooks_dev, p.scales_dev, [256], None)
    assert y0.shape == (0, 256)


def test_gemm_one_hot_rows_reproduce_the_gather(ops):
    """x = e_k makes y[k][n] = act(scale[n] * W[n][k]): the GEMM path decodes exactly the entries the
    bit-exact gather does."""
    K, d, parts, Kin = 256, 8, [256, 128], 256
    p = Problem(11, K, d, 1, parts, Kin, Kin, "bf16").to(DEV)
    eye = torch.eye(Kin, dtype=torch.bfloat16, device=DEV)
    y = ops.pq_gemm(eye, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, None)       # [Kin, N]
    w = ops.pq_dequant(p.codes_dev, p.codebooks_dev, parts)                             # [N, Kin]
    want = (w.float() * p.scales_dev.view(-1, 1).float()).to(torch.bfloat16).t().contiguous()
    assert torch.equal(y, want)


def test_misaligned_scales_and_bias_take_another_kernel(ops):
    """The tiled kernel reads scales and bias eight bytes at a time and refuses tensors that are not 8-byte
    aligned (views at an odd offset); the launch then runs on another kernel with the same result."""
    parts, Kin, T = [512, 256], 1024, 200
    p = Problem(18, 256, 8, 1, parts, Kin, T, "bf16", bias=True).to(DEV)
    packed = ops.pq_repack_codes(p.codes_dev, p.codebooks_dev, parts)
    want = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, p.bias_dev)
    sc = torch.zeros(p.N + 1, 1, 1, 1, device=DEV, dtype=torch.bfloat16)
    sc[1:] = p.scales_dev
    bi = torch.zeros(p.N + 3, device=DEV, dtype=torch.bfloat16)
    bi[3:] = p.bias_dev
    assert sc[1:].data_ptr() % 8 and bi[3:].data_ptr() % 8
    got = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, sc[1:], parts, bi[3:])
    _, y32 = p.oracle_gemm(order=2)
    assert_close_to_exact(bits_of(got), y32, "bf16", "misaligned scales/bias")
    assert_close_to_exact(bits_of(want), y32, "bf16", "aligned scales/bias")


@pytest.mark.parametrize("name,parts,Kin", [
    ("qkv_proj", [4096, 1024, 1024], 4096), ("o_proj", [4096], 4096),
    ("gate_up_proj", [14336, 14336], 4096), ("down_proj", [4096], 14336)])
@pytest.mark.parametrize("T", [1, 16])
def test_llama3_8b_full_size_properties(ops, name, parts, Kin, T):
    """BASELINE sizes (Llama-3-8B, K=256, d=8, bf16, TP=1): too large for the CPU oracle in a test, so
    check size-independent properties on the device: (1) the gather equals torch's own index_select
    bit for bit; (2) the fused GEMM equals an fp32 matmul over that weight to the stated tolerance."""
    torch.manual_seed(0)
    N, G = sum(parts), Kin // 8
    codes = torch.randint(-128, 128, (N, G, 1), dtype=torch.int8, device=DEV)
    cb = torch.randn(len(parts), 256, 1, 8, device=DEV).to(torch.bfloat16)
    scales = (torch.rand(N, 1, 1, 1, device=DEV) + 0.5).to(torch.bfloat16)
    x = torch.randn(T, Kin, device=DEV).to(torch.bfloat16)
    w = ops.pq_dequant(codes, cb, parts)
    off = 0
    for pi, sz in enumerate(parts):
        idx = (codes[off:off + sz, :, 0].long() & 255).reshape(-1)
        want = cb[pi, :, 0, :].index_select(0, idx).view(sz, Kin)
        assert torch.equal(w[off:off + sz], want)
        off += sz
    y = ops.pq_gemm(x, codes, cb, scales, parts, None)
    ref = (x.float() @ w.float().t()) * scales.view(1, -1).float()
    rms = ref.pow(2).mean().sqrt()
    tol = 0.51 * 2.0 ** -7 * ref.abs() + 1e-4 * rms        # fp32 reference itself carries ~1e-5 rms noise
    assert ((y.float() - ref).abs() <= tol).all()


# ---- through the plugin surface -------------------------------------------------------------------
def test_apply_through_parallel_linear_layers(ops):
    from vllm_pq_amd import PQConfig, PQLinearMethod
    from vllm_pq_amd.host.linear import MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear
    cfg = PQConfig(8, 8, 1, 1)
    torch.manual_seed(1)
    qkv = QKVParallelLinear(512, 64, 8, 2, bias=False, params_dtype=torch.bfloat16, quant_config=cfg)
    gate_up = MergedColumnParallelLinear(512, [1024, 1024], bias=False, params_dtype=torch.bfloat16, quant_config=cfg)
    down = RowParallelLinear(1024, 512, bias=True, params_dtype=torch.bfloat16, quant_config=cfg)
    for layer in (qkv, gate_up, down):
        assert isinstance(layer.quant_method, PQLinearMethod)
        layer.codes.data = torch.randint(-128, 128, layer.codes.shape, dtype=torch.int8)
        layer.codebooks.data = torch.randn(layer.codebooks.shape).to(torch.bfloat16)
        layer.scales.data = (torch.rand(layer.scales.shape) + 0.5).to(torch.bfloat16)
        if layer.bias is not None:
            layer.bias.data = torch.randn(layer.bias.shape).to(torch.bfloat16)
        layer.to(DEV)
        layer.quant_method.process_weights_after_loading(layer)
        x = torch.randn(5, layer.input_size, device=DEV).to(torch.bfloat16)
        y, out_bias = layer(x)
        assert out_bias is None
        parts = layer.codebooks.output_partition_sizes
        _, y32 = orc.gemm(bits_of(x), layer.codes.cpu().numpy(), bits_of(layer.codebooks).reshape(-1, 256, 8),
                          bits_of(layer.scales).reshape(-1), parts,
                          None if layer.bias is None else bits_of(layer.bias), "bf16", order=2, want_f32=True)
        assert_close_to_exact(bits_of(y), y32, "bf16", type(layer).__name__)


def test_apply_follows_replaced_weight_tensors(ops):
    """apply() keeps a layer's launch arguments validated once; they are rebuilt when a weight tensor is replaced
    (another storage), never used stale: doubled scales double the output exactly, new codebooks and re-processed
    codes give the other weight's result, and a wrong input is still refused."""
    from vllm_pq_amd import PQConfig
    from vllm_pq_amd.host.linear import RowParallelLinear
    a = Problem(31, 256, 8, 1, [256], 512, 6, "bf16", bias=False).to(DEV)
    b = Problem(32, 256, 8, 1, [256], 512, 6, "bf16", bias=False).to(DEV)
    layer = RowParallelLinear(512, 256, bias=False, params_dtype=torch.bfloat16, quant_config=PQConfig(8, 8, 1, 1))
    layer.codes.data, layer.codebooks.data, layer.scales.data = a.codes, a.codebooks, a.scales
    layer.to(DEV)
    m = layer.quant_method
    m.process_weights_after_loading(layer)
    y0 = m.apply(layer, a.x_dev)
    assert layer.pq_prepared is not None and torch.equal(y0, m.apply(layer, a.x_dev))
    want = ops.pq_gemm_packed(a.x_dev, layer.pq_packed_codes, layer.codes, layer.codebooks, layer.scales, [256], None)
    assert torch.equal(y0, want)
    first = layer.pq_prepared
    layer.scales.data = (layer.scales.data.float() * 2).to(torch.bfloat16)          # a new tensor
    y1 = m.apply(layer, a.x_dev)
    assert layer.pq_prepared is not first
    assert torch.equal(y1, (y0.float() * 2).to(torch.bfloat16))
    layer.codes.data, layer.codebooks.data, layer.scales.data = b.codes_dev, b.codebooks_dev, b.scales_dev
    m.process_weights_after_loading(layer)
    yb = m.apply(layer, a.x_dev)
    wantb = ops.pq_gemm_packed(a.x_dev, layer.pq_packed_codes, layer.codes, layer.codebooks, layer.scales, [256], None)
    assert torch.equal(yb, wantb) and not torch.equal(yb, y0)
    # 3-D activations keep their leading dimensions; wrong width, dtype or device are refused
    y3 = m.apply(layer, a.x_dev.reshape(2, 3, 512))
    assert y3.shape == (2, 3, 256) and torch.equal(y3.reshape(6, 256), yb)
    with pytest.raises(ValueError):
        m.apply(layer, a.x_dev[:, :504])
    with pytest.raises(ValueError):
        m.apply(layer, a.x_dev.to(torch.float16))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.apply(layer, a.x_dev.cpu())


def test_ops_reject_cpu_tensors_and_bad_shapes(ops):
    p = Problem(3, 256, 8, 1, [64], 128, 2, "bf16")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.pq_gemm(p.x, p.codes, p.codebooks, p.scales, [64], None)
    p.to(DEV)
    with pytest.raises(ValueError):
        ops.pq_gemm(p.x_dev, p.codes_dev, p.codebooks_dev, p.scales_dev, [32, 16], None)   # sum != N, P*M mismatch
    with pytest.raises(ValueError):
        ops.pq_gemm(p.x_dev[:, :64], p.codes_dev, p.codebooks_dev, p.scales_dev, [64], None)


# ---- every tile shape of both MFMA kernels, forced one by one ----------------------------------------
def _device_reference(ops, p, parts):
    """fp32 matmul over the (bit-exact) dequantised weight, on the device: (y_ref, rms)."""
    w = ops.pq_dequant(p.codes_dev, p.codebooks_dev, parts)
    ref = (p.x_dev.float() @ w.float().t()) * p.scales_dev.view(1, -1).float()
    if p.bias_dev is not None:
        ref = ref + p.bias_dev.float()
    return ref, ref.pow(2).mean().sqrt()


@pytest.mark.parametrize("cfg", range(7))
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_tiled_kernel_every_tile_shape(ops, monkeypatch, cfg, dtype):
    """Partial row tiles in both partitions, a partial token tile, bias; the shape is forced, so the largest
    ones run far below their efficient size -- this checks indexing, not speed."""
    parts, Kin, T = [528, 272], 1024, 200
    p = Problem(77 + cfg, 256, 8, 1, parts, Kin, T, dtype, bias=True).to(DEV)
    packed = ops.pq_repack_codes(p.codes_dev, p.codebooks_dev, parts)
    monkeypatch.setenv("PQ_AMD_FORCE_KERNEL", "tiled")
    monkeypatch.setenv("PQ_AMD_TILED_CFG", str(cfg))
    _lib.reload_knobs()
    y = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, p.bias_dev)
    y2 = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, p.bias_dev)
    assert torch.equal(y, y2)                                   # run-to-run identical bits
    _, y32 = p.oracle_gemm(order=2)
    assert_close_to_exact(bits_of(y), y32, dtype, f"tiled cfg {cfg}")


@pytest.mark.parametrize("slots", [1, 2, 3])
@pytest.mark.parametrize("cfg,fmt", [(5, (256, 8, 1)), (2, (256, 8, 1)), (4, (256, 8, 1)), (5, (256, 4, 1)),
                                     (2, (256, 8, 2)), (5, (4096, 8, 1))])
def test_tiled_kernel_blocks_walk_several_tiles(ops, monkeypatch, slots, cfg, fmt):
    """A block of the tiled kernel takes every `8 * slots`-th tile of its run: with few blocks each one computes
    several tiles in a row, across token tiles, row tiles and partitions (codebook image rebuilt in between, ring
    position and code registers carried over).  Same bits as with one tile per block."""
    K, d, M = fmt
    parts, Kin, T = [208, 64, 144], 1024, 330
    p = Problem(900 + cfg, K, d, M, parts, Kin, T, "bf16", bias=True).to(DEV)
    packed = ops.pq_repack_codes(p.codes_dev, p.codebooks_dev, parts)
    assert packed is not None
    monkeypatch.setenv("PQ_AMD_FORCE_KERNEL", "tiled")
    monkeypatch.setenv("PQ_AMD_TILED_CFG", str(cfg))
    monkeypatch.setenv("PQ_AMD_TILED_SLOTS", "100000")
    _lib.reload_knobs()
    y1 = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, p.bias_dev)
    monkeypatch.setenv("PQ_AMD_TILED_SLOTS", str(slots))
    _lib.reload_knobs()
    y = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, p.bias_dev)
    assert torch.equal(y, y1)
    _, y32 = p.oracle_gemm(order=2)
    assert_close_to_exact(bits_of(y), y32, "bf16", f"K={K} d={d} M={M} cfg {cfg} slots {slots}")


@pytest.mark.parametrize("fmt", [(256, 4, 1), (16, 2, 1), (4096, 8, 1), (256, 8, 2)])
@pytest.mark.parametrize("cfg", [0, 2, 5, 6])
def test_tiled_kernel_other_formats_tile_shapes(ops, monkeypatch, fmt, cfg):
    K, d, M = fmt
    parts, Kin, T = [272, 144], 1024, 150
    p = Problem(5 + cfg, K, d, M, parts, Kin, T, "bf16", bias=False).to(DEV)
    packed = ops.pq_repack_codes(p.codes_dev, p.codebooks_dev, parts)
    assert packed is not None
    monkeypatch.setenv("PQ_AMD_FORCE_KERNEL", "tiled")
    monkeypatch.setenv("PQ_AMD_TILED_CFG", str(cfg))
    _lib.reload_knobs()
    y = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, None)
    _, y32 = p.oracle_gemm(order=2)
    assert_close_to_exact(bits_of(y), y32, "bf16", f"K={K} d={d} M={M} cfg {cfg}")


@pytest.mark.parametrize("cfg,T", [("1x1", 3), ("2x1", 16), ("4x1", 9), ("7x1", 16), ("8x1", 1), ("4x2", 30),
                                   ("4x2", 70), ("2x4", 64), ("4x4", 100), ("7x2", 32), ("7x2", 50),
                                   ("8x1", 64), ("2x1", 40)])
def test_stream_kernel_every_tile_shape(ops, monkeypatch, cfg, T):
    """K spans 2.5 chunks per wave set (ragged last chunk), rows leave partial tiles for every NF."""
    parts, Kin = [400, 176], 1280
    p = Problem(31, 256, 8, 1, parts, Kin, T, "bf16", bias=True).to(DEV)
    packed = ops.pq_repack_codes(p.codes_dev, p.codebooks_dev, parts)
    monkeypatch.setenv("PQ_AMD_FORCE_KERNEL", "stream")
    monkeypatch.setenv("PQ_AMD_STREAM_CFG", cfg)
    _lib.reload_knobs()
    y = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, p.bias_dev)
    y2 = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, p.bias_dev)
    assert torch.equal(y, y2)
    _, y32 = p.oracle_gemm(order=2)
    assert_close_to_exact(bits_of(y), y32, "bf16", f"stream cfg {cfg}")


def test_tiled_kernel_strided_activations(ops):
    parts, Kin, T = [512], 1024, 100
    p = Problem(8, 256, 8, 1, parts, Kin, T, "bf16").to(DEV)
    packed = ops.pq_repack_codes(p.codes_dev, p.codebooks_dev, parts)
    want = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, None)
    wide = torch.randn(T, 2048, device=DEV).to(torch.bfloat16)
    wide[:, 512:1536] = p.x_dev
    got = ops.pq_gemm_packed(wide[:, 512:1536], packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, None)
    assert torch.equal(got, want)


@pytest.mark.parametrize("name,parts,Kin", [
    ("qkv_proj", [4096, 1024, 1024], 4096), ("o_proj", [4096], 4096),
    ("gate_up_proj", [14336, 14336], 4096), ("down_proj", [4096], 14336)])
@pytest.mark.parametrize("T", [1, 16, 64, 300, 4096])
def test_llama3_8b_full_size_packed_path(ops, name, parts, Kin, T):
    """The production path at BASELINE sizes: packed codes, whichever kernel and tile shape the dispatch
    picks at this T, against an fp32 matmul over the bit-exact dequantised weight (on the device)."""
    torch.manual_seed(T)
    N, G = sum(parts), Kin // 8
    codes = torch.randint(-128, 128, (N, G, 1), dtype=torch.int8, device=DEV)
    cb = torch.randn(len(parts), 256, 1, 8, device=DEV).to(torch.bfloat16)
    scales = (torch.rand(N, 1, 1, 1, device=DEV) + 0.5).to(torch.bfloat16)
    x = torch.randn(T, Kin, device=DEV).to(torch.bfloat16)
    packed = ops.pq_repack_codes(codes, cb, parts)
    y = ops.pq_gemm_packed(x, packed, codes, cb, scales, parts, None)
    w = ops.pq_dequant(codes, cb, parts)
    ref = (x.float() @ w.float().t()) * scales.view(1, -1).float()
    rms = ref.pow(2).mean().sqrt()
    tol = 0.51 * 2.0 ** -7 * ref.abs() + 1e-4 * rms        # fp32 reference itself carries ~1e-5 rms noise
    bad = (y.float() - ref).abs() > tol
    assert not bad.any(), f"{name} T={T}: {int(bad.sum())} of {bad.numel()} outside tolerance"
    # linearity in x (size-independent property): y(2x) == 2 y(x) exactly in binary floating point
    y2 = ops.pq_gemm_packed(x * 2, packed, codes, cb, scales, parts, None)
    assert torch.equal(y2, y * 2)


@pytest.mark.parametrize("hidden", [4096, 8192])
@pytest.mark.parametrize("K", [16, 256, 4096])
@pytest.mark.parametrize("d", [2, 4, 8])
@pytest.mark.parametrize("T", [16, 256, 2048])
def test_square_sweep_formats_full_size(ops, hidden, K, d, T):
    """Every (codebook size, sub-vector length) of the square sweep at its full size (hidden x hidden), on the
    kernel the dispatch picks at this T: against an fp32 matmul over the bit-exact dequantised weight on the device,
    plus linearity in x (exact in binary floating point), the size-independent property of the path."""
    torch.manual_seed(hidden + K + d + T)
    N, G, parts = hidden, hidden // d, [hidden]
    idt = torch.int8 if K <= 256 else torch.int16
    lo, hi = (-(K // 2), K // 2) if K >= 256 else (-128, 128)      # stored values reduce mod K to valid indices
    codes = torch.randint(lo, hi, (N, G, 1), dtype=torch.int32, device=DEV).to(idt)
    cb = torch.randn(1, K, 1, d, device=DEV).to(torch.bfloat16)
    scales = (torch.rand(N, 1, 1, 1, device=DEV) + 0.5).to(torch.bfloat16)
    x = torch.randn(T, hidden, device=DEV).to(torch.bfloat16)
    packed = ops.pq_repack_codes(codes, cb, parts)
    assert packed is not None

    def run(inp):
        return ops.pq_gemm_packed(inp, packed, codes, cb, scales, parts, None)

    y = run(x)
    w = ops.pq_dequant(codes, cb, parts)
    # the dequantised weight the reference is built from is itself checked: torch's own gather, bit for bit
    idx = (codes[:, :, 0].long() & (K - 1)).reshape(-1)
    assert torch.equal(w, cb[0, :, 0, :].index_select(0, idx).view(N, hidden))
    del idx
    ref = (x.float() @ w.float().t()) * scales.view(1, -1).float()
    rms = ref.pow(2).mean().sqrt()
    tol = 0.51 * 2.0 ** -7 * ref.abs() + 1e-4 * rms
    bad = (y.float() - ref).abs() > tol
    kern = ops.gemm_kernel_name(T, N, G, 1, K, d, codes.element_size(), x.dtype, parts, packed=True)
    assert not bad.any(), f"K={K} d={d} T={T} {kern}: {int(bad.sum())} of {bad.numel()} outside tolerance"
    assert torch.equal(run(x * 2), y * 2)


# ---- K split over blocks (workspace) -----------------------------------------------------------------
@pytest.mark.parametrize("T", [1, 7, 16, 40, 100])
@pytest.mark.parametrize("ksplit", [2, 3, 4, 8])
def test_stream_kernel_k_split(ops, monkeypatch, T, ksplit):
    """Partial sums through the workspace, last block adds them up: within tolerance of the oracle, identical
    bits run to run, counters left at zero (a second and third launch reuse them), no split without a workspace."""
    parts, Kin = [208, 112], 512 * 11            # 11 chunks: uneven ranges for every split
    p = Problem(41 + ksplit, 256, 8, 1, parts, Kin, T, "bf16", bias=True).to(DEV)
    packed = ops.pq_repack_codes(p.codes_dev, p.codebooks_dev, parts)
    monkeypatch.setenv("PQ_AMD_FORCE_KERNEL", "stream")
    monkeypatch.setenv("PQ_AMD_STREAM_KSPLIT", str(ksplit))
    _lib.reload_knobs()
    N, G = p.N, p.G
    need = ops.workspace_bytes(T, N, G, 1, 256, 8, 1, torch.bfloat16)
    assert need >= 16384 + ksplit * T * N * 4
    ops.drop_workspace()
    y_plain = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, p.bias_dev)
    ops.ensure_workspace(torch.device(DEV), need)
    try:
        ys = [ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, p.bias_dev)
              for _ in range(3)]
        torch.cuda.synchronize()
        ws = ops._workspace(torch.device(DEV))
        assert int(ws[:16384].view(torch.int32).abs().sum()) == 0
    finally:
        ops.drop_workspace()
    assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2])
    _, y32 = p.oracle_gemm(order=2)
    assert_close_to_exact(bits_of(ys[0]), y32, "bf16", f"ksplit {ksplit}")
    assert_close_to_exact(bits_of(y_plain), y32, "bf16", "no workspace")


def test_k_split_default_on_long_k_layers(ops):
    """down_proj-like shape: the library asks for a workspace, and with one the result stays within tolerance."""
    parts, Kin, T = [512], 14336, 16
    p = Problem(3, 256, 8, 1, parts, Kin, T, "bf16").to(DEV)
    need = ops.workspace_bytes(T, p.N, p.G, 1, 256, 8, 1, torch.bfloat16)
    assert need > 0
    assert ops.workspace_bytes(T, p.N, 512, 1, 256, 8, 1, torch.bfloat16) == 0      # K = 4096: never split
    packed = ops.pq_repack_codes(p.codes_dev, p.codebooks_dev, parts)
    ops.ensure_workspace(torch.device(DEV), need)
    try:
        y = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, None)
    finally:
        ops.drop_workspace()
    ref, rms = _device_reference(ops, p, parts)
    tol = 0.51 * 2.0 ** -7 * ref.abs() + 1e-4 * rms
    assert ((y.float() - ref).abs() <= tol).all()


@pytest.mark.parametrize("T", [16, 128])
def test_k_split_many_rows_long_k(ops, T):
    """8192 rows x 14336 inputs: at 128 tokens the preferred split's partial sums (16 MiB) exceed the fixed
    workspace, so the launch takes the largest split that fits; the workspace is the same buffer at every batch
    size (never replaced), and its counters are zero again afterwards."""
    parts, Kin = [8192], 14336
    p = Problem(5 + T, 256, 8, 1, parts, Kin, T, "bf16").to(DEV)
    assert ops.workspace_bytes(T, p.N, p.G, 1, 256, 8, 1, torch.bfloat16) > 0
    packed = ops.pq_repack_codes(p.codes_dev, p.codebooks_dev, parts)
    dev = torch.device(DEV)
    ops.ensure_workspace(dev)
    try:
        ws = ops._workspace(dev)
        y = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, None)
        ops.ensure_workspace(dev, ops.workspace_bytes(128, p.N, p.G, 1, 256, 8, 1, torch.bfloat16))
        assert ops._workspace(dev) is ws
        y_again = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, None)
        torch.cuda.synchronize()
        assert int(ws[:16384].view(torch.int32).abs().sum()) == 0
    finally:
        ops.drop_workspace()
    assert torch.equal(y, y_again)
    ref, rms = _device_reference(ops, p, parts)
    tol = 0.51 * 2.0 ** -7 * ref.abs() + 1e-4 * rms
    assert ((y.float() - ref).abs() <= tol).all()


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("fmt", [(256, 4, 1), (256, 2, 1), (16, 2, 1), (4096, 4, 1), (4096, 2, 1), (256, 8, 2)])
@pytest.mark.parametrize("cfg", [0, 2])
def test_tiled_kernel_wide_token_tiles(ops, monkeypatch, cfg, fmt, dtype):
    """The 256 x 256 and 128 x 256 tiles whose waves take 8 token fragments (the formats with several gathers or a sum
    per weight fragment), both dtypes: two token tiles, the second ragged, ragged row tiles, two partitions, bias."""
    K, d, M = fmt
    parts, Kin, T = [272, 144], 1024, 300
    p = Problem(300 + cfg, K, d, M, parts, Kin, T, dtype, bias=True).to(DEV)
    packed = ops.pq_repack_codes(p.codes_dev, p.codebooks_dev, parts)
    assert packed is not None
    monkeypatch.setenv("PQ_AMD_FORCE_KERNEL", "tiled")
    monkeypatch.setenv("PQ_AMD_TILED_CFG", str(cfg))
    _lib.reload_knobs()
    y = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, p.bias_dev)
    y2 = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, p.bias_dev)
    assert torch.equal(y, y2)
    _, y32 = p.oracle_gemm(order=2)
    assert_close_to_exact(bits_of(y), y32, dtype, f"K={K} d={d} M={M} cfg {cfg} {dtype}")


@pytest.mark.parametrize("fmt", [(256, 8, 1), (256, 4, 1), (16, 2, 1), (4096, 8, 1), (4096, 4, 1), (256, 8, 2)])
@pytest.mark.parametrize("cfg", range(7))
def test_tiled_kernel_long_k_every_shape_and_format(ops, monkeypatch, fmt, cfg):
    """Enough K for several trips round the main loop of every tile shape (10 to 40 loop steps: full unrolled
    bodies, the back edge, and both tail lengths over the cases), for every code format."""
    K, d, M = fmt
    parts, Kin, T = [272, 144], 2560, 70
    p = Problem(200 + cfg, K, d, M, parts, Kin, T, "fp16" if cfg % 2 else "bf16", bias=False).to(DEV)
    packed = ops.pq_repack_codes(p.codes_dev, p.codebooks_dev, parts)
    assert packed is not None
    monkeypatch.setenv("PQ_AMD_FORCE_KERNEL", "tiled")
    monkeypatch.setenv("PQ_AMD_TILED_CFG", str(cfg))
    _lib.reload_knobs()
    y = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, None)
    _, y32 = p.oracle_gemm(order=2)
    assert_close_to_exact(bits_of(y), y32, "fp16" if cfg % 2 else "bf16", f"K={K} d={d} M={M} cfg {cfg}")


# ---- the per-rank shapes of the tensor-parallel configurations ------------------------------------------
TP_SHAPES = [  # what one rank holds (column-parallel: rows split; row-parallel: input features split)
    ("8B/tp4.qkv", [1024, 256, 256], 4096), ("8B/tp4.o", [4096], 1024), ("8B/tp4.gate_up", [3584, 3584], 4096),
    ("8B/tp4.down", [4096], 3584), ("70B/tp8.qkv", [1024, 128, 128], 8192), ("70B/tp8.o", [8192], 1024),
    ("70B/tp8.gate_up", [3584, 3584], 8192), ("70B/tp8.down", [8192], 3584)]


@pytest.mark.parametrize("name,parts,Kin", TP_SHAPES, ids=[s[0] for s in TP_SHAPES])
@pytest.mark.parametrize("T", [2, 16, 48, 300])
def test_tp_rank_shapes_against_oracle(ops, name, parts, Kin, T):
    """Llama-3-8B TP=4 and Llama-3-70B TP=8 per-rank linears against the CPU oracle (a sample of the tokens for the
    largest ones keeps the oracle within seconds), packed path, whatever kernel and tile shape the library picks."""
    p = Problem(sum(parts) % 1000 + Kin % 97 + T, 256, 8, 1, parts, Kin, T, "bf16", bias=False).to(DEV)
    packed = ops.pq_repack_codes(p.codes_dev, p.codebooks_dev, parts)
    y = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, None)
    rows = np.arange(T) if T <= 48 else np.arange(0, T, 13)
    a = p.np_args()
    _, y32 = orc.gemm(a["x"][rows], a["codes"], a["codebooks"], a["scales"], parts, None, "bf16", order=2,
                      acc_double=True, want_f32=True)
    assert_close_to_exact(bits_of(y)[rows], y32, "bf16", f"{name} T={T}")


# ---- whatever the dispatcher picks --------------------------------------------------------------------
def _random_problems():
    rng = np.random.RandomState(2024)
    fmts = [(256, 8, 1), (16, 2, 1), (4096, 4, 1), (256, 8, 2), (4096, 8, 1), (16, 8, 1), (256, 4, 1), (4096, 2, 1)]
    out = []
    for i in range(48):
        K, d, M = fmts[i % len(fmts)]
        nparts = int(rng.randint(1, 4))
        parts = [int(rng.randint(1, 40)) * 16 for _ in range(nparts)]
        Kin = int(rng.choice([256, 512, 1024, 1536, 2048, 4096]))
        T = int(rng.choice([1, 2, 3, 7, 16, 17, 24, 33, 48, 64, 100, 129, 200]))
        out.append((i, K, d, M, parts, Kin, T, "fp16" if i % 3 == 0 else "bf16", bool(i % 2)))
    return out


@pytest.mark.parametrize("i,K,d,M,parts,Kin,T,dtype,bias", _random_problems(),
                         ids=lambda v: str(v) if not isinstance(v, list) else "x".join(map(str, v)))
def test_random_problems_default_dispatch(ops, i, K, d, M, parts, Kin, T, dtype, bias):
    """Seeded random formats, partitions, widths and batch sizes through the default kernel choice (tile shape,
    K split, stream or tiled), packed and reference layout, with the K-split workspace in place."""
    p = Problem(7000 + i, K, d, M, parts, Kin, T, dtype, bias=bias).to(DEV)
    N, G = sum(parts), Kin // d
    ops.ensure_workspace(torch.device(DEV), ops.workspace_bytes(128, N, G, M, K, d, p.codes.element_size(),
                                                                p.x.dtype))
    _, y32 = p.oracle_gemm(order=2)
    y = ops.pq_gemm(p.x_dev, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, p.bias_dev)
    assert_close_to_exact(bits_of(y), y32, dtype, f"case {i} reference layout")
    packed = ops.pq_repack_codes(p.codes_dev, p.codebooks_dev, parts)
    if packed is not None:
        y = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, p.bias_dev)
        assert_close_to_exact(bits_of(y), y32, dtype, f"case {i} packed")


@pytest.mark.parametrize("T,bias", [(4, False), (40, False), (40, True), (300, False)])
def test_codebook_too_large_for_lds_takes_the_dequant_route(ops, T, bias):
    """AQLM 1x16 (65536 entries of 8, two-byte codes): no MFMA kernel.  Small batches run the generic fused kernel,
    larger ones dequantise + dense GEMM in the reference's order for that route (aqlm.py:143-156: the GEMM output
    is scaled, or with a bias the weight is); both against the oracle in that order."""
    from vllm_pq_amd import PQConfig
    from vllm_pq_amd.host.linear import MergedColumnParallelLinear
    parts, Kin = [96, 160], 256
    p = Problem(4242 + T, 65536, 8, 1, parts, Kin, T, "fp16", bias=bias)
    layer = MergedColumnParallelLinear(Kin, parts, bias=bias, params_dtype=torch.float16, quant_config=PQConfig(8, 16, 1, 1))
    layer.codes.data, layer.codebooks.data, layer.scales.data = p.codes, p.codebooks, p.scales
    if bias:
        layer.bias.data = p.bias
    layer.to(DEV)
    layer.quant_method.process_weights_after_loading(layer)
    assert layer.pq_packed_codes is None and layer.pq_dequant_route
    y, _ = layer(p.x.to(DEV))
    a = p.np_args()
    if T <= 8:
        _, y32 = p.oracle_gemm(order=2)
        assert_close_to_exact(bits_of(y), y32, "fp16", "generic kernel")
    else:
        ref = orc.gemm(a["x"], a["codes"], a["codebooks"], a["scales"], parts, a["bias"], "fp16", order=1 if not bias else 0)
        assert_close_to_reference_order(bits_of(y), ref, "fp16", "dequant route")


def test_row_length_not_a_multiple_of_8_keeps_the_reference_layout(ops):
    """in_features = 12 (d = 2, 6 groups): no packed layout (rows of x could not start 16-byte aligned), and the
    layer runs on the reference layout through the linear method -- against the oracle."""
    from vllm_pq_amd import PQConfig, PQLinearMethod
    p = Problem(77, 256, 2, 1, [32], 12, 5, "bf16", bias=True).to(DEV)
    assert ops.pq_repack_codes(p.codes_dev, p.codebooks_dev, [32]) is None
    layer = torch.nn.Module()
    m = PQLinearMethod(PQConfig(2, 8, 1, 1))
    m.create_weights(layer, 12, [32], 12, 32, torch.bfloat16)
    layer.codes.data, layer.codebooks.data, layer.scales.data = p.codes_dev, p.codebooks_dev, p.scales_dev
    m.process_weights_after_loading(layer)
    assert layer.pq_packed_codes is None
    y = m.apply(layer, p.x_dev, p.bias_dev)
    _, y32 = p.oracle_gemm(order=2)
    assert_close_to_exact(bits_of(y), y32, "bf16", "unpacked fallback")


def test_min_capability_passes_on_this_device():
    """vLLM compares QuantizationConfig.get_min_capability() with major * 10 + minor of the device
    (vllm/config.py:3333-3342; the ROCm platform reports torch.cuda.get_device_capability): the MI355X must pass."""
    from vllm_pq_amd import PQConfig
    major, minor = torch.cuda.get_device_capability(0)
    assert major * 10 + minor >= PQConfig.get_min_capability(), (major, minor)
    assert "gfx950" in torch.cuda.get_device_properties(0).gcnArchName


# ---- 16-entry codebooks folded in pairs -----------------------------------------------------------------
@pytest.mark.parametrize("d", [2, 4])
@pytest.mark.parametrize("T", [1, 16, 100, 512])
def test_folded_pairs_same_weights_and_outputs(ops, d, T):
    """ops.fold_pairs restates a 16-entry format with 256-entry codebooks of twice the sub-vector length: the
    dequantised weight is the same bit for bit, and the packed GEMM on the folded tensors meets the same bound against
    the fp32 reference as the unfolded launch does (different kernels, same weights)."""
    torch.manual_seed(7 * d + T)
    parts, Kin = [256, 128, 128], 2048
    N, G = sum(parts), Kin // d
    codes = torch.randint(-128, 128, (N, G, 1), dtype=torch.int8, device=DEV)       # stored values reduce mod 16
    cb = torch.randn(len(parts), 16, 1, d, device=DEV).to(torch.bfloat16)
    scales = (torch.rand(N, 1, 1, 1, device=DEV) + 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV).to(torch.bfloat16)
    x = torch.randn(T, Kin, device=DEV).to(torch.bfloat16)
    codes2, cb2 = ops.fold_pairs(codes, cb)
    assert codes2.shape == (N, G // 2, 1) and cb2.shape == (len(parts), 256, 1, 2 * d)
    w = ops.pq_dequant(codes, cb, parts)
    assert torch.equal(ops.pq_dequant(codes2, cb2, parts), w)
    packed = ops.pq_repack_codes(codes, cb, parts)
    packed2 = ops.pq_repack_codes(codes2, cb2, parts)
    assert packed2.numel() * 2 == packed.numel()
    ref = (x.float() @ w.float().t()) * scales.view(1, -1).float() + bias.float()
    tol = 0.51 * 2.0 ** -7 * ref.abs() + 1e-4 * ref.pow(2).mean().sqrt()
    for y in (ops.pq_gemm_packed(x, packed, codes, cb, scales, parts, bias),
              ops.pq_gemm_packed(x, packed2, codes2, cb2, scales, parts, bias)):
        assert ((y.float() - ref).abs() <= tol).all()
    assert ops.fold_pairs(codes[:, : G - 1], cb) is None                            # odd number of groups
    assert ops.fold_pairs(codes, torch.randn(len(parts), 256, 1, d, device=DEV).to(torch.bfloat16)) is None


def test_folded_pairs_through_the_method(ops):
    """A layer with 16-entry codebooks of d = 4: process_weights_after_loading folds it, apply() runs the d = 8
    kernels and returns what the unfolded operator call returns within the rounding of either."""
    from vllm_pq_amd import PQConfig
    from vllm_pq_amd.host.linear import MergedColumnParallelLinear
    qc = PQConfig(4, 4, 1, 1)
    layer = MergedColumnParallelLinear(1024, [512, 512], bias=False, params_dtype=torch.bfloat16, quant_config=qc)
    layer.codes.data = torch.randint(-128, 128, layer.codes.shape, dtype=torch.int8)
    layer.codebooks.data = torch.randn(layer.codebooks.shape).to(torch.bfloat16)
    layer.scales.data = (torch.rand(layer.scales.shape) + 0.5).to(torch.bfloat16)
    layer.to(DEV)
    layer.quant_method.process_weights_after_loading(layer)
    assert layer.pq_folded is not None and layer.pq_folded[1].shape[1:] == (256, 1, 8)
    for T in (3, 16, 300):
        x = torch.randn(T, 1024, device=DEV).to(torch.bfloat16)
        y, _ = layer(x)
        w = ops.pq_dequant(layer.codes, layer.codebooks, [512, 512])
        ref = (x.float() @ w.float().t()) * layer.scales.view(1, -1).float()
        tol = 0.51 * 2.0 ** -7 * ref.abs() + 1e-4 * ref.pow(2).mean().sqrt()
        assert ((y.float() - ref).abs() <= tol).all()
        act = layer.quant_method.apply_silu_and_mul(layer, x)
        want = torch.empty_like(act)
        ops.silu_and_mul(want, y)
        assert (act.float() - want.float()).abs().max() <= 2.0 ** -6 * want.float().abs().max()


@pytest.mark.parametrize("case", [c for c in G.cases("gemm") if c["K"] == 16 and c["d"] in (2, 4) and c["M"] == 1], ids=_id)
def test_folded_pairs_against_the_reference_goldens(ops, case):
    """The reference's own outputs for its 16-entry formats (golden vectors), reached through the folded tensors and
    the packed entry point: the same bounds as the unfolded launch in test_gemm_golden."""
    dt = case["dtype"]
    codes, cb, sc = _golden_tensors(case)
    folded = ops.fold_pairs(codes, cb)
    if folded is None:
        pytest.skip("odd number of groups")
    codes2, cb2 = folded
    packed2 = ops.pq_repack_codes(codes2, cb2, case["parts"])
    assert packed2 is not None
    x = tensor_of(G.get(case, "x"), dt, DEV)
    b = G.get(case, "bias")
    b = None if b is None else tensor_of(b, dt, DEV)
    y = bits_of(ops.pq_gemm_packed(x, packed2, codes2, cb2, sc, case["parts"], b))
    _, y32 = orc.gemm(G.get(case, "x"), G.get(case, "codes"), G.get(case, "codebooks").reshape(-1, case["K"], case["d"]),
                      G.get(case, "scales").reshape(-1), case["parts"], G.get(case, "bias"), dt, order=2,
                      acc_double=True, want_f32=True)
    assert_close_to_exact(y, y32, dt, "vs exact")
    assert_close_to_reference_order(y, G.get(case, "out"), dt, "vs reference output")


# ---- a last partial round of the tiled kernel as a second launch of half-height tiles ------------------------
@pytest.mark.parametrize("parts,split", [([2048, 1024, 512], 3), ([2048, 1024, 512], 4), ([2048, 1024, 512], 6),
                                         ([1536, 1040], 2), ([1536, 1040], 4), ([4096], 5)])
@pytest.mark.parametrize("T,bias,dtype", [(130, True, "bf16"), (512, False, "bf16"), (300, True, "fp16")])
def test_tiled_kernel_split_launch(ops, monkeypatch, parts, split, T, bias, dtype):
    """PQ_AMD_TILED_SPLIT = N: the first N 512-row tiles (in the tile map's order: every partition tiled separately,
    ragged last tiles included) as one launch of 512 x 128 tiles, the remaining rows as a second launch of 256 x 128
    tiles -- every output element written once, the same bits as the single launch (same sums in the same order)."""
    p = Problem(900 + split, 256, 8, 1, parts, 1024, T, dtype, bias=bias).to(DEV)
    packed = ops.pq_repack_codes(p.codes_dev, p.codebooks_dev, parts)
    monkeypatch.setenv("PQ_AMD_FORCE_KERNEL", "tiled")
    monkeypatch.setenv("PQ_AMD_TILED_SPLIT", "0")
    _lib.reload_knobs()
    want = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, p.bias_dev)
    monkeypatch.setenv("PQ_AMD_TILED_SPLIT", str(split))
    _lib.reload_knobs()
    got = ops.pq_gemm_packed(p.x_dev, packed, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, p.bias_dev)
    monkeypatch.delenv("PQ_AMD_TILED_SPLIT")
    monkeypatch.delenv("PQ_AMD_FORCE_KERNEL")
    _lib.reload_knobs()
    assert torch.equal(got, want)
    _, y32 = p.oracle_gemm(order=2)
    assert_close_to_exact(bits_of(got), y32, dtype)


def test_tiled_kernel_split_chosen_by_the_model(ops):
    """gate_up_proj of Llama-3-8B at 2048 tokens (896 tiles of 512 x 128: three and a half rounds of the chip): the
    default dispatch splits the launch; the result is the bits of the unsplit launch."""
    import os
    torch.manual_seed(5)
    parts, Kin, T = [14336, 14336], 4096, 2048
    N, Gq = sum(parts), Kin // 8
    codes = torch.randint(-128, 128, (N, Gq, 1), dtype=torch.int8, device=DEV)
    cb = torch.randn(2, 256, 1, 8, device=DEV).to(torch.bfloat16)
    scales = (torch.rand(N, 1, 1, 1, device=DEV) + 0.5).to(torch.bfloat16)
    x = torch.randn(T, Kin, device=DEV).to(torch.bfloat16)
    packed = ops.pq_repack_codes(codes, cb, parts)
    y = ops.pq_gemm_packed(x, packed, codes, cb, scales, parts, None)
    os.environ["PQ_AMD_TILED_SPLIT"] = "0"
    _lib.reload_knobs()
    try:
        y0 = ops.pq_gemm_packed(x, packed, codes, cb, scales, parts, None)
    finally:
        del os.environ["PQ_AMD_TILED_SPLIT"]
        _lib.reload_knobs()
    assert torch.equal(y, y0)


def test_folded_pairs_with_released_reference_codes(ops):
    """release_reference_codes on a folded layer: the reference-layout codes give their storage back after the
    load-time work, apply() keeps working from the packed folded codes."""
    from vllm_pq_amd import PQConfig
    from vllm_pq_amd.host.linear import RowParallelLinear
    torch.manual_seed(3)
    qc = PQConfig(4, 4, 1, 1, release_reference_codes=True)
    layer = RowParallelLinear(2048, 512, bias=False, params_dtype=torch.bfloat16, quant_config=qc)
    codes = torch.randint(-128, 128, layer.codes.shape, dtype=torch.int8)
    layer.codes.data = codes.clone()
    layer.codebooks.data = torch.randn(layer.codebooks.shape).to(torch.bfloat16)
    layer.scales.data = (torch.rand(layer.scales.shape) + 0.5).to(torch.bfloat16)
    layer.to(DEV)
    w = ops.pq_dequant(layer.codes, layer.codebooks, [512])        # before the codes are released
    layer.quant_method.process_weights_after_loading(layer)
    assert layer.pq_folded is not None and layer.codes.untyped_storage().nbytes() <= 16
    for T in (2, 40):
        x = torch.randn(T, 2048, device=DEV).to(torch.bfloat16)
        y, _ = layer(x)
        ref = (x.float() @ w.float().t()) * layer.scales.view(1, -1).float()
        tol = 0.51 * 2.0 ** -7 * ref.abs() + 1e-4 * ref.pow(2).mean().sqrt()
        assert ((y.float() - ref).abs() <= tol).all()
