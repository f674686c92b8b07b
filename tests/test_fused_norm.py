"""The fused add + RMSNorm carried by the PQ linears either side of it (pq_gemm_packed_add_norm /
pq_gemm_packed_normed) against the three separate steps -- pq_gemm_packed, fused_add_rms_norm, pq_gemm_packed, each
of which is checked against its own oracle elsewhere -- and against the exact value of the chain in float64.

The roundings differ by construction (include/pq_amd.h): the reference rounds the first GEMM's output, the sum, x / rms
and (x / rms) * w to the act dtype; the fused pair rounds r = y + residual once, r * w once and applies 1 / rms to the
second GEMM's fp32 sum.  Stated tolerance: the new residual is the exact sum rounded once, and differs from the three-step one by no more than that one's two roundings; the second GEMM's
output within 4 ulp of max(|y|, rms(y)) of the three-step one element-wise, and no further from the exact chain (in
RMS) than the three-step result is, plus 10 %."""
import numpy as np
import pytest
import torch

from pq_testlib import Problem

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EPS = 1e-5


def _layer(seed, parts, kin, dtype, bias):
    p = Problem(seed, 256, 8, 1, parts, kin, 1, dtype, bias=bias).to(DEV)
    return p


def _chain(ops, T, H, kin1, parts2, dtype, bias, seed):
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    p1 = _layer(seed, [H], kin1, dtype, bias)
    p2 = _layer(seed + 1, parts2, H, dtype, bias)
    g = torch.Generator(device="cpu").manual_seed(seed + 2)
    x = torch.randn(T, kin1, generator=g).to(tdt).to(DEV)
    res = torch.randn(T, H, generator=g).to(tdt).to(DEV)
    w = (torch.rand(H, generator=g) + 0.5).to(tdt).to(DEV)
    pk1 = ops.pq_repack_codes(p1.codes_dev, p1.codebooks_dev, [H])
    pk2 = ops.pq_repack_codes(p2.codes_dev, p2.codebooks_dev, parts2)
    ops.ensure_workspace(torch.device(DEV))
    # three steps
    h = ops.pq_gemm_packed(x, pk1, p1.codes_dev, p1.codebooks_dev, p1.scales_dev, [H], p1.bias_dev)
    res_u = res.clone()
    ops.fused_add_rms_norm(h, res_u, w, EPS)          # h <- norm(h + res) * w, res_u <- h + res
    y_u = ops.pq_gemm_packed(h, pk2, p2.codes_dev, p2.codebooks_dev, p2.scales_dev, parts2, p2.bias_dev)
    # fused pair
    assert ops.gemm_add_norm_supported(T, p1.codes_dev, p1.codebooks_dev, [H])
    assert ops.gemm_normed_supported(T, p2.codes_dev, p2.codebooks_dev, parts2)
    res_f = res.clone()
    xw, ssq = ops.pq_gemm_packed_add_norm(x, pk1, p1.codes_dev, p1.codebooks_dev, p1.scales_dev, [H], p1.bias_dev, res_f, w)
    y_f = ops.pq_gemm_packed_normed(xw, ssq, EPS, pk2, p2.codes_dev, p2.codebooks_dev, p2.scales_dev, parts2, p2.bias_dev)
    # exact chain in float64 from the bit-exact dequantised weights
    w1 = ops.pq_dequant(p1.codes_dev, p1.codebooks_dev, [H]).double() * p1.scales_dev.view(-1, 1).double()
    w2 = ops.pq_dequant(p2.codes_dev, p2.codebooks_dev, parts2).double() * p2.scales_dev.view(-1, 1).double()
    r = x.double() @ w1.t() + (p1.bias_dev.double() if bias else 0.0) + res.double()
    xn = r * torch.rsqrt(r.pow(2).mean(dim=-1, keepdim=True) + EPS) * w.double()
    y_x = xn @ w2.t() + (p2.bias_dev.double() if bias else 0.0)
    return res_u, res_f, y_u, y_f, r, y_x, ssq, res


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("T,H,kin1,parts2,bias", [
    (1, 1024, 2048, [1536, 1536], False), (7, 1024, 2048, [1536, 1536], True), (16, 2048, 512, [256, 64, 64], False),
    (40, 1024, 1536, [512], True), (128, 512, 1024, [1024, 1024], False)])
def test_fused_add_norm_pair_small(dtype, T, H, kin1, parts2, bias):
    from vllm_pq_amd import ops
    res_u, res_f, y_u, y_f, r, y_x, ssq, res = _chain(ops, T, H, kin1, parts2, dtype, bias, 4000 + T)
    _check(dtype, res_u, res_f, y_u, y_f, r, y_x, ssq, T, res)


@pytest.mark.parametrize("kin1,parts2", [(4096, [14336, 14336]), (14336, [4096, 1024, 1024])])
@pytest.mark.parametrize("T", [3, 16])
def test_fused_add_norm_pair_llama3_8b(kin1, parts2, T):
    """o_proj -> post_attention_layernorm -> gate_up_proj, and down_proj (K split over blocks) -> input_layernorm ->
    qkv_proj, at full size."""
    from vllm_pq_amd import ops
    res_u, res_f, y_u, y_f, r, y_x, ssq, res = _chain(ops, T, 4096, kin1, parts2, "bf16", False, 5000 + T)
    _check("bf16", res_u, res_f, y_u, y_f, r, y_x, ssq, T, res)


@pytest.mark.parametrize("kin1,parts2", [(3584, [7168, 7168]), (3584, [1024, 128, 128])])
@pytest.mark.parametrize("T", [5, 16])
def test_fused_add_norm_pair_hidden_8192(kin1, parts2, T):
    """The per-rank shapes of Llama-3-70B at TP = 8 run as one rank (hidden 8192: the GEMM behind the norm is split
    along K over two blocks, and a thread adds sixteen fragments' sums of squares for its token)."""
    from vllm_pq_amd import ops
    res_u, res_f, y_u, y_f, r, y_x, ssq, res = _chain(ops, T, 8192, kin1, parts2, "bf16", False, 6000 + T)
    _check("bf16", res_u, res_f, y_u, y_f, r, y_x, ssq, T, res)


def _check(dtype, res_u, res_f, y_u, y_f, r, y_x, ssq, T, res):
    ulp = 2.0 ** -7 if dtype == "bf16" else 2.0 ** -10       # spacing relative to a value's power of two
    # new residual: one rounding of the exact sum (fused) against two (three steps): each within its own bound
    rf, ru, rx = res_f.double(), res_u.double(), r
    assert ((rf - rx).abs() <= 0.51 * ulp * rx.abs() + 2e-5 * rx.pow(2).mean().sqrt()).all(), "fused residual is not the rounded exact sum"
    # the three-step residual carries the rounding of the first GEMM's output h = r - residual (half a spacing of h, and
    # next to a power of two a spacing is up to 2 ulp of the smaller neighbour) besides its own
    h_abs = (rx - res.double()).abs()
    assert ((rf - ru).abs() <= 1.01 * ulp * (h_abs + 2 * rx.abs())).all(), "residuals further apart than their roundings allow"
    # the per-fragment sums of squares add up to the row sums of r^2
    tot = ssq.double().sum(dim=0)[:T]
    assert torch.allclose(tot, rx.pow(2).sum(dim=-1), rtol=2e-4), "sums of squares"
    # second GEMM
    yf, yu = y_f.double(), y_u.double()
    rms = y_x.pow(2).mean().sqrt()
    lim = 4 * ulp * torch.maximum(y_x.abs(), rms.expand_as(y_x))
    bad = (yf - yu).abs() > lim
    assert not bad.any(), f"{int(bad.sum())} of {bad.numel()} outputs more than 4 ulp from the three-step result"
    err_f = (yf - y_x).pow(2).mean().sqrt()
    err_u = (yu - y_x).pow(2).mean().sqrt()
    assert err_f <= 1.1 * err_u + 1e-6 * rms, f"fused RMS error {float(err_f):.3g} against {float(err_u):.3g} of the three steps"


def test_norm_pair_supported_through_the_method():
    """The plugin-level query: supported for a Llama-3-8B pair at decode sizes, not beyond 128 tokens, not for a
    hidden size whose sums a block cannot add up (16384; 8192 is within reach), not across a tensor-parallel
    row-parallel layer."""
    from vllm_pq_amd import PQConfig
    from vllm_pq_amd.host.linear import MergedColumnParallelLinear, RowParallelLinear
    qc = PQConfig(8, 8, 1, 1)

    def make(cls, *a):
        layer = cls(*a, bias=False, params_dtype=torch.bfloat16, quant_config=qc)
        layer.codes.data = torch.randint(-128, 128, layer.codes.shape, dtype=torch.int8)
        layer.codebooks.data = torch.randn(layer.codebooks.shape).to(torch.bfloat16)
        layer.scales.data = torch.ones(layer.scales.shape, dtype=torch.bfloat16)
        layer.to(DEV)
        layer.quant_method.process_weights_after_loading(layer)
        return layer
    o = make(RowParallelLinear, 1024, 1024)
    gu = make(MergedColumnParallelLinear, 1024, [512, 512])
    m = o.quant_method
    assert m.norm_pair_supported(o, gu, 16) and m.norm_pair_supported(o, gu, 128)
    assert not m.norm_pair_supported(o, gu, 129)
    big = make(RowParallelLinear, 512, 8192)
    gu8 = make(MergedColumnParallelLinear, 8192, [64, 64])
    assert m.norm_pair_supported(big, gu8, 16)
    huge = make(RowParallelLinear, 512, 16384)
    gu16 = make(MergedColumnParallelLinear, 16384, [64, 64])
    assert not m.norm_pair_supported(huge, gu16, 16)
    # preferred: whenever supported up to 48 tokens; beyond, only where the plain dispatch runs both layers on the
    # decode kernel anyway
    assert m.norm_pair_preferred(o, gu, 16) and m.norm_pair_preferred(o, gu, 48)
    assert not m.norm_pair_preferred(o, gu, 129)
    from vllm_pq_amd import ops

    def plain(layer, t):
        n, g, _ = layer.codes.shape
        parts = list(layer.codebooks.output_partition_sizes)
        return ops.gemm_kernel_name(t, n, g, 1, 256, 8, 1, torch.bfloat16, parts, packed=True)
    for t in (64, 128):
        both = plain(o, t) == "mfma_decode_packed" and plain(gu, t) == "mfma_decode_packed"
        assert m.norm_pair_preferred(o, gu, t) == both
    o8 = make(RowParallelLinear, 4096, 4096)
    gu8 = make(MergedColumnParallelLinear, 4096, [14336, 14336])
    assert m.norm_pair_supported(o8, gu8, 128) and not m.norm_pair_preferred(o8, gu8, 128)
    assert m.norm_pair_preferred(o8, gu8, 32)
    res = torch.randn(5, 1024, device=DEV).to(torch.bfloat16)
    w = torch.ones(1024, device=DEV, dtype=torch.bfloat16)
    x = torch.randn(5, 1024, device=DEV).to(torch.bfloat16)
    want_h, _ = o(x)
    want_res = (want_h.float() + res.float()).to(torch.bfloat16)
    xw, ssq = m.apply_add_norm(o, x, res, w)
    assert (res.float() - want_res.float()).abs().max() <= 2.0 ** -7 * want_res.float().abs().max()
    y = m.apply_normed(gu, xw, ssq, 1e-5)
    assert y.shape == (5, 1024) and torch.isfinite(y).all()


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("T,H,I,bias", [(1, 1024, 1536, False), (7, 1024, 176, True), (16, 4096, 14336, False),
                                        (3, 4096, 14336, True), (40, 1536, 512, True), (128, 512, 1024, False),
                                        (16, 2048, 48, False)])
def test_normed_gate_up_with_activation(dtype, T, H, I, bias):
    """pq_gemm_packed_normed_silu_mul against pq_gemm_packed_normed followed by silu_and_mul: bit for bit (the same
    kernel, the same order of sums -- a layer of at most 4096 inputs is never split along K -- the same roundings)."""
    from vllm_pq_amd import ops
    if dtype == "fp16" and H == 4096:
        pytest.skip("full-size case once")
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    p = _layer(6000 + T, [I, I], H, dtype, bias)
    pk = ops.pq_repack_codes(p.codes_dev, p.codebooks_dev, [I, I])
    g = torch.Generator(device="cpu").manual_seed(6100 + T)
    xw = torch.randn(T, H, generator=g).to(tdt).to(DEV)
    Tp = (T + 15) // 16 * 16
    ssq = (torch.rand(H // 16, Tp, generator=g) * 16 + 4).to(DEV)
    assert ops.gemm_normed_supported(T, p.codes_dev, p.codebooks_dev, [I, I])
    assert ops.gemm_normed_silu_mul_supported(T, p.codes_dev, p.codebooks_dev)
    y = ops.pq_gemm_packed_normed(xw, ssq, EPS, pk, p.codes_dev, p.codebooks_dev, p.scales_dev, [I, I], p.bias_dev)
    want = torch.empty(T, I, dtype=tdt, device=DEV)
    ops.silu_and_mul(want, y)
    got = ops.pq_gemm_packed_normed_silu_mul(xw, ssq, EPS, pk, p.codes_dev, p.codebooks_dev, p.scales_dev, p.bias_dev)
    assert got.shape == (T, I)
    gb, wb = got.view(torch.int16), want.view(torch.int16)          # bit patterns: NaN (inf * 0 in fp16) compares too
    assert torch.equal(gb, wb), f"{int((gb != wb).sum())} of {got.numel()} values differ"


def test_normed_gate_up_with_activation_limits():
    """Not for more than 128 tokens, codebooks of other sizes, or layers of more than 4096 inputs; the method falls back
    to the two steps there."""
    from vllm_pq_amd import ops, PQConfig
    from vllm_pq_amd.host.linear import MergedColumnParallelLinear
    p = _layer(6500, [256, 256], 1024, "bf16", False)
    assert ops.gemm_normed_silu_mul_supported(128, p.codes_dev, p.codebooks_dev)
    assert not ops.gemm_normed_silu_mul_supported(129, p.codes_dev, p.codebooks_dev)
    p16 = Problem(6501, 16, 8, 1, [256, 256], 1024, 1, "bf16").to(DEV)
    assert not ops.gemm_normed_silu_mul_supported(4, p16.codes_dev, p16.codebooks_dev)
    big = _layer(6502, [64, 64], 8192, "bf16", False)
    assert not ops.gemm_normed_silu_mul_supported(4, big.codes_dev, big.codebooks_dev)
    # through the method, on a layer the library fuses and on one it does not (16 entries)
    for prob, K_bits in ((p, 8), (p16, 4)):
        layer = MergedColumnParallelLinear(1024, [256, 256], bias=False, params_dtype=torch.bfloat16,
                                           quant_config=PQConfig(8, K_bits, 1, 1))
        layer.codes.data, layer.codebooks.data, layer.scales.data = prob.codes, prob.codebooks, prob.scales
        layer.to(DEV)
        m = layer.quant_method
        m.process_weights_after_loading(layer)
        xw = torch.randn(5, 1024, device=DEV).to(torch.bfloat16)
        ssq = torch.rand(64, 16, device=DEV) * 16 + 4
        if not ops.gemm_normed_supported(5, layer.codes, layer.codebooks, [256, 256]):
            continue
        y = m.apply_normed(layer, xw, ssq, EPS)
        want = torch.empty(5, 256, dtype=torch.bfloat16, device=DEV)
        ops.silu_and_mul(want, y)
        assert torch.equal(m.apply_normed_silu_and_mul(layer, xw, ssq, EPS).view(torch.int16), want.view(torch.int16))


def _random_act_cases(n=24, seed=77):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        H = int(rng.choice([512, 1024, 1536, 2048, 2560, 3072, 3584, 4096]))
        I = 16 * int(rng.integers(1, 130))
        T = int(rng.choice([1, 2, 3, 5, 8, 9, 15, 16, 17, 31, 33, 64, 100, 128]))
        out.append((i, T, H, I, bool(rng.integers(0, 2)), "bf16" if rng.integers(0, 2) else "fp16"))
    return out


@pytest.mark.parametrize("i,T,H,I,bias,dtype", _random_act_cases(), ids=lambda v: str(v))
def test_normed_gate_up_with_activation_random_shapes(i, T, H, I, bias, dtype):
    """Seeded random layer sizes (ragged row tiles, every chunk count up to 8), token counts across the token
    groups, both dtypes: the fused launch against the two steps, bit for bit; and the plain fused entry at up to 16
    tokens against the same two steps without the norm (within the layers' tolerance where the unfused GEMM runs on
    another kernel)."""
    from vllm_pq_amd import ops
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    p = _layer(7000 + i, [I, I], H, dtype, bias)
    pk = ops.pq_repack_codes(p.codes_dev, p.codebooks_dev, [I, I])
    g = torch.Generator(device="cpu").manual_seed(7100 + i)
    xw = (torch.randn(T, H, generator=g) * 0.25).to(tdt).to(DEV)
    ssq = (torch.rand(H // 16, (T + 15) // 16 * 16, generator=g) * 16 + 4).to(DEV)
    ops.ensure_workspace(torch.device(DEV))
    assert ops.gemm_normed_silu_mul_supported(T, p.codes_dev, p.codebooks_dev)
    y = ops.pq_gemm_packed_normed(xw, ssq, EPS, pk, p.codes_dev, p.codebooks_dev, p.scales_dev, [I, I], p.bias_dev)
    want = torch.empty(T, I, dtype=tdt, device=DEV)
    ops.silu_and_mul(want, y)
    got = ops.pq_gemm_packed_normed_silu_mul(xw, ssq, EPS, pk, p.codes_dev, p.codebooks_dev, p.scales_dev, p.bias_dev)
    gb, wb = got.view(torch.int16), want.view(torch.int16)
    assert torch.equal(gb, wb), f"{int((gb != wb).sum())} of {got.numel()} values differ"
    if T <= 16:
        y2 = ops.pq_gemm_packed(xw, pk, p.codes_dev, p.codebooks_dev, p.scales_dev, [I, I], p.bias_dev)
        want2 = torch.empty(T, I, dtype=tdt, device=DEV)
        ops.silu_and_mul(want2, y2)
        got2 = ops.pq_gemm_packed_silu_mul(xw, pk, p.codes_dev, p.codebooks_dev, p.scales_dev, p.bias_dev)
        ok = torch.isfinite(want2.float()) & torch.isfinite(got2.float())
        err = (got2.float() - want2.float()).abs()[ok]
        lim = (2.0 ** -6 if dtype == "bf16" else 2.0 ** -9) * want2.float().abs()[ok] + 1e-3
        assert (err <= lim).all(), f"{int((err > lim).sum())} values beyond two ulp"


@pytest.mark.parametrize("T,H,parts", [(5, 1024, [256, 256]), (40, 2048, [512, 512]), (128, 4096, [1024, 1024]),
                                       (23, 1536, [176, 176])])
def test_normed_gemm_does_not_depend_on_the_batch(T, H, parts):
    """A token's output of the GEMM behind the norm (with and without the activation) is the same bit pattern whether
    it is computed alone or inside a batch: the block shape changes with the batch size, the order in which the
    token's sum of squares and its partial sums are added does not."""
    from vllm_pq_amd import ops
    p = _layer(8000 + T, parts, H, "bf16", True)
    pk = ops.pq_repack_codes(p.codes_dev, p.codebooks_dev, parts)
    g = torch.Generator(device="cpu").manual_seed(8100 + T)
    xw = torch.randn(T, H, generator=g).to(torch.bfloat16).to(DEV)
    ssq = (torch.rand(H // 16, (T + 15) // 16 * 16, generator=g) * 16 + 4).to(DEV)
    full = ops.pq_gemm_packed_normed(xw, ssq, EPS, pk, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, p.bias_dev)
    full_act = ops.pq_gemm_packed_normed_silu_mul(xw, ssq, EPS, pk, p.codes_dev, p.codebooks_dev, p.scales_dev, p.bias_dev)
    for t in (0, T // 2, T - 1):
        s1 = torch.zeros(H // 16, 16, device=DEV)
        s1[:, 0] = ssq[:, t]
        x1 = xw[t:t + 1].contiguous()
        one = ops.pq_gemm_packed_normed(x1, s1, EPS, pk, p.codes_dev, p.codebooks_dev, p.scales_dev, parts, p.bias_dev)
        assert torch.equal(one.view(torch.int16), full[t:t + 1].view(torch.int16)), f"token {t}"
        one = ops.pq_gemm_packed_normed_silu_mul(x1, s1, EPS, pk, p.codes_dev, p.codebooks_dev, p.scales_dev, p.bias_dev)
        assert torch.equal(one.view(torch.int16), full_act[t:t + 1].view(torch.int16)), f"token {t} (activation)"
