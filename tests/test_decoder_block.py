"""One decoder layer's worth of the path end to end on the GPU: RMSNorm -> qkv_proj -> o_proj -> fused add +
RMSNorm -> gate_up_proj + SiLU-and-mul -> down_proj, built from the host layer classes with weights from the
offline quantiser, against an fp64 evaluation of the same chain on the dequantised weights (attention itself is
not part of the path: the value heads stand in for its output).  Checks that the pieces compose: shard ids,
partition-to-codebook mapping, in-place residual handling, the fused and two-step MLP entry."""
import numpy as np
import pytest
import torch

from vllm_pq_amd import PQConfig
from vllm_pq_amd.quantize import pq_quantize_weight, pq_reconstruct

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dense(seed, n, k):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(n, k, generator=g) / k ** 0.5).to(torch.bfloat16)


@pytest.mark.parametrize("T", [3, 16, 70])
def test_decoder_block_composes(T):
    from vllm_pq_amd.host.layers import RMSNorm
    from vllm_pq_amd.host.linear import MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear
    H, heads, kv, hd, inter = 256, 4, 2, 64, 512
    cfg = PQConfig(8, 8, 1, 1)
    dt = torch.bfloat16
    qkv = QKVParallelLinear(H, hd, heads, kv, bias=False, params_dtype=dt, quant_config=cfg)
    o = RowParallelLinear(heads * hd, H, bias=False, params_dtype=dt, quant_config=cfg)
    gate_up = MergedColumnParallelLinear(H, [inter, inter], bias=False, params_dtype=dt, quant_config=cfg)
    down = RowParallelLinear(inter, H, bias=False, params_dtype=dt, quant_config=cfg)
    W = {}
    for name, layer, sid, n, k in (("q", qkv, "q", heads * hd, H), ("k", qkv, "k", kv * hd, H), ("v", qkv, "v", kv * hd, H),
                                   ("o", o, None, H, heads * hd), ("gate", gate_up, 0, inter, H),
                                   ("up", gate_up, 1, inter, H), ("down", down, None, H, inter)):
        q = pq_quantize_weight(_dense(len(name) * 7 + n, n, k), 8, 8, 1, iters=3, seed=1)
        W[name] = pq_reconstruct(*q).double()
        for pname, t in zip(("codes", "codebooks", "scales"), q):
            p = getattr(layer, pname)
            p.weight_loader(p, t) if sid is None else p.weight_loader(p, t, sid)
    n1, n2 = RMSNorm(H, eps=1e-5), RMSNorm(H, eps=1e-5)
    g = torch.Generator().manual_seed(5)
    n1.weight.data = (1 + 0.1 * torch.randn(H, generator=g)).to(dt)
    n2.weight.data = (1 + 0.1 * torch.randn(H, generator=g)).to(dt)
    for m in (qkv, o, gate_up, down, n1, n2):
        m.to(DEV)
    for m in (qkv, o, gate_up, down):
        m.quant_method.process_weights_after_loading(m)
    x = torch.randn(T, H, generator=g).to(dt)

    # ---- the path
    h = x.to(DEV)
    residual = h.clone()
    a = n1(h)
    qkv_out, _ = qkv(a)
    attn = qkv_out[:, heads * hd + kv * hd:].repeat(1, heads // kv)       # value heads stand in for attention
    h2, _ = o(attn.contiguous())
    h2, residual = n2(h2, residual)
    act = gate_up.quant_method.apply_silu_and_mul(gate_up, h2)
    y, _ = down(act)

    # ---- fp64 on the dequantised weights
    def rms(v, w):
        return v / torch.sqrt((v * v).mean(-1, keepdim=True) + 1e-5) * w

    xd = x.double()
    ad = rms(xd, n1.weight.data.cpu().double())
    vd = ad @ W["v"].t()
    h2d = vd.repeat(1, heads // kv) @ W["o"].t()
    resd = h2d + xd
    h3d = rms(resd, n2.weight.data.cpu().double())
    gd, ud = h3d @ W["gate"].t(), h3d @ W["up"].t()
    yd = (gd / (1 + torch.exp(-gd)) * ud) @ W["down"].t()

    # bf16 is rounded after each of the seven steps: a few units of 2^-8 relative to the output's scale
    err = (y.double().cpu() - yd).abs().max() / yd.abs().max()
    assert err < 0.03, float(err)
    rerr = (residual.double().cpu() - resd).abs().max() / resd.abs().max()
    assert rerr < 0.02, float(rerr)
    # the two-step MLP entry gives the same values as the fused one
    from vllm_pq_amd.host.layers import SiluAndMul
    gu, _ = gate_up(h2)
    assert torch.equal(SiluAndMul()(gu), act) or T > 16


def test_ops_capture_into_a_hip_graph():
    """No op allocates through the library, synchronises or reads host state that changes between replays: a
    captured chain (norm -> gate_up + SiLU-and-mul, fused and two-step -> down with a split K) replays with new
    inputs to the values the eager calls give."""
    from pq_testlib import Problem
    from vllm_pq_amd import ops
    T, H, inter = 8, 512, 1024
    gu = Problem(1, 256, 8, 1, [inter, inter], H, T, "bf16").to(DEV)
    dn = Problem(2, 256, 8, 1, [H], 16 * inter, T, "bf16").to(DEV)       # long K: the K-split path
    pg = ops.pq_repack_codes(gu.codes_dev, gu.codebooks_dev, [inter, inter])
    pd = ops.pq_repack_codes(dn.codes_dev, dn.codebooks_dev, [H])
    ops.ensure_workspace(torch.device(DEV), ops.workspace_bytes(128, H, 16 * inter // 8, 1, 256, 8, 1, torch.bfloat16))
    w = torch.ones(H, dtype=torch.bfloat16, device=DEV)
    x = torch.randn(T, H, device=DEV).to(torch.bfloat16)
    wide = torch.randn(T, 16 * inter, device=DEV).to(torch.bfloat16)

    def chain():
        a = torch.empty_like(x)
        ops.rms_norm(a, x, w, 1e-5)
        f = ops.pq_gemm_packed_silu_mul(a, pg, gu.codes_dev, gu.codebooks_dev, gu.scales_dev, None)
        y = ops.pq_gemm_packed(a, pg, gu.codes_dev, gu.codebooks_dev, gu.scales_dev, [inter, inter], None)
        s = torch.empty(T, inter, dtype=y.dtype, device=DEV)
        ops.silu_and_mul(s, y)
        d = ops.pq_gemm_packed(wide, pd, dn.codes_dev, dn.codebooks_dev, dn.scales_dev, [H], None)
        return f, s, d

    chain()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = chain()
    for seed in (1, 2):
        torch.manual_seed(seed)
        x.copy_(torch.randn(T, H, device=DEV).to(torch.bfloat16))
        wide.copy_(torch.randn(T, 16 * inter, device=DEV).to(torch.bfloat16))
        g.replay()
        torch.cuda.synchronize()
        got = [o.clone() for o in outs]
        want = chain()
        torch.cuda.synchronize()
        for a, b in zip(got, want):
            assert torch.equal(a, b)
        assert torch.equal(got[0], got[1])                      # fused == two-step
