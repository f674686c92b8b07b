"""Offline quantiser and checkpoint round trip: the tensors it writes are in the layout the loaders and the
oracle read, load into the fused layers through the per-shard weight loaders, and reconstruct the weight."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import pq_oracle as orc
from pq_testlib import bits_of
from vllm_pq_amd import PQConfig
from vllm_pq_amd.host.linear import MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear
from vllm_pq_amd.quantize import (pq_quantize_weight, pq_reconstruct, quantization_config, quantize_state_dict,
                                  save_pq_checkpoint)


def _weight(seed, n, kin, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    # rows of different magnitude so the row scale matters
    return (torch.randn(n, kin, generator=g) * (0.5 + torch.rand(n, 1, generator=g))).to(dtype)


@pytest.mark.parametrize("d,nbits,M,dtype", [(8, 8, 1, torch.bfloat16), (4, 8, 1, torch.float16),
                                             (8, 10, 1, torch.bfloat16), (8, 8, 2, torch.float16),
                                             (2, 4, 1, torch.bfloat16)])
def test_layout_dtypes_and_oracle_agreement(d, nbits, M, dtype):
    w = _weight(1, 96, 128, dtype)
    codes, cb, sc = pq_quantize_weight(w, d, nbits, M, iters=6, seed=3)
    K = 1 << nbits
    assert codes.shape == (96, 128 // d, M) and codes.dtype == (torch.int8 if nbits <= 8 else torch.int16)
    assert cb.shape == (M, K, 1, d) and cb.dtype == dtype
    assert sc.shape == (96, 1, 1, 1) and sc.dtype == dtype
    # the oracle's dequantisation (the reference's formula) of these tensors equals the quantiser's own view
    name = "bf16" if dtype == torch.bfloat16 else "fp16"
    ref = orc.dequant(codes.numpy(), bits_of(cb).reshape(-1, K, d), None, [96], name)
    unscaled = pq_reconstruct(codes, cb, torch.ones_like(sc))
    if M == 1:      # one codebook: the oracle's act-dtype weight is the codebook entry itself
        assert np.array_equal(orc.to_f32(ref, name), unscaled.numpy())
    else:           # the oracle rounds the sum of the codebook entries to the act dtype
        assert np.allclose(orc.to_f32(ref, name), unscaled.numpy(), rtol=2.0 ** -7, atol=1e-6)


def test_reconstruction_error_and_determinism():
    w = _weight(2, 256, 256)
    a = pq_quantize_weight(w, 8, 8, 1, iters=10, seed=5)
    b = pq_quantize_weight(w, 8, 8, 1, iters=10, seed=5)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    rel = lambda q: float((pq_reconstruct(*q) - w.float()).norm() / w.float().norm())
    e1 = rel(a)
    # gaussian weights at 1 bit per weight: k-means lands well under the error of a random codebook (~1.0+)
    assert e1 < 0.75, e1
    assert rel(pq_quantize_weight(w, 8, 8, 2, iters=10, seed=5)) < e1 - 0.05        # residual codebook helps
    assert rel(pq_quantize_weight(w, 4, 8, 1, iters=10, seed=5)) < e1 - 0.05        # 2 bits per weight helps
    assert rel(pq_quantize_weight(w, 8, 8, 1, iters=0, seed=5)) > e1                # Lloyd iterations help
    # a weight that IS a PQ weight is recovered exactly (up to the act-dtype rounding of scale*entry)
    g = torch.Generator().manual_seed(7)
    cb = torch.randn(16, 8, generator=g)
    idx = torch.randint(0, 16, (64, 8), generator=g)
    exact = cb[idx].reshape(64, 64)
    q = pq_quantize_weight(exact, 8, 4, 1, iters=20, seed=0, dtype=torch.float16)
    assert float((pq_reconstruct(*q) - exact).norm() / exact.norm()) < 2e-3


def test_rejects_bad_arguments():
    with pytest.raises(ValueError):
        pq_quantize_weight(torch.zeros(4, 12), 8, 8)
    with pytest.raises(ValueError):
        pq_quantize_weight(torch.zeros(4, 16), 8, 17)
    with pytest.raises(ValueError):
        pq_quantize_weight(torch.zeros(4), 8, 8)
    # all-zero and tiny inputs do not produce NaNs
    q = pq_quantize_weight(torch.zeros(4, 16), 8, 8, iters=2)
    assert torch.isfinite(q[1].float()).all() and torch.isfinite(q[2].float()).all()
    assert torch.equal(pq_reconstruct(*q), torch.zeros(4, 16))


def test_checkpoint_round_trip_through_the_layer_loaders(tmp_path):
    from safetensors.torch import load_file
    H, heads, kv, hd, inter = 64, 4, 2, 16, 96
    state = {
        "model.embed_tokens.weight": _weight(10, 50, H),
        "model.layers.0.self_attn.q_proj.weight": _weight(11, heads * hd, H),
        "model.layers.0.self_attn.k_proj.weight": _weight(12, kv * hd, H),
        "model.layers.0.self_attn.v_proj.weight": _weight(13, kv * hd, H),
        "model.layers.0.self_attn.o_proj.weight": _weight(14, H, heads * hd),
        "model.layers.0.mlp.gate_proj.weight": _weight(15, inter, H),
        "model.layers.0.mlp.up_proj.weight": _weight(16, inter, H),
        "model.layers.0.mlp.down_proj.weight": _weight(17, H, inter),
        "model.layers.0.input_layernorm.weight": torch.ones(H, dtype=torch.bfloat16),
    }
    report = []
    q = quantize_state_dict(state, d=8, nbits=8, iters=4, report=report)
    assert "model.embed_tokens.weight" in q and "model.layers.0.input_layernorm.weight" in q
    assert "model.layers.0.mlp.down_proj.weight" not in q and len(report) == 7
    save_pq_checkpoint(q, str(tmp_path), 8, 8, 1, base_config={"model_type": "llama", "hidden_size": H})
    with open(os.path.join(tmp_path, "config.json")) as f:
        hf = json.load(f)
    assert hf["hidden_size"] == H and hf["quantization_config"] == quantization_config(8, 8, 1)
    cfg = PQConfig.from_config(hf["quantization_config"])
    assert (cfg.in_group_size, cfg.nbits_per_codebook, cfg.num_codebooks, cfg.out_group_size) == (8, 8, 1, 1)

    disk = load_file(os.path.join(tmp_path, "model.safetensors"))
    qkv = QKVParallelLinear(H, hd, heads, kv, bias=False, params_dtype=torch.bfloat16, quant_config=cfg)
    gate_up = MergedColumnParallelLinear(H, [inter, inter], bias=False, params_dtype=torch.bfloat16, quant_config=cfg)
    down = RowParallelLinear(inter, H, bias=False, params_dtype=torch.bfloat16, quant_config=cfg)
    # the stacked-parameter mapping of the reference's llama loader (models/llama.py:371-377)
    mapping = [("self_attn.q_proj", qkv, "q"), ("self_attn.k_proj", qkv, "k"), ("self_attn.v_proj", qkv, "v"),
               ("mlp.gate_proj", gate_up, 0), ("mlp.up_proj", gate_up, 1), ("mlp.down_proj", down, None)]
    for sub, layer, sid in mapping:
        for pname in ("codes", "codebooks", "scales"):
            t = disk[f"model.layers.0.{sub}.{pname}"]
            param = getattr(layer, pname)
            if sid is None:
                param.weight_loader(param, t)
            else:
                param.weight_loader(param, t, sid)

    def layer_weight(layer, parts):
        """Dense weight of a loaded fused layer via the oracle, partition by partition."""
        K = layer.codebooks.shape[1]
        w = orc.dequant(layer.codes.data.numpy(), bits_of(layer.codebooks.data).reshape(-1, K, 8),
                        bits_of(layer.scales.data).reshape(-1), parts, "bf16")
        return orc.to_f32(w, "bf16")

    for layer, parts, names in ((qkv, [heads * hd, kv * hd, kv * hd], ["self_attn.q_proj", "self_attn.k_proj",
                                                                      "self_attn.v_proj"]),
                                (gate_up, [inter, inter], ["mlp.gate_proj", "mlp.up_proj"]),
                                (down, [H], ["mlp.down_proj"])):
        got = layer_weight(layer, parts)
        dense = torch.cat([state[f"model.layers.0.{n}.weight"] for n in names]).float().numpy()
        own = torch.cat([pq_reconstruct(*[q[f"model.layers.0.{n}.{p}"] for p in ("codes", "codebooks", "scales")])
                         for n in names]).numpy()
        # loaded layer == quantiser's reconstruction up to the bf16 rounding of scale*entry ...
        assert np.allclose(got, own, rtol=2.0 ** -7, atol=1e-6)
        # ... and approximates the dense weight (1 bit per weight on gaussian data)
        assert np.linalg.norm(got - dense) / np.linalg.norm(dense) < 0.8


@pytest.mark.gpu
def test_quantised_weight_through_the_layer_on_gpu():
    """Quantise on the GPU, load into a fused layer, run the HIP GEMM: equal to the oracle on the same tensors
    within the GEMM tolerance, and close to the dense layer within the quantisation error."""
    from pq_testlib import assert_close_to_exact
    dev = "cuda:0"
    H, inter = 512, 768
    dense = [_weight(21, inter, H).to(dev), _weight(22, inter, H).to(dev)]
    cfg = PQConfig(8, 8, 1, 1)
    layer = MergedColumnParallelLinear(H, [inter, inter], bias=False, params_dtype=torch.bfloat16, quant_config=cfg)
    for sid, w in enumerate(dense):
        codes, cb, sc = pq_quantize_weight(w, 8, 8, 1, iters=8, seed=sid)
        assert codes.device.type == "cuda"
        for pname, t in (("codes", codes), ("codebooks", cb), ("scales", sc)):
            param = getattr(layer, pname)
            param.weight_loader(param, t.cpu(), sid)
    layer.to(dev)
    layer.quant_method.process_weights_after_loading(layer)
    x = torch.randn(33, H, device=dev).to(torch.bfloat16)
    y, _ = layer(x)
    _, y32 = orc.gemm(bits_of(x), layer.codes.cpu().numpy(), bits_of(layer.codebooks).reshape(-1, 256, 8),
                      bits_of(layer.scales).reshape(-1), [inter, inter], None, "bf16", order=2, want_f32=True)
    assert_close_to_exact(bits_of(y), y32, "bf16", "quantised gate_up")
    y_dense = x.float() @ torch.cat(dense).float().t()
    rel = float((y.float() - y_dense).norm() / y_dense.norm())
    assert rel < 0.8, rel
