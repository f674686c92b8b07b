"""The host mirror's weight loaders and vocabulary shard logic against what the reference's own loaders produced
(tests/golden/loaders_golden.npz / .json, written by tests/golden/make_golden_loaders.py from
vllm/model_executor/layers/linear.py:370-408, :494-640, :873-1058, :1144-1180 and
vocab_parallel_embedding.py:50-53, :139-156, :272-401): the same checkpoint tensors through
vllm_pq_amd.host.linear / vocab_parallel_embedding for TP in {1, 2, 4, 8} and every rank give the same per-rank
parameter tensors bit for bit.  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from vllm_pq_amd import PQConfig
from vllm_pq_amd.host import parallel_state as ps
from vllm_pq_amd.host.linear import (ColumnParallelLinear, MergedColumnParallelLinear, QKVParallelLinear,
                                     RowParallelLinear)
from vllm_pq_amd.host.vocab_parallel_embedding import (ParallelLMHead, VocabParallelEmbedding,
                                                       get_masked_input_and_mask, pad_vocab_size)

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "loaders_golden.json")) as f:
    MAN = json.load(f)
ARR = np.load(os.path.join(HERE, "golden", "loaders_golden.npz"), allow_pickle=False)
S = MAN["shapes"]
DT = torch.float16


def tensor(key):
    a = ARR[key]
    if a.dtype == np.uint16:
        return torch.from_numpy(a.view(np.int16).copy()).view(DT)
    return torch.from_numpy(a.copy())


def ck(name):
    return {p: tensor(f"ck.{name}.{p}") for p in ("codes", "codebooks", "scales", "bias")}


@pytest.fixture
def tp(monkeypatch):
    def set_(rank, size):
        monkeypatch.setattr(ps, "_TP_RANK", rank)
        monkeypatch.setattr(ps, "_TP_SIZE", size)
    return set_


def load3(layer, src, *shard):
    for pname in ("codes", "codebooks", "scales"):
        p = getattr(layer, pname)
        p.weight_loader(p, src[pname], *shard)
    if layer.bias is not None:
        layer.bias.weight_loader(layer.bias, src["bias"], *shard)


def zero(layer):
    for pname in ("codes", "codebooks", "scales", "bias"):
        p = getattr(layer, pname, None)
        if p is not None:
            p.data.zero_()


def same(tag, layer):
    rec = [r for r in MAN["layers"] if r["tag"] == tag][0]
    assert list(layer.codebooks.output_partition_sizes) == rec["output_partition_sizes"], tag
    for pname, shape in rec["params"].items():
        got, want = getattr(layer, pname).data, tensor(f"{tag}.{pname}")
        assert list(got.shape) == shape == list(want.shape), (tag, pname)
        assert got.dtype == want.dtype, (tag, pname)
        assert torch.equal(got.view(torch.int16) if got.dtype == DT else got,
                           want.view(torch.int16) if want.dtype == DT else want), (tag, pname)
    return rec


@pytest.mark.parametrize("size", MAN["tps"])
def test_linear_loaders_reproduce_the_reference(tp, size):
    cfg = PQConfig(S["D"], 8, 1, 1)
    for rank in range(size):
        tp(rank, size)
        tag = f"tp{size}.r{rank}"
        qkv = QKVParallelLinear(S["HID"], S["HEAD"], S["HEADS"], S["KV_HEADS"], bias=True, params_dtype=DT, quant_config=cfg)
        zero(qkv)
        for sid in "qkv":
            load3(qkv, ck(sid), sid)
        rec = same(f"{tag}.qkv", qkv)
        assert (qkv.num_heads, qkv.num_kv_heads, qkv.num_kv_head_replicas) == (
            rec["num_heads"], rec["num_kv_heads"], rec["num_kv_head_replicas"])

        gate_up = MergedColumnParallelLinear(S["HID"], [S["INTER"]] * 2, bias=True, params_dtype=DT, quant_config=cfg)
        zero(gate_up)
        load3(gate_up, ck("gate"), 0)
        load3(gate_up, ck("up"), 1)
        same(f"{tag}.gate_up", gate_up)

        fused = MergedColumnParallelLinear(S["HID"], [S["INTER"]] * 2, bias=False, params_dtype=DT, quant_config=cfg)
        zero(fused)
        g, u = ck("gate"), ck("up")
        for pname in ("codes", "scales", "codebooks"):       # a checkpoint that stores gate and up already fused
            p = getattr(fused, pname)
            p.weight_loader(p, torch.cat([g[pname], u[pname]], dim=0))
        same(f"{tag}.gate_up_fused", fused)

        down = RowParallelLinear(S["INTER"], S["HID"], bias=True, params_dtype=DT, quant_config=cfg)
        zero(down)
        load3(down, ck("down"))
        same(f"{tag}.down", down)

        col = ColumnParallelLinear(S["HID"], S["COL_OUT"], bias=True, params_dtype=DT, quant_config=cfg)
        zero(col)
        load3(col, ck("col"))
        same(f"{tag}.col", col)


@pytest.mark.parametrize("rec", MAN["vocab"], ids=lambda r: r["tag"])
def test_vocabulary_shards_reproduce_the_reference(tp, rec):
    tp(rec["rank"], rec["tp"])
    vocab, org, tag = rec["vocab"], rec["org"], rec["tag"]
    emb = VocabParallelEmbedding(vocab, S["HID"], params_dtype=DT, org_num_embeddings=org)
    assert emb.num_embeddings_padded == rec["num_embeddings_padded"]
    assert emb.org_vocab_size_padded == rec["org_vocab_size_padded"]
    assert emb.num_embeddings_per_partition == rec["num_embeddings_per_partition"]
    for k, v in rec["indices"].items():
        assert getattr(emb.shard_indices, k) == v, k
    assert emb.get_sharded_to_full_mapping() == rec["mapping"]
    emb.weight.data.fill_(7.0)
    emb.weight_loader(emb.weight, tensor(f"ck.vocab{vocab}.weight")[:emb.org_vocab_size])
    assert torch.equal(emb.weight.data.view(torch.int16), tensor(f"{tag}.weight").view(torch.int16))

    # the LM head's PQ codes go through the same loader (the reference's raises on the scales parameter, which is
    # packed along the sharded dimension: recorded in the fixture, see make_golden_loaders.py)
    head_cfg = PQConfig(S["D"], 8, 1, 1)
    head_cfg.lm_head_quantized = True
    head = ParallelLMHead(vocab, S["HID"], params_dtype=DT, org_num_embeddings=org, quant_config=head_cfg)
    head.codes.data.fill_(5)
    head.codes.weight_loader(head.codes, tensor(f"ck.vocab{vocab}.codes")[:head.org_vocab_size])
    assert torch.equal(head.codes.data, tensor(f"{tag}.head_codes"))
    assert rec["head_scales_loader"].startswith("AttributeError")
    head.scales.data.fill_(3.0)
    head.scales.weight_loader(head.scales, torch.ones(head.org_vocab_size, 1, 1, 1, dtype=DT))
    n_org = emb.shard_indices.org_vocab_end_index - emb.shard_indices.org_vocab_start_index
    assert bool((head.scales.data[:n_org] == 1).all()) and not head.scales.data[n_org:].any()

    si = emb.shard_indices
    masked, mask = get_masked_input_and_mask(tensor(f"ck.vocab{vocab}.ids"), si.org_vocab_start_index,
                                             si.org_vocab_end_index, si.num_org_vocab_padding,
                                             si.added_vocab_start_index, si.added_vocab_end_index)
    assert torch.equal(masked, tensor(f"{tag}.masked_ids"))
    assert torch.equal(mask.to(torch.uint8), tensor(f"{tag}.mask"))


def test_pad_vocab_size_reproduces_the_reference():
    for rec in MAN["pad_vocab_size"]:
        assert pad_vocab_size(rec["vocab"], rec["pad_to"]) == rec["out"]
