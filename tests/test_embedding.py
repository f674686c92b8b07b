"""Vocabulary table and LM head on the PQ method: shard layout and id masking of the host classes (known answers
taken from the reference's own tests, tests/lora/test_layers.py:1082-1290, and its class docstring,
vocab_parallel_embedding.py:169-186), the embedding-row kernel bit for bit against the oracle's dequantised
table, and the head's logits against the oracle GEMM."""
import random

import numpy as np
import pytest
import torch

import vllm_pq_amd
from vllm_pq_amd import PQConfig, PQEmbeddingMethod
from vllm_pq_amd.host import parallel_state
from vllm_pq_amd.host.vocab_parallel_embedding import (ParallelLMHead, UnquantizedEmbeddingMethod,
                                                       VocabParallelEmbedding, compute_logits,
                                                       get_masked_input_and_mask, pad_vocab_size)

from pq_testlib import Problem, assert_close_to_exact, bits_of

DEV = "cuda:0"


# ---- host logic (CPU) -------------------------------------------------------------------------------------
def _with_rank(monkeypatch, rank, size):
    monkeypatch.setattr(parallel_state, "_TP_RANK", rank)
    monkeypatch.setattr(parallel_state, "_TP_SIZE", size)


def test_pad_vocab_size():
    assert pad_vocab_size(1010) == 1024 and pad_vocab_size(1024) == 1024 and pad_vocab_size(1, 64) == 64
    assert pad_vocab_size(128256) == 128256


def test_docstring_layout_of_the_reference(monkeypatch):
    """vocab 1010 + 16 added, padded to 64: 1088 rows; TP=2 puts ids 0..511 and added 1010..1025 on rank 0 (rows
    0..511, 512..527) and ids 512..1009 on rank 1 (rows 0..497, the rest padding)."""
    _with_rank(monkeypatch, 0, 1)
    e = VocabParallelEmbedding(1026, 8, org_num_embeddings=1010)
    assert e.num_embeddings_padded == 1088 and e.org_vocab_size_padded == 1024
    _with_rank(monkeypatch, 0, 2)
    e0 = VocabParallelEmbedding(1026, 8, org_num_embeddings=1010)
    s = e0.shard_indices
    assert (s.org_vocab_start_index, s.org_vocab_end_index, s.added_vocab_start_index, s.added_vocab_end_index) == \
        (0, 512, 1010, 1026)
    assert e0.num_embeddings_per_partition == 544 and s.num_org_vocab_padding == 0 and s.num_added_vocab_padding == 16
    _with_rank(monkeypatch, 1, 2)
    e1 = VocabParallelEmbedding(1026, 8, org_num_embeddings=1010)
    s = e1.shard_indices
    assert (s.org_vocab_start_index, s.org_vocab_end_index, s.added_vocab_start_index, s.added_vocab_end_index) == \
        (512, 1010, 1026, 1026)
    assert s.num_org_vocab_padding == 14 and s.num_added_elements == 0


@pytest.mark.parametrize("tp_size", [1, 2, 4, 8])
@pytest.mark.parametrize("seed", list(range(24)))
def test_shards_tile_the_vocabulary(monkeypatch, tp_size, seed):
    """The reference's own property test (tests/lora/test_layers.py:1082-1159): contiguous, disjoint ranges that
    add up, and get_sharded_to_full_mapping() reorders the gathered rows to id order with the padding last."""
    random.seed(seed)
    vocab = random.randint(4000, 64000)
    added = random.randint(0, 1024)
    org = vocab - added
    last_org, last_added = 0, org
    n_pad = n_org = n_added = 0
    ids = []
    for r in range(tp_size):
        _with_rank(monkeypatch, r, tp_size)
        e = VocabParallelEmbedding(vocab, 1, org_num_embeddings=org)
        s = e.shard_indices
        assert s.org_vocab_start_index == last_org and s.added_vocab_start_index == last_added
        n_pad += s.num_elements_padded
        n_org += s.num_org_elements
        n_added += s.num_added_elements
        ids += list(range(s.org_vocab_start_index, s.org_vocab_end_index)) + [-1] * s.num_org_vocab_padding
        ids += list(range(s.added_vocab_start_index, s.added_vocab_end_index)) + [-1] * s.num_added_vocab_padding
        last_org, last_added = s.org_vocab_end_index, s.added_vocab_end_index
    assert n_pad == e.num_embeddings_padded and n_org == org and n_added == added
    mapping = e.get_sharded_to_full_mapping()
    assert (mapping is None) == (tp_size == 1)
    if mapping is not None:
        re = torch.tensor(ids)[mapping]
        assert re[:vocab].equal(torch.arange(vocab)) and bool((re[vocab:] == -1).all())


def test_masked_input_known_answers():
    """Expected tensors of the reference's test_get_masked_input_and_mask (tests/lora/test_layers.py:1162-1290)."""
    x = torch.arange(12)
    f = lambda a, b, c, d, p: get_masked_input_and_mask(x, a, b, p, c, d)[0].tolist()   # noqa: E731
    assert f(0, 8, 8, 12, 0) == list(range(12))
    assert f(0, 4, 8, 10, 0) == [0, 1, 2, 3, 0, 0, 0, 0, 4, 5, 0, 0]
    assert f(4, 8, 10, 12, 0) == [0, 0, 0, 0, 0, 1, 2, 3, 0, 0, 4, 5]
    assert f(0, 2, 8, 9, 0) == [0, 1, 0, 0, 0, 0, 0, 0, 2, 0, 0, 0]
    assert f(2, 4, 9, 10, 0) == [0, 0, 0, 1, 0, 0, 0, 0, 0, 2, 0, 0]
    assert f(4, 6, 10, 11, 0) == [0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 2, 0]
    assert f(6, 8, 11, 12, 0) == [0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 2]
    assert f(0, 8, 8, 12, 2) == [0, 1, 2, 3, 4, 5, 6, 7, 10, 11, 12, 13]
    assert f(0, 4, 8, 10, 2) == [0, 1, 2, 3, 0, 0, 0, 0, 6, 7, 0, 0]
    assert f(4, 8, 10, 12, 2) == [0, 0, 0, 0, 0, 1, 2, 3, 0, 0, 6, 7]
    _, other = get_masked_input_and_mask(x, 4, 8, 2, 10, 12)
    assert other.tolist() == [True] * 4 + [False] * 4 + [True] * 2 + [False] * 2


def test_config_picks_the_method(monkeypatch):
    _with_rank(monkeypatch, 0, 1)
    plain = PQConfig(8, 8, 1, 1)
    assert isinstance(VocabParallelEmbedding(300, 64, torch.bfloat16, quant_config=plain).quant_method,
                      UnquantizedEmbeddingMethod)
    assert isinstance(ParallelLMHead(300, 64, params_dtype=torch.bfloat16, quant_config=plain).quant_method,
                      UnquantizedEmbeddingMethod)
    both = PQConfig.from_config({"in_group_size": 8, "nbits_per_codebook": 8, "num_codebooks": 1, "out_group_size": 1,
                                 "lm_head": True, "embeddings": True})
    e = VocabParallelEmbedding(300, 64, torch.bfloat16, quant_config=both)
    h = ParallelLMHead(300, 64, bias=True, params_dtype=torch.bfloat16, quant_config=both)
    assert isinstance(e.quant_method, PQEmbeddingMethod) and isinstance(h.quant_method, PQEmbeddingMethod)
    assert tuple(e.codes.shape) == (320, 8, 1) and tuple(e.codebooks.shape) == (1, 256, 1, 8)
    assert tuple(h.scales.shape) == (320, 1, 1, 1) and tuple(h.bias.shape) == (320,)
    only_head = PQConfig(8, 8, 1, 1, lm_head=True)
    assert isinstance(VocabParallelEmbedding(300, 64, torch.bfloat16, quant_config=only_head).quant_method,
                      UnquantizedEmbeddingMethod)
    with pytest.raises(RuntimeError):
        h(torch.zeros(1, dtype=torch.long))


def test_loader_shards_rows_and_zeroes_the_padding(monkeypatch):
    """codes and scales of the whole vocabulary -> this rank's rows, zeros behind them; the codebook whole."""
    qc = PQConfig(4, 8, 1, 1, embeddings=True)
    g = torch.Generator().manual_seed(3)
    vocab, dim = 200, 32
    codes = torch.randint(-128, 128, (vocab, dim // 4, 1), generator=g, dtype=torch.int64).to(torch.int8)
    cb = torch.randn(1, 256, 1, 4, generator=g).to(torch.float16)
    sc = torch.rand(vocab, 1, 1, 1, generator=g).to(torch.float16) + 0.5
    for rank in range(2):
        _with_rank(monkeypatch, rank, 2)
        e = VocabParallelEmbedding(vocab, dim, torch.float16, quant_config=qc)     # 256 padded rows, 128 a rank
        for name, t in (("codes", codes), ("codebooks", cb), ("scales", sc)):
            p = getattr(e, name)
            p.weight_loader(p, t)
        lo, hi = 128 * rank, min(128 * (rank + 1), vocab)
        assert torch.equal(e.codes.data[:hi - lo], codes[lo:hi]) and not e.codes.data[hi - lo:].any()
        assert torch.equal(e.scales.data[:hi - lo], sc[lo:hi]) and not e.scales.data[hi - lo:].any()
        assert torch.equal(e.codebooks.data, cb)


def test_embedding_on_cpu_raises(monkeypatch):
    _with_rank(monkeypatch, 0, 1)
    e = VocabParallelEmbedding(64, 32, torch.bfloat16, quant_config=PQConfig(8, 8, 1, 1, embeddings=True))
    with pytest.raises(RuntimeError):
        e(torch.tensor([1, 2]))


# ---- GPU: the embedding-row kernel and the head ------------------------------------------------------------
@pytest.fixture(scope="module")
def ops():
    from vllm_pq_amd import ops as o
    return o


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", [(256, 8, 1), (256, 4, 1), (16, 2, 1), (4096, 8, 1), (256, 8, 2), (256, 3, 1)])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("scaled", [False, True])
def test_embedding_rows_bit_exact(ops, fmt, dtype, scaled):
    """out[t] == dequantised row ids[t], every bit, against the oracle's table; ids repeat, come unsorted and in
    two dimensions; ids outside the table give zeros."""
    K, d, M = fmt
    parts = [80, 48] if d != 3 else [128]
    Kin = 24 * d
    p = Problem(41, K, d, M, parts, Kin, 1, dtype, scales="normal").to(DEV)
    table = p.oracle_dequant(scaled=scaled)                       # uint16 bits [N][Kin]
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, p.N, (7, 9), generator=g)
    ids[0, 0], ids[0, 1], ids[6, 8] = 0, p.N - 1, 79
    out = ops.pq_embedding(ids.to(DEV), p.codes_dev, p.codebooks_dev, p.scales_dev if scaled else None, parts)
    assert tuple(out.shape) == (7, 9, Kin) and out.dtype == p.codebooks.dtype
    assert np.array_equal(bits_of(out).reshape(-1, Kin), table[ids.reshape(-1).numpy()])
    same = ops.pq_dequant(p.codes_dev, p.codebooks_dev, parts, p.scales_dev if scaled else None)[ids.to(DEV)]
    assert torch.equal(out, same)
    bad = torch.tensor([-1, p.N, 3], device=DEV)
    ob = ops.pq_embedding(bad, p.codes_dev, p.codebooks_dev, None, parts)
    assert not ob[:2].any() and np.array_equal(bits_of(ob[2]), p.oracle_dequant()[3])
    assert ops.pq_embedding(torch.zeros(0, dtype=torch.long, device=DEV), p.codes_dev, p.codebooks_dev, None,
                            parts).shape == (0, Kin)


@pytest.mark.gpu
def test_pq_table_and_head_through_the_layers(ops):
    """Llama-sized rows (hidden 4096) on a small vocabulary: VocabParallelEmbedding.forward == rows of the oracle's
    table; compute_logits on the head == oracle GEMM over the same parameters, padding cut off."""
    parallel_state.destroy_model_parallel()
    vocab, hidden, T = 1000, 4096, 37
    p = Problem(77, 256, 8, 1, [vocab], hidden, T, "bf16", bias=True).to(DEV)
    qc = PQConfig(8, 8, 1, 1, lm_head=True, embeddings=True)
    emb = VocabParallelEmbedding(vocab, hidden, torch.bfloat16, quant_config=qc)
    head = ParallelLMHead(vocab, hidden, bias=True, params_dtype=torch.bfloat16, quant_config=qc)
    for layer in (emb, head):
        for name, t in (("codes", p.codes), ("codebooks", p.codebooks), ("scales", p.scales)):
            prm = getattr(layer, name)
            prm.weight_loader(prm, t)
        layer.to(DEV)
        layer.quant_method.process_weights_after_loading(layer)
    head.bias.weight_loader(head.bias, p.bias)
    assert emb.codes.shape[0] == 1024 and not emb.scales.data[vocab:].any()
    ids = torch.randint(0, vocab, (T,), generator=torch.Generator().manual_seed(1))
    out = emb(ids.to(DEV))
    assert np.array_equal(bits_of(out), p.oracle_dequant(scaled=True)[ids.numpy()])
    logits = compute_logits(head, p.x_dev, head.bias)
    assert tuple(logits.shape) == (T, vocab)
    _, y32 = p.oracle_gemm(order=2)
    assert_close_to_exact(bits_of(logits), y32, "bf16", "lm_head logits")


@pytest.mark.gpu
def test_embedding_op_traces_with_fake_tensors(ops):
    from torch._subclasses.fake_tensor import FakeTensorMode
    p = Problem(2, 256, 8, 1, [64], 64, 1, "bf16")
    with FakeTensorMode():
        ids = torch.empty((3, 5), dtype=torch.long, device=DEV)
        codes = torch.empty(p.codes.shape, dtype=p.codes.dtype, device=DEV)
        cb = torch.empty(p.codebooks.shape, dtype=p.codebooks.dtype, device=DEV)
        out = torch.ops.pq_amd.embedding(ids, codes, cb, None, [64])
        assert tuple(out.shape) == (3, 5, 64) and out.dtype == torch.bfloat16
