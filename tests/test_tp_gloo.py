"""Tensor-parallel path on 2 CPU processes (gloo): the attribute-driven loaders shard codes / scales /
codebooks as the reference's do (linear.py:370-408, :494-640, :873-1058, :1144-1180), the row-parallel
layer adds the bias on rank 0 only and all-reduces (linear.py:1206-1213), and the sharded result equals
the unsharded one.  The PQ method's apply() is replaced by the CPU oracle here (test infrastructure):
the GPU kernels are covered by tests/test_gpu_parity.py, the collective path by this file."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle.torch_restatement import pq_linear_cpu, reconstruct_partition
from vllm_pq_amd import PQConfig, PQEmbeddingMethod, PQLinearMethod
from vllm_pq_amd.host import parallel_state as ps
from vllm_pq_amd.host.linear import (ColumnParallelLinear, MergedColumnParallelLinear, QKVParallelLinear,
                                     RowParallelLinear)
from vllm_pq_amd.host.vocab_parallel_embedding import ParallelLMHead, VocabParallelEmbedding, compute_logits


class OraclePQMethod(PQLinearMethod):
    """PQLinearMethod with the CPU oracle as apply(): everything but the kernel is the product code."""

    def apply(self, layer, x, bias=None):
        return pq_linear_cpu(x, layer.codes, layer.codebooks, layer.scales,
                             layer.codebooks.output_partition_sizes, bias)


class OraclePQEmbeddingMethod(PQEmbeddingMethod):
    """The vocabulary-table method with the CPU oracle in place of both kernels."""

    def apply(self, layer, x, bias=None):
        return pq_linear_cpu(x, layer.codes, layer.codebooks, layer.scales, [layer.codes.shape[0]], bias)

    def embedding(self, layer, input_):
        return reconstruct_partition(layer.codes, layer.codebooks, layer.scales)[input_]


class OraclePQConfig(PQConfig):
    def get_quant_method(self, layer, prefix):
        if isinstance(layer, VocabParallelEmbedding):
            return OraclePQEmbeddingMethod(self)
        return OraclePQMethod(self)


VOCAB = 150      # padded to 192: 96 rows a rank, 42 of rank 1's are padding


HID, INTER, HEADS, KV, HD, D = 64, 96, 4, 1, 16, 8
DT = torch.float16


def full_weights():
    g = torch.Generator().manual_seed(7)

    def mat(n, kin):
        return dict(codes=torch.randint(-128, 128, (n, kin // D, 1), generator=g, dtype=torch.int8),
                    codebooks=torch.randn(1, 256, 1, D, generator=g).to(DT),
                    scales=(torch.rand(n, 1, 1, 1, generator=g) + 0.5).to(DT))

    w = {name: mat(n, kin) for name, n, kin in (("q", HEADS * HD, HID), ("k", KV * HD, HID), ("v", KV * HD, HID),
                                                ("o", HID, HEADS * HD), ("gate", INTER, HID), ("up", INTER, HID),
                                                ("down", HID, INTER), ("col", 48, HID))}
    w["vocab"] = mat(VOCAB, HID)
    w["ids"] = torch.randint(0, VOCAB, (3, 7), generator=g)
    w["down_bias"] = torch.randn(HID, generator=g).to(DT)
    w["x"] = torch.randn(5, HID, generator=g).to(DT)
    return w


def load(layer, w, shard_id=None):
    for pname in ("codes", "codebooks", "scales"):
        p = getattr(layer, pname)
        if shard_id is None:
            p.weight_loader(p, w[pname])
        else:
            p.weight_loader(p, w[pname], shard_id)


def dense(w, x, bias=None):
    return pq_linear_cpu(x, w["codes"], w["codebooks"], w["scales"], [w["codes"].shape[0]], bias)


def worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ps.initialize_model_parallel(world)
    try:
        torch.set_num_threads(1)
        cfg = OraclePQConfig(D, 8, 1, 1)
        w = full_weights()
        x = w["x"]
        res = {}

        qkv = QKVParallelLinear(HID, HD, HEADS, KV, bias=False, params_dtype=DT, quant_config=cfg)
        assert qkv.num_kv_head_replicas == world and qkv.num_kv_heads == 1      # 1 kv head replicated
        for sid in "qkv":
            load(qkv, w[sid], sid)
        y, _ = qkv(x)                                      # [T, (heads/tp + 2) * HD] per rank
        res["qkv"] = y

        gate_up = MergedColumnParallelLinear(HID, [INTER, INTER], bias=False, params_dtype=DT, quant_config=cfg)
        load(gate_up, w["gate"], 0)
        load(gate_up, w["up"], 1)
        h, _ = gate_up(x)
        res["gate_up"] = h

        down = RowParallelLinear(INTER, HID, bias=True, params_dtype=DT, quant_config=cfg)
        load(down, w["down"])
        down.bias.weight_loader(down.bias, w["down_bias"])
        xs = torch.randn(5, INTER, generator=torch.Generator().manual_seed(3)).to(DT)
        shard = INTER // world
        y, _ = down(xs[:, rank * shard:(rank + 1) * shard].contiguous())      # input_is_parallel
        res["down"] = y
        down_split = RowParallelLinear(INTER, HID, bias=True, input_is_parallel=False, params_dtype=DT, quant_config=cfg)
        load(down_split, w["down"])
        down_split.bias.weight_loader(down_split.bias, w["down_bias"])
        res["down_split"] = down_split(xs)[0]

        col = ColumnParallelLinear(HID, 48, bias=False, gather_output=True, params_dtype=DT, quant_config=cfg)
        load(col, w["col"])
        res["col_gathered"] = col(x)[0]

        # vocabulary-parallel PQ table and head: ids of the other rank are masked out and the partial lookups summed
        # (vocab_parallel_embedding.py:403-422); the head's shards are gathered and the padding cut off
        emb = VocabParallelEmbedding(VOCAB, HID, DT, quant_config=cfg)
        head = ParallelLMHead(VOCAB, HID, params_dtype=DT, quant_config=cfg)
        for layer in (emb, head):
            load(layer, w["vocab"])
        assert emb.codes.shape[0] == 96
        res["emb"] = emb(w["ids"])
        res["logits_shard"] = head.quant_method.apply(head, x)
        res["logits"] = compute_logits(head, x)

        # known-answer all-reduce, as the reference's comm-op test does (tests/distributed/test_comm_ops.py:30-38)
        t = torch.arange(8, dtype=torch.float32) * (rank + 1)
        res["allreduce"] = ps.tensor_model_parallel_all_reduce(t)
        assert torch.equal(t, torch.arange(8, dtype=torch.float32) * (rank + 1))     # out of place
        own = t.clone()                                                              # in place for a tensor the caller owns
        assert ps.tensor_model_parallel_all_reduce(own, in_place=True) is own
        res["allreduce_in_place"] = own
        torch.save(res, os.path.join(out, f"rank{rank}.pt"))
    finally:
        ps.destroy_model_parallel()
        dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(180)
def test_two_rank_tensor_parallel_matches_unsharded(tmp_path):
    world = 2
    mp.spawn(worker, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"rank{i}.pt", weights_only=True) for i in range(world)]
    w = full_weights()
    x = w["x"]
    # fp16 outputs of sums taken in different orders: a few ulps
    tol = dict(rtol=4e-3, atol=4e-3 * 8)

    q_full, k_full, v_full = dense(w["q"], x), dense(w["k"], x), dense(w["v"], x)
    qs = HEADS * HD // world
    for rank in range(world):
        y = r[rank]["qkv"]
        assert y.shape == (5, qs + 2 * HD)
        assert torch.equal(y[:, :qs], q_full[:, rank * qs:(rank + 1) * qs])          # own query heads
        assert torch.equal(y[:, qs:qs + HD], k_full) and torch.equal(y[:, qs + HD:], v_full)   # replicated kv head

    gate_full, up_full = dense(w["gate"], x), dense(w["up"], x)
    half = INTER // world
    for rank in range(world):
        h = r[rank]["gate_up"]
        assert torch.equal(h[:, :half], gate_full[:, rank * half:(rank + 1) * half])
        assert torch.equal(h[:, half:], up_full[:, rank * half:(rank + 1) * half])

    xs = torch.randn(5, INTER, generator=torch.Generator().manual_seed(3)).to(DT)
    down_full = dense(w["down"], xs, w["down_bias"])
    for rank in range(world):
        assert torch.allclose(r[rank]["down"].float(), down_full.float(), **tol)     # bias added exactly once
        assert torch.allclose(r[rank]["down_split"].float(), down_full.float(), **tol)
        assert torch.equal(r[rank]["col_gathered"], dense(w["col"], x))
        assert torch.equal(r[rank]["allreduce"], torch.arange(8, dtype=torch.float32) * 3)
        assert torch.equal(r[rank]["allreduce_in_place"], torch.arange(8, dtype=torch.float32) * 3)
    assert torch.equal(r[0]["down"], r[1]["down"])

    table = reconstruct_partition(w["vocab"]["codes"], w["vocab"]["codebooks"], w["vocab"]["scales"])
    logits_full = dense(w["vocab"], x)
    for rank in range(world):
        assert torch.equal(r[rank]["emb"], table[w["ids"]])                  # exact: one rank's row plus zeros
        lo, hi = 96 * rank, min(96 * (rank + 1), VOCAB)
        assert torch.equal(r[rank]["logits_shard"][:, :hi - lo], logits_full[:, lo:hi])
        assert not r[rank]["logits_shard"][:, hi - lo:].any()                # padding rows: zero scales
        assert r[rank]["logits"].shape == (5, VOCAB)
        assert torch.equal(r[rank]["logits"][:, :96], logits_full[:, :96])
