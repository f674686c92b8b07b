#!/usr/bin/env python3
"""Golden vectors for the tensor-parallel weight loaders and the vocabulary shard logic, from the reference's own
Python.

Runs ONLY in the build container (needs /root/reference).  Loads, *by file path*,
  vllm/model_executor/layers/quantization/base_config.py, vllm/model_executor/utils.py,
  vllm/model_executor/layers/linear.py, .../quantization/aqlm.py, .../vocab_parallel_embedding.py
after installing empty package shells and stubs for what they import: ``vllm.distributed`` (rank and world size
settable from here; the collectives are never called), ``vllm.logger``, ``vllm.model_executor.parameter`` (empty
classes, only used in isinstance checks), ``vllm.platforms`` (reports no accelerator; "eager" compile backend) and
``vllm._custom_ops`` (empty).  Nothing of the reference is copied: the output is data only -- the seeded
checkpoint tensors and the per-rank parameter tensors the reference's loaders produced -- written to
``tests/golden/loaders_golden.npz`` / ``.json``.

Reference code exercised (vllm/model_executor/layers/):
  linear.py  ColumnParallelLinear.weight_loader :370-408, MergedColumnParallelLinear.weight_loader :494-640
             (is_metadata :619-624), QKVParallelLinear.weight_loader :873-1058 (is_metadata :1038-1044),
             RowParallelLinear.weight_loader :1144-1180, with the parameters AQLMLinearMethod.create_weights
             registers (quantization/aqlm.py:253-316)
  vocab_parallel_embedding.py  pad_vocab_size :50-53, VocabParallelEmbedding._get_indices :272-298,
             get_sharded_to_full_mapping :300-346, weight_loader :348-401, get_masked_input_and_mask :139-156

For TP in {1, 2, 4, 8} and every rank.  fp16 tensors are stored as uint16 bit patterns.
"""
import importlib.util
import json
import logging
import os
import sys
import types

import numpy as np
import torch
from torch.nn.parameter import Parameter

REF_ROOT = os.environ.get("PQ_REFERENCE_ROOT", "/root/reference")
MDIR = os.path.join(REF_ROOT, "vllm/model_executor")
HERE = os.path.dirname(os.path.abspath(__file__))

TP_STATE = {"rank": 0, "size": 1}

# the layer shapes of the fixtures (shared with tests/test_loaders_golden.py through the manifest)
HID, HEAD, HEADS, KV_HEADS, INTER, COL_OUT, D = 64, 16, 8, 2, 128, 64, 8
VOCABS = [(150, None), (1000, None), (1010, 1000)]      # (num_embeddings, org_num_embeddings)
TPS = (1, 2, 4, 8)
DT = torch.float16


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def load_reference():
    for n in ("vllm", "vllm.model_executor", "vllm.model_executor.layers", "vllm.model_executor.layers.quantization"):
        m = types.ModuleType(n)
        m.__path__ = []
        sys.modules[n] = m

    dist = types.ModuleType("vllm.distributed")

    def divide(a, b):
        assert a % b == 0, f"{a} is not divisible by {b}"
        return a // b

    def never(*a, **k):
        raise RuntimeError("collectives are not part of the loader fixtures")

    dist.divide = divide
    dist.get_tensor_model_parallel_rank = lambda: TP_STATE["rank"]
    dist.get_tensor_model_parallel_world_size = lambda: TP_STATE["size"]
    dist.split_tensor_along_last_dim = never
    dist.tensor_model_parallel_all_gather = never
    dist.tensor_model_parallel_all_reduce = never
    sys.modules[dist.__name__] = dist

    lg = types.ModuleType("vllm.logger")
    lg.init_logger = logging.getLogger
    sys.modules[lg.__name__] = lg

    pm = types.ModuleType("vllm.model_executor.parameter")
    for cls in ("BasevLLMParameter", "BlockQuantScaleParameter", "PackedColumnParameter", "PackedvLLMParameter",
                "PerTensorScaleParameter", "RowvLLMParameter"):
        setattr(pm, cls, type(cls, (Parameter,), {}))
    sys.modules[pm.__name__] = pm

    pf = types.ModuleType("vllm.platforms")

    class _Platform:
        simple_compile_backend = "eager"

        def __getattr__(self, name):          # is_tpu / is_hpu / is_cuda_alike / is_cpu ...: no accelerator
            if name.startswith("is_"):
                return lambda: False
            raise AttributeError(name)

    pf.current_platform = _Platform()
    sys.modules[pf.__name__] = pf

    ops = types.ModuleType("vllm._custom_ops")
    sys.modules[ops.__name__] = ops
    sys.modules["vllm"]._custom_ops = ops

    load_by_path("vllm.model_executor.layers.quantization.base_config", os.path.join(MDIR, "layers/quantization/base_config.py"))
    load_by_path("vllm.model_executor.utils", os.path.join(MDIR, "utils.py"))
    lin = load_by_path("vllm.model_executor.layers.linear", os.path.join(MDIR, "layers/linear.py"))
    aq = load_by_path("vllm.model_executor.layers.quantization.aqlm", os.path.join(MDIR, "layers/quantization/aqlm.py"))
    voc = load_by_path("vllm.model_executor.layers.vocab_parallel_embedding",
                       os.path.join(MDIR, "layers/vocab_parallel_embedding.py"))
    return lin, aq, voc


def bits(t):
    return t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def store(arrays, key, t):
    t = t.detach()
    arrays[key] = bits(t) if t.dtype in (torch.float16, torch.bfloat16) else t.contiguous().numpy().copy()


def checkpoint():
    """The unsharded 'checkpoint' tensors, one set per logical matrix: the recipe the test repeats."""
    g = torch.Generator().manual_seed(2024)

    def mat(n, kin):
        return {"codes": torch.randint(-128, 128, (n, kin // D, 1), generator=g, dtype=torch.int8),
                "codebooks": torch.randn(1, 256, 1, D, generator=g).to(DT),
                "scales": (torch.rand(n, 1, 1, 1, generator=g) + 0.5).to(DT),
                "bias": torch.randn(n, generator=g).to(DT)}

    ck = {"q": mat(HEADS * HEAD, HID), "k": mat(KV_HEADS * HEAD, HID), "v": mat(KV_HEADS * HEAD, HID),
          "gate": mat(INTER, HID), "up": mat(INTER, HID), "down": mat(HID, INTER), "col": mat(COL_OUT, HID)}
    for (vocab, org) in VOCABS:
        ck[f"vocab{vocab}"] = {"weight": torch.randn(vocab, HID, generator=g).to(DT),
                               "codes": torch.randint(-128, 128, (vocab, HID // D, 1), generator=g, dtype=torch.int8),
                               "ids": torch.randint(0, vocab, (4, 9), generator=g)}
    return ck


def main():
    lin, aq, voc = load_reference()
    cfg = aq.AQLMConfig(in_group_size=D, nbits_per_codebook=8, num_codebooks=1, out_group_size=1)

    class HeadConfig(aq.AQLMConfig):
        """AQLMConfig hands its method to linear layers only (aqlm.py:213-217); for the LM head fixture the same
        method -- hence the same parameter attributes -- is attached to the vocabulary-parallel layer."""

        def get_quant_method(self, layer, prefix):
            return aq.AQLMLinearMethod(self)

    head_cfg = HeadConfig(in_group_size=D, nbits_per_codebook=8, num_codebooks=1, out_group_size=1)
    ck = checkpoint()
    arrays, manifest = {}, {"shapes": dict(HID=HID, HEAD=HEAD, HEADS=HEADS, KV_HEADS=KV_HEADS, INTER=INTER,
                                           COL_OUT=COL_OUT, D=D), "vocabs": VOCABS, "tps": list(TPS),
                            "seed": 2024, "layers": [], "vocab": [], "pad_vocab_size": [], "masked_input": []}

    for name, tensors in ck.items():         # the checkpoint itself travels with the fixtures
        for pname, t in tensors.items():
            store(arrays, f"ck.{name}.{pname}", t)

    def load3(layer, src, *shard):
        for pname in ("codes", "codebooks", "scales"):
            p = getattr(layer, pname)
            p.weight_loader(p, src[pname], *shard)
        if layer.bias is not None:
            layer.bias.weight_loader(layer.bias, src["bias"], *shard)

    def dump(tag, layer):
        rec = {"tag": tag, "params": {}}
        for pname in ("codes", "codebooks", "scales", "bias"):
            p = getattr(layer, pname, None)
            if p is None:
                continue
            store(arrays, f"{tag}.{pname}", p.data)
            rec["params"][pname] = list(p.shape)
        rec["output_partition_sizes"] = list(layer.codebooks.output_partition_sizes)
        manifest["layers"].append(rec)

    for tp in TPS:
        for rank in range(tp):
            TP_STATE.update(rank=rank, size=tp)
            tag = f"tp{tp}.r{rank}"
            # q / k / v from separate checkpoint tensors; the two kv heads are replicated from TP = 4 on
            qkv = lin.QKVParallelLinear(HID, HEAD, HEADS, KV_HEADS, bias=True, params_dtype=DT, quant_config=cfg)
            qkv.codes.data.zero_(); qkv.codebooks.data.zero_(); qkv.scales.data.zero_(); qkv.bias.data.zero_()
            for sid in "qkv":
                load3(qkv, ck[sid], sid)
            dump(f"{tag}.qkv", qkv)
            manifest["layers"][-1].update(num_heads=qkv.num_heads, num_kv_heads=qkv.num_kv_heads,
                                          num_kv_head_replicas=qkv.num_kv_head_replicas)

            gate_up = lin.MergedColumnParallelLinear(HID, [INTER, INTER], bias=True, params_dtype=DT, quant_config=cfg)
            for p in (gate_up.codes, gate_up.codebooks, gate_up.scales, gate_up.bias):
                p.data.zero_()
            load3(gate_up, ck["gate"], 0)
            load3(gate_up, ck["up"], 1)
            dump(f"{tag}.gate_up", gate_up)

            # the same layer from a checkpoint that stores gate and up already fused (loaded_shard_id None,
            # linear.py:503-583): codes and scales only -- the stacked codebooks have no output_dim and the
            # reference copies such a tensor whole
            fused = lin.MergedColumnParallelLinear(HID, [INTER, INTER], bias=False, params_dtype=DT, quant_config=cfg)
            for p in (fused.codes, fused.codebooks, fused.scales):
                p.data.zero_()
            for pname in ("codes", "scales"):
                p = getattr(fused, pname)
                p.weight_loader(p, torch.cat([ck["gate"][pname], ck["up"][pname]], dim=0))
            fused.codebooks.weight_loader(fused.codebooks, torch.cat([ck["gate"]["codebooks"], ck["up"]["codebooks"]], dim=0))
            dump(f"{tag}.gate_up_fused", fused)

            down = lin.RowParallelLinear(INTER, HID, bias=True, params_dtype=DT, quant_config=cfg)
            for p in (down.codes, down.codebooks, down.scales, down.bias):
                p.data.zero_()
            load3(down, ck["down"])
            dump(f"{tag}.down", down)

            col = lin.ColumnParallelLinear(HID, COL_OUT, bias=True, params_dtype=DT, quant_config=cfg)
            for p in (col.codes, col.codebooks, col.scales, col.bias):
                p.data.zero_()
            load3(col, ck["col"])
            dump(f"{tag}.col", col)

            # ---- vocabulary-parallel layers
            for (vocab, org) in VOCABS:
                emb = voc.VocabParallelEmbedding(vocab, HID, params_dtype=DT, org_num_embeddings=org)
                emb.weight.data.fill_(7.0)
                src = ck[f"vocab{vocab}"]
                emb.weight_loader(emb.weight, src["weight"][:emb.org_vocab_size])
                si = emb.shard_indices
                vtag = f"{tag}.vocab{vocab}"
                store(arrays, f"{vtag}.weight", emb.weight.data)
                mapping = emb.get_sharded_to_full_mapping()
                rec = {"tag": vtag, "tp": tp, "rank": rank, "vocab": vocab, "org": org,
                       "num_embeddings_padded": emb.num_embeddings_padded,
                       "org_vocab_size_padded": emb.org_vocab_size_padded,
                       "num_embeddings_per_partition": emb.num_embeddings_per_partition,
                       "indices": {k: getattr(si, k) for k in (
                           "padded_org_vocab_start_index", "padded_org_vocab_end_index",
                           "padded_added_vocab_start_index", "padded_added_vocab_end_index",
                           "org_vocab_start_index", "org_vocab_end_index", "added_vocab_start_index",
                           "added_vocab_end_index")},
                       "mapping": mapping}
                # the LM head with the PQ-attributed codes parameter: rows sharded by the same loader
                head = voc.ParallelLMHead(vocab, HID, params_dtype=DT, org_num_embeddings=org, quant_config=head_cfg)
                head.codes.data.fill_(5)
                head.codes.weight_loader(head.codes, src["codes"][:head.org_vocab_size])
                store(arrays, f"{vtag}.head_codes", head.codes.data)
                # scales are packed along the dimension they are sharded on (packed_dim == output_dim == 0): the
                # reference's loader reads param.packed_factor there (vocab_parallel_embedding.py:376-380), which
                # a plain Parameter with a pack_factor attribute does not have
                try:
                    head.scales.weight_loader(head.scales, torch.ones(head.org_vocab_size, 1, 1, 1, dtype=DT))
                    rec["head_scales_loader"] = "ok"
                except AttributeError as e:
                    rec["head_scales_loader"] = f"AttributeError: {e}"
                # masked ids of this rank (vocab_parallel_embedding.py:139-156), eager backend
                masked, mask = voc.get_masked_input_and_mask(
                    src["ids"], si.org_vocab_start_index, si.org_vocab_end_index, si.num_org_vocab_padding,
                    si.added_vocab_start_index, si.added_vocab_end_index)
                store(arrays, f"{vtag}.masked_ids", masked)
                store(arrays, f"{vtag}.mask", mask.to(torch.uint8))
                manifest["vocab"].append(rec)

    for v in (1, 63, 64, 65, 150, 32000, 128256):
        for pad in (64, 128):
            manifest["pad_vocab_size"].append({"vocab": v, "pad_to": pad, "out": voc.pad_vocab_size(v, pad)})

    np.savez_compressed(os.path.join(HERE, "loaders_golden.npz"), **arrays)
    with open(os.path.join(HERE, "loaders_golden.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print(f"{len(manifest['layers'])} layer records, {len(manifest['vocab'])} vocabulary records, {len(arrays)} arrays")


if __name__ == "__main__":
    main()
