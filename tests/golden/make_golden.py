#!/usr/bin/env python3
"""Generate golden vectors for the PQ linear path from the reference's own Python.

Runs ONLY in the build container (needs /root/reference).  It loads
``vllm/model_executor/layers/quantization/aqlm.py`` *by file path* (the ``vllm``
package itself is not importable: its ``__init__`` chain needs ``msgspec``), after
installing empty package shells and three tiny stub modules for the imports at
aqlm.py:13-17.  Nothing of the reference is copied: the output is data only
(inputs + the tensors the reference functions returned), written to
``tests/golden/pq_golden.npz`` and ``tests/golden/pq_golden.json``.

Reference functions exercised (file = vllm/model_executor/layers/quantization/aqlm.py):
  get_int_dtype            :20-29
  unpack_int_data          :32-34
  dequantize_weight        :37-74
  dequantize_gemm          :77-90
  generic_dequantize_gemm  :94-127
  AQLMConfig / AQLMLinearMethod.create_weights  :159-323

fp16/bf16 tensors are stored as uint16 bit patterns (numpy has no bf16).
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

REF_ROOT = os.environ.get("PQ_REFERENCE_ROOT", "/root/reference")
QDIR = os.path.join(REF_ROOT, "vllm/model_executor/layers/quantization")
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference_aqlm():
    def shell(name):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
        return m

    for n in ("vllm", "vllm.model_executor", "vllm.model_executor.layers",
              "vllm.model_executor.layers.quantization"):
        shell(n)
    spec = importlib.util.spec_from_file_location(
        "vllm.model_executor.layers.quantization.base_config",
        os.path.join(QDIR, "base_config.py"))
    bc = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = bc
    spec.loader.exec_module(bc)

    ops = types.ModuleType("vllm._custom_ops")
    sys.modules[ops.__name__] = ops
    sys.modules["vllm"]._custom_ops = ops

    lin = types.ModuleType("vllm.model_executor.layers.linear")

    class LinearBase(torch.nn.Module):
        pass

    class LinearMethodBase(bc.QuantizeMethodBase):
        pass

    lin.LinearBase = LinearBase
    lin.LinearMethodBase = LinearMethodBase
    sys.modules[lin.__name__] = lin

    ut = types.ModuleType("vllm.model_executor.utils")

    def set_weight_attrs(weight, attrs):
        if attrs is None:
            return
        for k, v in attrs.items():
            assert not hasattr(weight, k), f"overwriting attr {k}"
            setattr(weight, k, v)

    ut.set_weight_attrs = set_weight_attrs
    sys.modules[ut.__name__] = ut

    spec = importlib.util.spec_from_file_location(
        "vllm.model_executor.layers.quantization.aqlm",
        os.path.join(QDIR, "aqlm.py"))
    aq = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = aq
    spec.loader.exec_module(aq)
    return aq, lin


DT = {"fp16": torch.float16, "bf16": torch.bfloat16}


def bits(t: torch.Tensor) -> np.ndarray:
    """fp16/bf16 tensor -> uint16 bit pattern array."""
    return t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def rand_codes(gen, shape, nbits, int_dtype):
    k = 1 << nbits
    lo, hi = -(k // 2), k // 2          # signed storage range that maps onto 0..K-1
    if nbits == 8 or nbits == 16:
        pass
    else:
        # K=16 in int8 / K=4096 in int16: the reference benchmark draws the full signed
        # range of the storage dtype; the modulo makes every value a valid index.
        info = torch.iinfo(int_dtype)
        lo, hi = info.min, info.max + 1
    return torch.randint(lo, hi, shape, generator=gen, dtype=torch.int64).to(int_dtype)


def main():
    aq, lin = load_reference_aqlm()
    arrays = {}
    manifest = {"cases": [], "int_dtype": {}, "unpack": [], "create_weights": []}

    # ---- get_int_dtype table -------------------------------------------------
    for nb in (1, 4, 8, 9, 12, 16, 17, 32, 33, 64):
        manifest["int_dtype"][str(nb)] = str(aq.get_int_dtype(nb)).replace("torch.", "")

    # ---- unpack_int_data edge values -----------------------------------------
    for nb, dt in ((4, torch.int8), (8, torch.int8), (12, torch.int16), (16, torch.int16)):
        info = torch.iinfo(dt)
        vals = torch.tensor([info.min, info.min + 1, -1, 0, 1, (1 << nb) // 2 - 1,
                             info.max - 1, info.max], dtype=dt)
        out = aq.unpack_int_data(vals, nb)
        manifest["unpack"].append({"nbits": nb, "dtype": str(dt).replace("torch.", ""),
                                   "in": vals.tolist(), "out": out.tolist()})

    case_id = 0

    def add_case(kind, K, d, M, N, Kin, parts, T, with_bias, dtn, scales_mode, edge=False):
        nonlocal case_id
        dt = DT[dtn]
        nbits = K.bit_length() - 1
        idt = aq.get_int_dtype(nbits)
        gen = torch.Generator().manual_seed(1234 + case_id)
        G = Kin // d
        P = len(parts)
        codes = rand_codes(gen, (N, G, M), nbits, idt)
        if edge:
            info = torch.iinfo(idt)
            flat = codes.view(-1)
            edge_vals = [info.min, info.max, -1, 0, 1, -(K // 2), K // 2 - 1, K - 1 if K - 1 <= info.max else -1]
            for i, v in enumerate(edge_vals):
                flat[i] = v
                flat[-1 - i] = v
        codebooks = torch.randn(P * M, K, 1, d, generator=gen).to(dt)
        if scales_mode == "normal":
            scales = torch.randn(N, 1, 1, 1, generator=gen).to(dt)
        else:
            scales = (torch.rand(N, 1, 1, 1, generator=gen) + 0.5).to(dt)
        name = f"c{case_id:03d}"
        rec = dict(id=name, kind=kind, K=K, d=d, M=M, N=N, Kin=Kin, parts=parts, T=T,
                   bias=with_bias, dtype=dtn, scales=scales_mode, edge=edge,
                   code_dtype=str(idt).replace("torch.", ""))
        arrays[name + "_codes"] = codes.numpy()
        arrays[name + "_codebooks"] = bits(codebooks)
        if kind == "gather":
            # pure gather: dequantize_weight(unpack(codes), codebooks, None), one partition
            assert P == 1
            w = aq.dequantize_weight(aq.unpack_int_data(codes, nbits), codebooks, None)
            arrays[name + "_out"] = bits(w)
        elif kind == "dequant":
            # scaled weight, per partition as generic_dequantize_gemm slices it (aqlm.py:115-126)
            arrays[name + "_scales"] = bits(scales)
            ws, off = [], 0
            for p, sz in enumerate(parts):
                ws.append(aq.dequantize_weight(
                    aq.unpack_int_data(codes.narrow(0, off, sz), nbits),
                    codebooks.narrow(0, p * M, M), scales.narrow(0, off, sz)))
                off += sz
            arrays[name + "_out"] = bits(torch.cat(ws, 0))
        elif kind == "gemm":
            arrays[name + "_scales"] = bits(scales)
            x = torch.randn(T, Kin, generator=gen).to(dt)
            arrays[name + "_x"] = bits(x)
            bias = None
            if with_bias:
                bias = torch.randn(N, generator=gen).to(dt)
                arrays[name + "_bias"] = bits(bias)
            y = aq.generic_dequantize_gemm(x, codes, codebooks, scales, parts, bias)
            assert y.shape == (T, N) and y.dtype == dt
            arrays[name + "_out"] = bits(y)
        else:
            raise ValueError(kind)
        manifest["cases"].append(rec)
        case_id += 1

    # ---- pure gather, every sweep point, both dtypes (bit-exact contract) ------
    for K in (16, 256, 4096):
        for d in (2, 4, 8):
            for dtn in ("fp16", "bf16"):
                if K == 4096 and dtn == "fp16" and d != 8:
                    continue                      # keep the fixture file small
                add_case("gather", K, d, 1, 64, 128, [64], 0, False, dtn, "uniform")
    # edge codes (min/max signed storage values, -1 <-> K-1)
    for K, d in ((16, 4), (256, 8), (256, 4), (4096, 8)):
        add_case("gather", K, d, 1, 32, 64, [32], 0, False, "bf16", "uniform", edge=True)

    # ---- scaled dequant incl. additive codebooks (M=2) and partitions ----------
    for dtn in ("fp16", "bf16"):
        add_case("dequant", 256, 8, 1, 192, 128, [128, 32, 32], 0, False, dtn, "normal")
        add_case("dequant", 256, 8, 2, 64, 128, [64], 0, False, dtn, "normal")
        add_case("dequant", 256, 4, 1, 96, 64, [48, 48], 0, False, dtn, "uniform")
        add_case("dequant", 16, 2, 2, 64, 64, [32, 32], 0, False, dtn, "normal")

    # ---- generic_dequantize_gemm ----------------------------------------------
    for dtn in ("fp16", "bf16"):
        for (N, parts) in ((64, [64]), (192, [128, 32, 32])):
            for Kin in (128, 256):
                for T in (1, 5, 16):
                    for wb in (False, True):
                        if dtn == "fp16" and (Kin == 128 and T == 5):
                            continue
                        add_case("gemm", 256, 8, 1, N, Kin, parts, T, wb, dtn, "uniform")
    for dtn in ("fp16", "bf16"):
        add_case("gemm", 256, 8, 2, 64, 128, [64], 5, True, dtn, "uniform")      # AQLM 2x8
        add_case("gemm", 256, 8, 2, 192, 128, [128, 32, 32], 3, False, dtn, "normal")
        add_case("gemm", 256, 4, 1, 96, 192, [48, 48], 7, True, dtn, "uniform")   # OPT-like, bias live
        add_case("gemm", 256, 4, 1, 128, 128, [128], 33, True, dtn, "normal")
        add_case("gemm", 16, 2, 1, 64, 64, [64], 4, False, dtn, "uniform")
        add_case("gemm", 16, 8, 1, 64, 128, [32, 32], 2, True, dtn, "uniform")
        add_case("gemm", 256, 2, 1, 64, 64, [64], 9, False, dtn, "uniform")
    add_case("gemm", 4096, 8, 1, 64, 128, [64], 5, True, "bf16", "uniform")
    add_case("gemm", 4096, 4, 1, 64, 64, [64], 2, False, "bf16", "uniform")
    add_case("gemm", 256, 8, 1, 128, 256, [64, 32, 16, 16], 40, True, "bf16", "uniform")  # 4 partitions, T>32
    add_case("gemm", 256, 8, 1, 64, 128, [64], 70, False, "bf16", "normal", edge=True)

    # ---- create_weights: shapes, dtypes and loader attrs -----------------------
    for (d, nbits, M, kin, parts) in ((8, 8, 1, 4096, [4096, 1024, 1024]), (4, 8, 1, 768, [768, 768, 768]),
                                      (8, 16, 1, 4096, [4096]), (8, 8, 2, 4096, [11008, 11008]),
                                      (2, 4, 1, 64, [64])):
        cfg = aq.AQLMConfig(d, nbits, M, 1)
        method = cfg.get_quant_method(lin.LinearBase(), "x")
        layer = torch.nn.Module()
        method.create_weights(layer, kin, parts, kin, sum(parts), torch.half, weight_loader="WL")
        rec = {"in_group_size": d, "nbits": nbits, "num_codebooks": M, "kin": kin, "parts": parts, "params": {}}
        for pn in ("codes", "codebooks", "scales"):
            p = getattr(layer, pn)
            attrs = {k: getattr(p, k) for k in ("input_dim", "output_dim", "packed_dim", "pack_factor",
                                                  "is_metadata", "output_partition_sizes", "weight_loader")
                     if hasattr(p, k)}
            rec["params"][pn] = {"shape": list(p.shape), "dtype": str(p.dtype).replace("torch.", ""),
                                 "requires_grad": p.requires_grad, "attrs": attrs}
        manifest["create_weights"].append(rec)
    cfg = aq.AQLMConfig.from_config({"in_group_size": 8, "nbits_per_codebook": 8,
                                     "num_codebooks": 1, "out_group_size": 1})
    manifest["config"] = {"name": cfg.get_name(), "repr": repr(cfg), "pack_factor": cfg.pack_factor,
                          "filenames": cfg.get_config_filenames(), "min_capability": cfg.get_min_capability()}

    np.savez_compressed(os.path.join(HERE, "pq_golden.npz"), **arrays)
    with open(os.path.join(HERE, "pq_golden.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    sz = os.path.getsize(os.path.join(HERE, "pq_golden.npz"))
    print(f"{len(manifest['cases'])} cases, {sz/1e6:.2f} MB, torch {torch.__version__}")


if __name__ == "__main__":
    main()
