#!/usr/bin/env python3
"""Golden vectors for the element-wise layers either side of the PQ linears, from the reference's own Python.

Runs ONLY in the build container (needs /root/reference).  Loads
``vllm/model_executor/layers/activation.py`` and ``layernorm.py`` *by file path* after installing empty package
shells and stubs for what they import (``CustomOp``: an nn.Module with a no-op ``register`` decorator;
``current_platform``: reports neither CUDA nor CPU so no native op is looked up; ``vllm.distributed``,
``set_weight_attrs``, ``LazyDict``: placeholders the two classes used here never call), then calls
``SiluAndMul.forward_native`` (activation.py:73-76) and ``RMSNorm.forward_native`` (layernorm.py:38-74) on
seeded inputs.  Nothing of the reference is copied: the output is data only, written to
``tests/golden/layers_golden.npz`` / ``.json``.  fp16/bf16 tensors are stored as uint16 bit patterns.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

REF_ROOT = os.environ.get("PQ_REFERENCE_ROOT", "/root/reference")
LDIR = os.path.join(REF_ROOT, "vllm/model_executor/layers")
HERE = os.path.dirname(os.path.abspath(__file__))
DT = {"fp16": torch.float16, "bf16": torch.bfloat16}


def load_reference_layers():
    def shell(name):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
        return m

    for n in ("vllm", "vllm.model_executor", "vllm.model_executor.layers"):
        shell(n)
    dist = types.ModuleType("vllm.distributed")
    dist.divide = lambda a, b: a // b
    dist.get_tensor_model_parallel_rank = lambda: 0
    dist.get_tensor_model_parallel_world_size = lambda: 1
    sys.modules[dist.__name__] = dist

    co = types.ModuleType("vllm.model_executor.custom_op")

    class CustomOp(torch.nn.Module):
        @classmethod
        def register(cls, name):
            return lambda c: c

    co.CustomOp = CustomOp
    sys.modules[co.__name__] = co

    ut = types.ModuleType("vllm.model_executor.utils")
    ut.set_weight_attrs = lambda w, attrs: None
    sys.modules[ut.__name__] = ut

    pf = types.ModuleType("vllm.platforms")

    class _Platform:
        def is_cuda_alike(self): return False
        def is_cpu(self): return False
        def is_xpu(self): return False

    pf.current_platform = _Platform()
    sys.modules[pf.__name__] = pf

    vu = types.ModuleType("vllm.utils")
    vu.LazyDict = dict
    sys.modules[vu.__name__] = vu

    mods = {}
    for name in ("activation", "layernorm"):
        spec = importlib.util.spec_from_file_location(f"vllm.model_executor.layers.{name}",
                                                      os.path.join(LDIR, f"{name}.py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods["activation"], mods["layernorm"]


def bits(t: torch.Tensor) -> np.ndarray:
    return t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def main():
    act, ln = load_reference_layers()
    arrays, cases = {}, []
    cid = 0
    # ---- SiluAndMul: seeded normal inputs of a few widths, plus edge values ------------------------
    for dtype in ("bf16", "fp16"):
        for (T, I, scale) in ((5, 64, 1.0), (3, 200, 4.0), (2, 37, 0.05), (16, 1024, 2.0)):
            g = torch.Generator().manual_seed(4000 + cid)
            x = (torch.randn(T, 2 * I, generator=g) * scale).to(DT[dtype])
            if cid % 4 == 0:     # zeros, large magnitudes of both signs, tiny values
                x[0, :8] = torch.tensor([0.0, -0.0, 30.0, -30.0, 88.0, -88.0, 1e-3, -1e-3]).to(DT[dtype])
            y = act.SiluAndMul().forward_native(x)
            c = {"id": f"s{cid}", "kind": "silu_and_mul", "dtype": dtype, "T": T, "I": I}
            arrays[f"{c['id']}_x"], arrays[f"{c['id']}_y"] = bits(x), bits(y)
            cases.append(c)
            cid += 1
    # ---- RMSNorm with and without the residual add ---------------------------------------------
    for dtype in ("bf16", "fp16"):
        for (T, H, eps, scale) in ((4, 64, 1e-6, 1.0), (3, 200, 1e-5, 3.0), (2, 37, 1e-6, 0.1), (7, 4096, 1e-5, 1.0)):
            for with_res in (False, True):
                g = torch.Generator().manual_seed(5000 + cid)
                x = (torch.randn(T, H, generator=g) * scale).to(DT[dtype])
                w = (1.0 + 0.2 * torch.randn(H, generator=g)).to(DT[dtype])
                res = (torch.randn(T, H, generator=g) * scale).to(DT[dtype]) if with_res else None
                layer = ln.RMSNorm(H, eps=eps)
                layer.weight.data = w
                out = layer.forward_native(x.clone(), None if res is None else res.clone())
                c = {"id": f"n{cid}", "kind": "rms_norm", "dtype": dtype, "T": T, "H": H, "eps": eps,
                     "residual": with_res}
                arrays[f"{c['id']}_x"], arrays[f"{c['id']}_w"] = bits(x), bits(w)
                if with_res:
                    arrays[f"{c['id']}_res"] = bits(res)
                    arrays[f"{c['id']}_y"], arrays[f"{c['id']}_res_out"] = bits(out[0]), bits(out[1])
                else:
                    arrays[f"{c['id']}_y"] = bits(out)
                cases.append(c)
                cid += 1
    np.savez_compressed(os.path.join(HERE, "layers_golden.npz"), **arrays)
    with open(os.path.join(HERE, "layers_golden.json"), "w") as f:
        json.dump({"cases": cases, "torch": torch.__version__}, f, indent=1)
    print(f"{len(cases)} cases, {sum(a.nbytes for a in arrays.values())} bytes")


if __name__ == "__main__":
    main()
