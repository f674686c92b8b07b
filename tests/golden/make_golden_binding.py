#!/usr/bin/env python3
"""Golden vectors for the DROP-IN case: this package's PQConfig / PQLinearMethod registered with, and driven by, the
reference's own classes.

Runs ONLY in the build container (needs /root/reference).  It loads, *by file path* and under their ``vllm.*`` names,
the reference's registry (vllm/model_executor/layers/quantization/__init__.py), base_config.py, linear.py,
vocab_parallel_embedding.py and model_executor/utils.py (the stubs for what they import are those of
make_golden_loaders.py), and only then imports ``vllm_pq_amd`` -- whose compat layer therefore takes the vLLM branch
(HAVE_VLLM is True): PQConfig derives from the reference's QuantizationConfig, PQLinearMethod from its
LinearMethodBase, and registration goes through its register_quantization_config.  Then, as the reference's own
plugin test does (tests/quantization/test_register_quantization_config.py:49-118):

  * "pq" is in the reference's QUANTIZATION_METHODS and maps to PQConfig; register() again changes nothing; registering
    the name directly once more, or a class that is no QuantizationConfig, raises the reference's ValueError
    (quantization/__init__.py:62-71);
  * the reference's QKVParallelLinear / MergedColumnParallelLinear / RowParallelLinear / ColumnParallelLinear, built
    with quant_config=PQConfig(...), get a PQLinearMethod from LinearBase.__init__ (linear.py:178-183), whose
    create_weights registers codes / codebooks / scales with the attributes the reference's v1 loaders read;
  * a seeded checkpoint is loaded through the reference's weight_loader methods for TP in {1, 2, 4} and every rank.

The output is data only: parameter shapes, dtypes, loader attributes and the per-rank tensors, written to
``tests/golden/binding_golden.npz`` / ``.json``.  tests/test_binding_golden.py asserts that the stand-alone mirror
(vllm_pq_amd.host), which is what runs when vLLM is absent, gives the same.
"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import make_golden_loaders as mgl  # noqa: E402  (stubs, the shapes and the seeded checkpoint of the loader fixtures)

TPS = (1, 2, 4)
LOADER_ATTRS = ("input_dim", "output_dim", "packed_dim", "pack_factor", "is_metadata", "output_partition_sizes")


def load_registry():
    """The reference's registry module as the package vllm.model_executor.layers.quantization (its sub-modules
    base_config and aqlm are in sys.modules already)."""
    name = "vllm.model_executor.layers.quantization"
    qdir = os.path.join(mgl.MDIR, "layers/quantization")
    keep = {k: v for k, v in sys.modules.items() if k.startswith(name + ".")}
    spec = importlib.util.spec_from_file_location(name, os.path.join(qdir, "__init__.py"), submodule_search_locations=[qdir])
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    sys.modules.update(keep)
    for k, v in keep.items():
        setattr(m, k.rsplit(".", 1)[1], v)
    return m


def rocm_allow_list():
    """The ROCm platform's quantization allow-list as the reference writes it (vllm/platforms/rocm.py:108-111), read
    out of the source as a literal: the platform module itself needs a ROCm build of vLLM to import."""
    import ast
    tree = ast.parse(open(os.path.join(mgl.REF_ROOT, "vllm/platforms/rocm.py")).read())
    for node in ast.walk(tree):
        if isinstance(node, ast.AnnAssign) and getattr(node.target, "id", "") == "supported_quantization":
            return list(ast.literal_eval(node.value))
    raise RuntimeError("supported_quantization not found in vllm/platforms/rocm.py")


def verify_quantization(platform_cls, quant):
    """Platform.verify_quantization as the reference defines it (vllm/platforms/interface.py:272-280): a non-empty
    allow-list rejects every name that is not on it."""
    if platform_cls.supported_quantization and quant not in platform_cls.supported_quantization:
        raise ValueError(f"{quant} quantization is currently not supported in {platform_cls.device_name}.")


def main():
    assert "vllm_pq_amd" not in sys.modules
    lin, _aq, voc = mgl.load_reference()
    reg = load_registry()
    before = list(reg.QUANTIZATION_METHODS)
    # a stand-in for vllm.platforms.rocm holding the reference's allow-list, which register() must extend
    import types
    rocm = types.ModuleType("vllm.platforms.rocm")
    allow = rocm_allow_list()

    class RocmPlatform:
        device_name = "rocm"
        supported_quantization = list(allow)

    rocm.RocmPlatform = RocmPlatform
    sys.modules[rocm.__name__] = rocm
    try:
        verify_quantization(RocmPlatform, "pq")
        raise AssertionError("the unextended ROCm allow-list must reject pq")
    except ValueError:
        pass

    import vllm_pq_amd
    from vllm_pq_amd import PQConfig, compat, pq

    manifest = {"tps": list(TPS), "shapes": dict(HID=mgl.HID, HEAD=mgl.HEAD, HEADS=mgl.HEADS, KV_HEADS=mgl.KV_HEADS,
                                                  INTER=mgl.INTER, COL_OUT=mgl.COL_OUT, D=mgl.D),
                "seed": 2024, "registry": {}, "layers": []}
    arrays = {}

    # ---- the compat layer took the vLLM branch: the reference's classes are the bases
    assert compat.HAVE_VLLM is True
    assert compat.LinearBase is lin.LinearBase and compat.LinearMethodBase is lin.LinearMethodBase
    assert compat.register_quantization_config is reg.register_quantization_config
    assert compat.QUANTIZATION_METHODS is reg.QUANTIZATION_METHODS
    assert issubclass(PQConfig, reg.QuantizationConfig)
    assert issubclass(pq.PQLinearMethod, lin.LinearMethodBase)
    assert compat.VocabParallelEmbedding is voc.VocabParallelEmbedding
    manifest["registry"]["have_vllm"] = True

    # ---- registration (the package registers at import, as a vllm.general_plugins entry point would)
    assert "pq" not in before and reg.QUANTIZATION_METHODS == before + ["pq"]
    assert reg._CUSTOMIZED_METHOD_TO_QUANT_CONFIG["pq"] is PQConfig
    vllm_pq_amd.register()
    vllm_pq_amd.register()
    assert reg.QUANTIZATION_METHODS.count("pq") == 1, "register() is not idempotent"
    # the ROCm platform gate (vllm/platforms/rocm.py:108-111, interface.py:272-280) now lets the name through, once
    assert RocmPlatform.supported_quantization == allow + ["pq"]
    verify_quantization(RocmPlatform, "pq")
    manifest["registry"]["rocm_allow_list_before"] = allow
    errors = {}
    try:
        reg.register_quantization_config("pq")(PQConfig)
        raise AssertionError("a second registration of the name must raise")
    except ValueError as e:
        errors["duplicate"] = str(e)
    try:
        reg.register_quantization_config("pq_not_a_config")(object)
        raise AssertionError("a class that is no QuantizationConfig must raise")
    except ValueError as e:
        errors["not_a_config"] = str(e)
    assert "pq_not_a_config" not in reg.QUANTIZATION_METHODS
    # get_quantization_config imports every built-in method's module first (quantization/__init__.py:79-106); where
    # those import here it must hand back PQConfig, otherwise the reason is recorded
    try:
        got = reg.get_quantization_config("pq")
        assert got is PQConfig
        manifest["registry"]["get_quantization_config"] = "PQConfig"
    except ImportError as e:
        manifest["registry"]["get_quantization_config"] = f"built-in methods do not import here ({type(e).__name__}: {e})"
    manifest["registry"].update(methods_before=before, methods_after=list(reg.QUANTIZATION_METHODS), errors=errors,
                                name=PQConfig.get_name(), min_capability=PQConfig.get_min_capability(),
                                act_dtypes=[str(d) for d in PQConfig.get_supported_act_dtypes()],
                                config_filenames=PQConfig.get_config_filenames())
    cfg = PQConfig.from_config({"in_group_size": mgl.D, "nbits_per_codebook": 8, "num_codebooks": 1, "out_group_size": 1})
    assert (cfg.in_group_size, cfg.nbits_per_codebook, cfg.num_codebooks, cfg.out_group_size) == (mgl.D, 8, 1, 1)
    # non-linear layers get no method (the vocabulary layers only when the config says so)
    assert cfg.get_quant_method(torch.nn.LayerNorm(8), "") is None
    assert cfg.get_quant_method(voc.VocabParallelEmbedding(64, mgl.HID, params_dtype=mgl.DT), "") is None

    ck = mgl.checkpoint()

    def load3(layer, src, *shard):
        for pname in ("codes", "codebooks", "scales"):
            p = getattr(layer, pname)
            p.weight_loader(p, src[pname], *shard)
        if layer.bias is not None:
            layer.bias.weight_loader(layer.bias, src["bias"], *shard)

    def dump(tag, layer):
        assert type(layer.quant_method) is pq.PQLinearMethod, type(layer.quant_method)
        rec = {"tag": tag, "class": type(layer).__name__, "quant_method": type(layer.quant_method).__name__, "params": {}}
        for pname in ("codes", "codebooks", "scales", "bias"):
            p = getattr(layer, pname, None)
            if p is None:
                continue
            assert p.requires_grad is False or pname == "bias"
            mgl.store(arrays, f"{tag}.{pname}", p.data)
            attrs = {a: getattr(p, a) for a in LOADER_ATTRS if hasattr(p, a)}
            # which loader the reference attached: the attribute-driven v1 one (PQLinearMethod is not in the
            # reference's WEIGHT_LOADER_V2_SUPPORTED list, linear.py:31-38)
            rec["params"][pname] = {"shape": list(p.shape), "dtype": str(p.dtype), "attrs": attrs,
                                    "loader": getattr(p.weight_loader, "__name__", "?")}
        manifest["layers"].append(rec)

    def zero(layer):
        for pname in ("codes", "codebooks", "scales", "bias"):
            p = getattr(layer, pname, None)
            if p is not None:
                p.data.zero_()

    for tp in TPS:
        for rank in range(tp):
            mgl.TP_STATE.update(rank=rank, size=tp)
            tag = f"tp{tp}.r{rank}"
            qkv = lin.QKVParallelLinear(mgl.HID, mgl.HEAD, mgl.HEADS, mgl.KV_HEADS, bias=True, params_dtype=mgl.DT,
                                        quant_config=cfg)
            zero(qkv)
            for sid in "qkv":
                load3(qkv, ck[sid], sid)
            dump(f"{tag}.qkv", qkv)
            gate_up = lin.MergedColumnParallelLinear(mgl.HID, [mgl.INTER, mgl.INTER], bias=True, params_dtype=mgl.DT,
                                                     quant_config=cfg)
            zero(gate_up)
            load3(gate_up, ck["gate"], 0)
            load3(gate_up, ck["up"], 1)
            dump(f"{tag}.gate_up", gate_up)
            down = lin.RowParallelLinear(mgl.INTER, mgl.HID, bias=True, params_dtype=mgl.DT, quant_config=cfg)
            zero(down)
            load3(down, ck["down"])
            dump(f"{tag}.down", down)
            col = lin.ColumnParallelLinear(mgl.HID, mgl.COL_OUT, bias=True, params_dtype=mgl.DT, quant_config=cfg)
            zero(col)
            load3(col, ck["col"])
            dump(f"{tag}.col", col)
            if tp == 1:
                # params_dtype the reference's AQLM method refuses and this method takes; a misaligned input size raises
                # this method's (= the reference's, aqlm.py:240-244) error through the reference's constructor
                bf = lin.ColumnParallelLinear(mgl.HID, mgl.COL_OUT, bias=False, params_dtype=torch.bfloat16, quant_config=cfg)
                assert bf.codebooks.dtype == torch.bfloat16
                try:
                    lin.RowParallelLinear(mgl.INTER + 4, mgl.HID, bias=False, params_dtype=mgl.DT, quant_config=cfg)
                    raise AssertionError("in_features % in_group_size != 0 must raise")
                except ValueError as e:
                    manifest["registry"]["errors"]["misaligned_input"] = str(e)

    np.savez_compressed(os.path.join(HERE, "binding_golden.npz"), **arrays)
    with open(os.path.join(HERE, "binding_golden.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print(f"wrote {len(arrays)} arrays, {len(manifest['layers'])} layers; get_quantization_config: "
          f"{manifest['registry']['get_quantization_config']}")


if __name__ == "__main__":
    main()
