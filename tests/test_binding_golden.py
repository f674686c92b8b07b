"""The drop-in case, pinned: tests/golden/make_golden_binding.py ran this package's PQConfig / PQLinearMethod against the
reference's REAL classes -- its registry (quantization/__init__.py:40-73), QuantizationConfig / LinearMethodBase, the four
parallel linear layers and their v1 weight loaders (linear.py:178-183, :349-358, :370-408, :494-640, :873-1058,
:1144-1180), as the reference's plugin test does (tests/quantization/test_register_quantization_config.py:49-118) --
and wrote what came out.  Here the stand-alone mirror (vllm_pq_amd.host: what runs when vLLM is absent) must give the
same: registry behaviour, parameter shapes / dtypes / loader attributes, and the per-rank tensors bit for bit.  CPU only.

When /root/reference is present (the build container) the generator itself is re-run in a fresh process, which
executes the package's vLLM branch (compat.py, pq.register()) against the reference, and must reproduce the committed
fixtures."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from vllm_pq_amd import PQConfig, compat, pq
from vllm_pq_amd.host import parallel_state as ps
from vllm_pq_amd.host.linear import (ColumnParallelLinear, MergedColumnParallelLinear, QKVParallelLinear,
                                     RowParallelLinear)

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
with open(os.path.join(GOLD, "binding_golden.json")) as f:
    MAN = json.load(f)
ARR = np.load(os.path.join(GOLD, "binding_golden.npz"), allow_pickle=False)
with open(os.path.join(GOLD, "loaders_golden.json")) as f:
    LOADERS_MAN = json.load(f)
CK = np.load(os.path.join(GOLD, "loaders_golden.npz"), allow_pickle=False)      # the seeded checkpoint travels there
S = MAN["shapes"]
DT = torch.float16


def tensor(arr, key):
    a = arr[key]
    if a.dtype == np.uint16:
        return torch.from_numpy(a.view(np.int16).copy()).view(DT)
    return torch.from_numpy(a.copy())


def ck(name):
    return {p: tensor(CK, f"ck.{name}.{p}") for p in ("codes", "codebooks", "scales", "bias")}


def test_registry_behaves_as_the_reference_registry_did():
    r = MAN["registry"]
    assert r["have_vllm"] is True                                  # the fixture came from the vLLM branch
    assert r["methods_after"] == r["methods_before"] + ["pq"]
    # the mirror (this process has no vLLM): same name, same gates, same errors
    assert compat.HAVE_VLLM is False
    assert "pq" in compat.QUANTIZATION_METHODS and compat.get_quantization_config("pq") is PQConfig
    n = compat.QUANTIZATION_METHODS.count("pq")
    pq.register()
    assert compat.QUANTIZATION_METHODS.count("pq") == n == 1
    with pytest.raises(ValueError) as e:
        compat.register_quantization_config("pq")(PQConfig)
    assert str(e.value) == r["errors"]["duplicate"]
    with pytest.raises(ValueError) as e:
        compat.register_quantization_config("pq_not_a_config")(object)
    assert str(e.value) == r["errors"]["not_a_config"]
    assert PQConfig.get_name() == r["name"] and PQConfig.get_min_capability() == r["min_capability"]
    assert [str(d) for d in PQConfig.get_supported_act_dtypes()] == r["act_dtypes"]
    assert PQConfig.get_config_filenames() == r["config_filenames"]
    assert "pq" not in r["rocm_allow_list_before"]                 # why register() extends the ROCm allow-list


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
    assert type(layer).__name__ == rec["class"] and type(layer.quant_method).__name__ == rec["quant_method"]
    for pname, want in rec["params"].items():
        p = getattr(layer, pname)
        assert list(p.shape) == want["shape"] and str(p.dtype) == want["dtype"], (tag, pname)
        for a, v in want["attrs"].items():
            assert hasattr(p, a), (tag, pname, a)
            got = getattr(p, a)
            assert (list(got) if isinstance(got, (list, tuple)) else got) == v, (tag, pname, a)
        for a in ("input_dim", "output_dim", "packed_dim", "pack_factor", "is_metadata", "output_partition_sizes"):
            assert hasattr(p, a) == (a in want["attrs"]), (tag, pname, a)      # no attribute the reference run lacked
        assert want["loader"] == "weight_loader"                   # the attribute-driven v1 loaders (linear.py:31-38)
        ref = tensor(ARR, f"{tag}.{pname}")
        got = p.data
        assert torch.equal(got.view(torch.int16) if got.dtype == DT else got,
                           ref.view(torch.int16) if ref.dtype == DT else ref), (tag, pname)


@pytest.mark.parametrize("size", MAN["tps"])
def test_mirror_layers_match_the_reference_layers_with_pq(tp, size):
    cfg = PQConfig(S["D"], 8, 1, 1)
    for rank in range(size):
        tp(rank, size)
        tag = f"tp{size}.r{rank}"
        qkv = QKVParallelLinear(S["HID"], S["HEAD"], S["HEADS"], S["KV_HEADS"], bias=True, params_dtype=DT, quant_config=cfg)
        zero(qkv)
        for sid in "qkv":
            load3(qkv, ck(sid), sid)
        same(f"{tag}.qkv", qkv)
        gate_up = MergedColumnParallelLinear(S["HID"], [S["INTER"], S["INTER"]], bias=True, params_dtype=DT, quant_config=cfg)
        zero(gate_up)
        load3(gate_up, ck("gate"), 0)
        load3(gate_up, ck("up"), 1)
        same(f"{tag}.gate_up", gate_up)
        down = RowParallelLinear(S["INTER"], S["HID"], bias=True, params_dtype=DT, quant_config=cfg)
        zero(down)
        load3(down, ck("down"))
        same(f"{tag}.down", down)
        col = ColumnParallelLinear(S["HID"], S["COL_OUT"], bias=True, params_dtype=DT, quant_config=cfg)
        zero(col)
        load3(col, ck("col"))
        same(f"{tag}.col", col)


def test_misaligned_input_raises_the_same_error():
    with pytest.raises(ValueError) as e:
        RowParallelLinear(S["INTER"] + 4, S["HID"], bias=False, params_dtype=DT, quant_config=PQConfig(S["D"], 8, 1, 1))
    assert str(e.value) == MAN["registry"]["errors"]["misaligned_input"]


@pytest.mark.skipif(not os.path.isdir(os.environ.get("PQ_REFERENCE_ROOT", "/root/reference")),
                    reason="the reference tree exists in the build container only")
def test_generator_reproduces_the_committed_fixtures(tmp_path):
    """Runs the package's vLLM branch against the reference's classes (a fresh process: the choice is made at import)."""
    out = tmp_path / "golden"
    out.mkdir()
    for name in ("make_golden_binding.py", "make_golden_loaders.py"):
        (out / name).write_text(open(os.path.join(GOLD, name)).read())
    env = dict(os.environ, PYTHONPATH=os.path.dirname(HERE))
    # the copies write beside themselves; they find the package through PYTHONPATH
    src = (out / "make_golden_binding.py").read_text().replace(
        "ROOT = os.path.dirname(os.path.dirname(HERE))", f"ROOT = {os.path.dirname(HERE)!r}")
    (out / "make_golden_binding.py").write_text(src)
    p = subprocess.run([sys.executable, str(out / "make_golden_binding.py")], env=env, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    with open(out / "binding_golden.json") as f:
        assert json.load(f) == MAN
    new = np.load(out / "binding_golden.npz", allow_pickle=False)
    assert sorted(new.files) == sorted(ARR.files)
    for k in ARR.files:
        assert np.array_equal(new[k], ARR[k]), k
