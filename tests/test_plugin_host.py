"""Host side of the drop-in boundary: registry, PQConfig, PQLinearMethod.create_weights and the
attribute-driven weight loaders, checked against what the reference's own classes produce
(tests/golden/pq_golden.json, written by tests/golden/make_golden.py) and against the reference's
plugin-surface test (tests/quantization/test_register_quantization_config.py:88-97).  CPU only."""
import pytest
import torch

from conftest import golden
from vllm_pq_amd import PQConfig, PQLinearMethod, get_int_dtype, unpack_int_data
from vllm_pq_amd.compat import (QUANTIZATION_METHODS, QuantizationConfig, get_quantization_config,
                                register_quantization_config)
from vllm_pq_amd.host.linear import (ColumnParallelLinear, LinearBase, MergedColumnParallelLinear,
                                     QKVParallelLinear, ReplicatedLinear, RowParallelLinear)

G = golden()


def test_pq_is_registered_and_duplicates_are_rejected():
    assert "pq" in QUANTIZATION_METHODS
    assert get_quantization_config("pq") is PQConfig
    with pytest.raises(ValueError):                      # already registered
        register_quantization_config("pq")(PQConfig)
    with pytest.raises(ValueError):                      # built-in name
        register_quantization_config("aqlm")(PQConfig)
    with pytest.raises(ValueError):                      # not a QuantizationConfig
        register_quantization_config("not_a_config")(object)
    with pytest.raises(ValueError):
        get_quantization_config("no_such_method")
    import vllm_pq_amd
    vllm_pq_amd.register()                               # re-entrant (runs in every worker process)
    assert QUANTIZATION_METHODS.count("pq") == 1


def test_register_custom_config_like_the_reference_test():
    @register_quantization_config("custom_quant_for_test")
    class Custom(QuantizationConfig):
        def get_name(self):
            return "custom_quant_for_test"

        def get_supported_act_dtypes(self):
            return [torch.half]

        @classmethod
        def get_min_capability(cls):
            return -1

        @staticmethod
        def get_config_filenames():
            return []

        @classmethod
        def from_config(cls, config):
            return cls()

        def get_quant_method(self, layer, prefix):
            return None

    assert get_quantization_config("custom_quant_for_test") is Custom


def test_config_matches_reference_semantics():
    ref = G.manifest["config"]
    cfg = PQConfig.from_config({"in_group_size": 8, "nbits_per_codebook": 8, "num_codebooks": 1, "out_group_size": 1})
    assert cfg.get_name() == "pq" and cfg.pack_factor == ref["pack_factor"]
    assert cfg.get_config_filenames() == ref["filenames"] == []
    assert repr(cfg) == ref["repr"].replace("AQLMConfig", "PQConfig")
    assert torch.bfloat16 in cfg.get_supported_act_dtypes() and torch.half in cfg.get_supported_act_dtypes()
    assert isinstance(cfg.get_min_capability(), int)
    # PQ-flavoured keys and defaults
    c2 = PQConfig.from_config({"sub_dim": 4, "codebook_size": 256})
    assert (c2.in_group_size, c2.nbits_per_codebook, c2.num_codebooks, c2.out_group_size) == (4, 8, 1, 1)
    with pytest.raises(ValueError):
        PQConfig.from_config({"nbits_per_codebook": 8})          # no sub-vector length
    with pytest.raises(AssertionError):
        PQConfig(8, 8, 1, 2)                                      # out_group_size must be 1 (aqlm.py:179)
    # only linear layers get the method (aqlm.py:213-217)
    assert cfg.get_quant_method(torch.nn.Embedding(4, 4), "embed") is None
    assert isinstance(cfg.get_quant_method(LinearBase(8, 8), "lin"), PQLinearMethod)
    assert cfg.packed_modules_mapping == {}


def test_int_dtype_and_unpack_match_reference():
    for nbits, want in G.manifest["int_dtype"].items():
        assert str(get_int_dtype(int(nbits))).replace("torch.", "") == want
    with pytest.raises(ValueError):
        get_int_dtype(65)
    for rec in G.manifest["unpack"]:
        vals = torch.tensor(rec["in"], dtype=getattr(torch, rec["dtype"]))
        assert unpack_int_data(vals, rec["nbits"]).tolist() == rec["out"]


@pytest.mark.parametrize("rec", G.manifest["create_weights"], ids=lambda r: f"d{r['in_group_size']}b{r['nbits']}M{r['num_codebooks']}")
def test_create_weights_matches_reference(rec):
    cfg = PQConfig(rec["in_group_size"], rec["nbits"], rec["num_codebooks"], 1)
    layer = torch.nn.Module()
    PQLinearMethod(cfg).create_weights(layer, rec["kin"], rec["parts"], rec["kin"], sum(rec["parts"]), torch.half,
                                       weight_loader="WL")
    for name, want in rec["params"].items():
        p = getattr(layer, name)
        assert list(p.shape) == want["shape"], name
        assert str(p.dtype).replace("torch.", "") == want["dtype"], name
        assert p.requires_grad is False
        for k, v in want["attrs"].items():
            assert getattr(p, k) == v, (name, k)
    # dtype and alignment errors (aqlm.py:238-251); bf16 is accepted here, unlike the reference
    PQLinearMethod(cfg).create_weights(torch.nn.Module(), rec["kin"], rec["parts"], rec["kin"], sum(rec["parts"]),
                                       torch.bfloat16)
    with pytest.raises(ValueError):
        PQLinearMethod(cfg).create_weights(torch.nn.Module(), rec["kin"], rec["parts"], rec["kin"], sum(rec["parts"]),
                                           torch.float32)
    with pytest.raises(ValueError):
        PQLinearMethod(cfg).create_weights(torch.nn.Module(), rec["kin"] + 1, rec["parts"], rec["kin"] + 1,
                                           sum(rec["parts"]), torch.half)


def _fill(t, seed):
    g = torch.Generator().manual_seed(seed)
    if t.dtype.is_floating_point:
        return torch.randn(t.shape, generator=g).to(t.dtype)
    return torch.randint(-128, 128, t.shape, generator=g).to(t.dtype)


def test_weight_loaders_tp1_fused_layers():
    """Per-matrix checkpoint tensors land in the right rows of the fused parameters (TP=1)."""
    cfg = PQConfig(8, 8, 1, 1)
    qkv = QKVParallelLinear(64, 16, 4, 2, bias=True, params_dtype=torch.bfloat16, quant_config=cfg)
    assert qkv.codes.shape == (64 + 32 + 32, 8, 1) and qkv.codebooks.shape == (3, 256, 1, 8)
    assert qkv.codebooks.output_partition_sizes == [64, 32, 32]
    want = {}
    for i, (sid, rows) in enumerate((("q", 64), ("k", 32), ("v", 32))):
        for pname, shape in (("codes", (rows, 8, 1)), ("codebooks", (1, 256, 1, 8)), ("scales", (rows, 1, 1, 1))):
            param = getattr(qkv, pname)
            w = _fill(torch.empty(shape, dtype=param.dtype), 10 * i + len(pname))
            param.weight_loader(param, w, sid)
            want[(pname, sid)] = w
    off = {"q": 0, "k": 64, "v": 96}
    for (pname, sid), w in want.items():
        param = getattr(qkv, pname)
        if pname == "codebooks":
            got = param.data["qkv".index(sid):"qkv".index(sid) + 1]
        else:
            got = param.data[off[sid]:off[sid] + w.shape[0]]
        assert torch.equal(got, w), (pname, sid)
    # a checkpoint that is already fused on disk: split by the loader itself
    fused = _fill(torch.empty_like(qkv.codes.data), 99)
    qkv.codes.weight_loader(qkv.codes, fused)
    assert torch.equal(qkv.codes.data, fused)

    gate_up = MergedColumnParallelLinear(64, [48, 80], bias=False, params_dtype=torch.half, quant_config=cfg)
    g0, g1 = _fill(torch.empty(48, 8, 1, dtype=torch.int8), 1), _fill(torch.empty(80, 8, 1, dtype=torch.int8), 2)
    gate_up.codes.weight_loader(gate_up.codes, g0, 0)
    gate_up.codes.weight_loader(gate_up.codes, g1, 1)
    assert torch.equal(gate_up.codes.data, torch.cat([g0, g1]))
    cb1 = _fill(torch.empty(1, 256, 1, 8, dtype=torch.half), 3)
    gate_up.codebooks.weight_loader(gate_up.codebooks, cb1, 1)
    assert torch.equal(gate_up.codebooks.data[1:2], cb1)
    with pytest.raises((AssertionError, RuntimeError)):
        gate_up.codes.weight_loader(gate_up.codes, g0, 1)          # 48 rows offered for the 80-row shard

    for cls in (ReplicatedLinear, ColumnParallelLinear, RowParallelLinear):
        lin = cls(64, 32, bias=True, params_dtype=torch.half, quant_config=cfg)
        assert isinstance(lin.quant_method, PQLinearMethod)
        w = _fill(torch.empty_like(lin.codes.data), 5)
        lin.codes.weight_loader(lin.codes, w)
        assert torch.equal(lin.codes.data, w)


def test_unquantized_layers_compute_on_cpu():
    """The layer classes themselves are device-agnostic; only the PQ method needs the GPU."""
    lin = RowParallelLinear(16, 8, bias=True, params_dtype=torch.float32)
    lin.weight.data.normal_()
    lin.bias.data.normal_()
    x = torch.randn(3, 16)
    y, b = lin(x)
    assert b is None and torch.allclose(y, x @ lin.weight.t() + lin.bias, atol=1e-6)
    skip = ColumnParallelLinear(16, 8, bias=True, skip_bias_add=True, params_dtype=torch.float32)
    skip.weight.data.normal_()
    y, b = skip(x)
    assert b is skip.bias and torch.allclose(y, x @ skip.weight.t(), atol=1e-6)


def test_apply_on_cpu_layer_fails_loudly():
    cfg = PQConfig(8, 8, 1, 1)
    lin = ReplicatedLinear(64, 32, bias=False, params_dtype=torch.bfloat16, quant_config=cfg)
    lin.codes.data.zero_()
    lin.codebooks.data.zero_()
    lin.scales.data.fill_(1)
    lin.quant_method.process_weights_after_loading(lin)
    assert lin.pq_packed_codes is None
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lin(torch.zeros(2, 64, dtype=torch.bfloat16))


def test_custom_ops_trace_with_fake_tensors():
    """The ops are torch custom ops with fake implementations (the counterpart of the reference's schema +
    register_fake pairs, vllm/_custom_ops.py:347-371): shapes and dtypes propagate under FakeTensorMode with no
    GPU and no library call, which is what torch.compile and graph capture rely on."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    from vllm_pq_amd import ops
    if not ops._HAVE_CUSTOM_OP:
        pytest.skip("this torch has no torch.library.custom_op")
    with FakeTensorMode():
        x = torch.empty(3, 5, 64, dtype=torch.bfloat16, device="cuda")
        codes = torch.empty(96, 8, 1, dtype=torch.int8, device="cuda")
        cb = torch.empty(2, 256, 1, 8, dtype=torch.bfloat16, device="cuda")
        sc = torch.empty(96, 1, 1, 1, dtype=torch.bfloat16, device="cuda")
        y = torch.ops.pq_amd.gemm(x, codes, cb, sc, [64, 32], None)
        assert y.shape == (3, 5, 96) and y.dtype == torch.bfloat16 and y.device.type == "cuda"
        packed = torch.empty(96 // 16 * 1024, dtype=torch.uint8, device="cuda")
        y = torch.ops.pq_amd.gemm_packed(x, packed, codes, cb, sc, [64, 32], None)
        assert y.shape == (3, 5, 96)
        w = torch.ops.pq_amd.dequant(codes, cb, [64, 32], None)
        assert w.shape == (96, 64) and w.dtype == torch.bfloat16
        out = torch.empty(3, 5, 32, dtype=torch.bfloat16, device="cuda")
        assert torch.ops.pq_amd.silu_and_mul(out, x) is None
        assert torch.ops.pq_amd.fused_add_rms_norm(x, torch.empty_like(x), torch.empty(64, dtype=torch.bfloat16,
                                                                                       device="cuda"), 1e-6) is None


def test_public_wrappers_route_fake_tensors_through_the_ops():
    """Plain eager CUDA tensors go straight to the implementation (no per-call dispatcher cost); anything else --
    fake tensors here -- takes the registered op and its fake implementation."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    from vllm_pq_amd import ops
    if not ops._HAVE_CUSTOM_OP:
        pytest.skip("this torch has no torch.library.custom_op")
    assert not ops._via_op(torch.empty(2, 2), None)          # a plain tensor: direct
    with FakeTensorMode():
        x = torch.empty(7, 64, dtype=torch.bfloat16, device="cuda")
        codes = torch.empty(96, 8, 1, dtype=torch.int8, device="cuda")
        cb = torch.empty(2, 256, 1, 8, dtype=torch.bfloat16, device="cuda")
        sc = torch.empty(96, 1, 1, 1, dtype=torch.bfloat16, device="cuda")
        assert ops._via_op(x, codes)
        assert ops.pq_gemm(x, codes, cb, sc, [64, 32], None).shape == (7, 96)
        packed = torch.empty(96 // 16 * 1024, dtype=torch.uint8, device="cuda")
        assert ops.pq_gemm_packed(x, packed, codes, cb, sc, [64, 32], None).shape == (7, 96)
        assert ops.pq_dequant(codes, cb, [64, 32]).shape == (96, 64)
        out = torch.empty(7, 32, dtype=torch.bfloat16, device="cuda")
        assert ops.silu_and_mul(out, x) is None


def test_rows_helper_refuses_tensors_it_would_have_to_copy():
    """The residual of the fused add + norm is updated in place through a 2-D view: a tensor whose leading dimensions
    do not collapse (a slice along the sequence dimension) must be refused, not silently copied."""
    import pytest
    import torch
    from vllm_pq_amd import ops
    full = torch.zeros(2, 6, 64, dtype=torch.bfloat16)
    ok = ops._rows16(full, 64, "residual")
    assert ok.shape == (12, 64) and ok.data_ptr() == full.data_ptr()
    sliced = full[:, :4, :]                       # [2, 4, 64] with a gap between the two batches
    with pytest.raises(ValueError, match="without a copy"):
        ops._rows16(sliced, 64, "residual")
    with pytest.raises(ValueError):
        ops._rows16(torch.zeros(3, 32, dtype=torch.bfloat16), 64, "residual")


@pytest.mark.parametrize("d", [2, 4])
def test_fold_pairs_is_the_same_weight_on_cpu(d):
    """ops.fold_pairs (host-side tensor operations, no kernel): the folded codes and 256-entry codebooks dequantise --
    with the CPU oracle -- to the weight the original 16-entry format gives, bit for bit, partition by partition;
    formats it cannot fold come back as None."""
    import numpy as np
    from oracle import pq_oracle as orc
    from vllm_pq_amd import ops
    g = torch.Generator().manual_seed(11 + d)
    parts, G = [32, 16, 48], 24
    N = sum(parts)
    codes = torch.randint(-128, 128, (N, G, 1), generator=g, dtype=torch.int8)      # stored values reduce mod 16
    cb = torch.randn(len(parts), 16, 1, d, generator=g).to(torch.bfloat16)
    codes2, cb2 = ops.fold_pairs(codes, cb)
    assert codes2.shape == (N, G // 2, 1) and codes2.dtype == torch.int8 and cb2.shape == (len(parts), 256, 1, 2 * d)
    bits = lambda t: t.contiguous().view(torch.int16).numpy().view(np.uint16)
    ones = np.full(N, 0x3F80, dtype=np.uint16)                                      # scale 1.0 in bf16
    want = orc.dequant(codes.numpy(), bits(cb).reshape(-1, 16, d), ones, parts, "bf16")
    got = orc.dequant(codes2.numpy(), bits(cb2).reshape(-1, 256, 2 * d), ones, parts, "bf16")
    assert np.array_equal(want, got)
    assert ops.fold_pairs(codes[:, :23], cb) is None                                # odd number of groups
    assert ops.fold_pairs(codes, torch.zeros(len(parts), 256, 1, d, dtype=torch.bfloat16)) is None
    assert ops.fold_pairs(codes, torch.zeros(len(parts), 16, 1, 8, dtype=torch.bfloat16)) is None
    assert ops.fold_pairs(codes.expand(N, G, 2), torch.zeros(2 * len(parts), 16, 1, d, dtype=torch.bfloat16)) is None
