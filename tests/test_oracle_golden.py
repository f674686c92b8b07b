"""Pins the CPU oracle (oracle/pq_oracle.c) to the golden vectors that the reference's own Python
functions produced (tests/golden/make_golden.py; aqlm.py:20-127).  CPU only."""
import numpy as np
import pytest

from conftest import golden
from oracle import pq_oracle as orc

G = golden()


@pytest.fixture(autouse=True)
def _torch_cpu_sum_rounding():
    # the golden vectors come from torch's CPU kernels; for M >= 2 bf16 codebooks they round the
    # sum over codebooks with ties away from zero (see oracle/pq_oracle.c)
    orc.set_bf16_sum_rounding("torch_cpu")
    yield
    orc.set_bf16_sum_rounding("rne")


def _id(c):
    return f"{c['id']}-K{c['K']}d{c['d']}M{c['M']}-{c['dtype']}"


def test_unpack_matches_reference_edge_values():
    np_dt = {"int8": np.int8, "int16": np.int16}
    for rec in G.manifest["unpack"]:
        vals = np.array(rec["in"], dtype=np_dt[rec["dtype"]])
        out = orc.unpack(vals.reshape(1, -1, 1), rec["nbits"]).reshape(-1)
        assert out.tolist() == rec["out"], rec


def test_fp16_bf16_conversions_exact():
    L = orc.lib()
    allbits = np.arange(65536, dtype=np.uint16)
    for dt in ("fp16", "bf16"):
        want = orc.to_f32(allbits, dt)
        got = np.array([L.pq_oracle_act_to_f32(int(b), orc._DT[dt]) for b in allbits], np.float32)
        ok = (want == got) | (np.isnan(want) & np.isnan(got))
        assert ok.all(), dt
    rng = np.random.default_rng(0)
    f = np.concatenate([rng.standard_normal(20000).astype(np.float32) * s for s in (1e-8, 1e-5, 1e-3, 1, 1e3, 7e4)])
    f = np.concatenate([f, np.array([0, -0.0, 65504, 65519.9, 65520, 1e9, -1e9, 2 ** -24, 2 ** -25, 2 ** -25 * 1.0001,
                                     6.1e-5, 6.0975e-5], np.float32)])
    for dt in ("fp16", "bf16"):
        want = orc.from_f32(f, dt)
        got = np.array([L.pq_oracle_f32_to_act(float(v), orc._DT[dt]) for v in f], np.uint16)
        assert (want == got).all(), (dt, f[want != got][:5])


@pytest.mark.parametrize("case", G.cases("gather"), ids=_id)
def test_gather_bit_exact(case):
    out = orc.dequant(G.get(case, "codes"), G.get(case, "codebooks").reshape(-1, case["K"], case["d"]), None,
                      case["parts"], case["dtype"])
    assert np.array_equal(out, G.get(case, "out"))


@pytest.mark.parametrize("case", G.cases("dequant"), ids=_id)
def test_scaled_dequant_bit_exact(case):
    out = orc.dequant(G.get(case, "codes"), G.get(case, "codebooks").reshape(-1, case["K"], case["d"]),
                      G.get(case, "scales").reshape(-1), case["parts"], case["dtype"])
    want = G.get(case, "out")
    assert np.array_equal(out, want), f"{(out != want).sum()} of {out.size} differ"


# Tolerance for the GEMM: the reference's CPU F.linear and the oracle both accumulate in >= fp32 and
# round once, but in different orders; results may differ by one ulp of the act dtype on a few outputs.
@pytest.mark.parametrize("case", G.cases("gemm"), ids=_id)
def test_gemm_matches_reference(case):
    dt = case["dtype"]
    args = (G.get(case, "x"), G.get(case, "codes"), G.get(case, "codebooks").reshape(-1, case["K"], case["d"]),
            G.get(case, "scales").reshape(-1), case["parts"], G.get(case, "bias"), dt)
    y, y32 = orc.gemm(*args, order=0, acc_double=True, want_f32=True)
    want = G.get(case, "out")
    wf, yf = orc.to_f32(want, dt), orc.to_f32(y, dt)
    ulp = 2.0 ** (-7 if dt == "bf16" else -10)
    # every output within one act-dtype ulp (relative to its own magnitude, floor for cancellation)
    scale = np.maximum(np.abs(y32), np.abs(y32).mean() * 0.05)
    err = np.abs(wf - yf) / scale
    assert err.max() <= 1.01 * ulp, (err.max(), ulp)
    # and the large majority bit-identical
    assert (want == y).mean() > 0.9, (want == y).mean()
    # fp32-accumulating variant (used for the timed CPU baseline) agrees to the same bar
    y_f = orc.gemm(*args, order=0, acc_double=False)
    assert (np.abs(orc.to_f32(y_f, dt) - yf) / scale).max() <= 1.01 * ulp


@pytest.mark.parametrize("case", G.cases("gemm"), ids=_id)
def test_torch_restatement_matches_reference(case):
    """The torch-CPU restatement timed as bench.py's cpu_baseline uses the reference's operators in the
    reference's order, so it reproduces the golden outputs bit for bit."""
    import torch
    from oracle.torch_restatement import pq_linear_cpu
    from pq_testlib import bits_of, tensor_of
    dt = case["dtype"]
    K, d = case["K"], case["d"]
    x = tensor_of(G.get(case, "x"), dt)
    codes = torch.from_numpy(G.get(case, "codes"))
    cb = tensor_of(G.get(case, "codebooks"), dt).view(-1, K, 1, d)
    sc = tensor_of(G.get(case, "scales"), dt).view(-1, 1, 1, 1)
    b = G.get(case, "bias")
    b = None if b is None else tensor_of(b, dt)
    y = pq_linear_cpu(x, codes, cb, sc, case["parts"], b)
    assert np.array_equal(bits_of(y), G.get(case, "out"))
