"""The C ABI library loads without a GPU, exports every symbol include/pq_amd.h declares, and rejects
bad arguments before touching the device.  CPU only (no compute calls)."""
import ctypes
import os
import re

import pytest
import torch

from vllm_pq_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "pq_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pq_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_are_exported():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_functions()
    assert "pq_gemm" in names and "pq_dequant" in names and "pq_repack_codes" in names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/pq_amd.h but not exported"
    assert sorted(_lib.EXPORTS) == names, "vllm_pq_amd/_lib.py EXPORTS out of sync with the header"


def test_version_and_caps():
    L = _lib.load()
    assert L.pq_version() == 1
    caps = L.pq_caps()
    assert caps & 1 and caps & 2 and caps & (1 << 4) and caps & (1 << 5)


def _gemm_rc(L, **kw):
    a = dict(x=256, T=4, xs=128, codes=256, cb=256, sc=256, bias=None, parts=[64], N=64, G=16, M=1, K=256, d=8,
             code_bytes=1, dtype=1, y=256, ys=64)
    a.update(kw)
    pa = _lib.parts_array(a["parts"]) if a["parts"] is not None else None
    return L.pq_gemm(a["x"], a["T"], a["xs"], a["codes"], a["cb"], a["sc"], a["bias"], pa,
                     len(a["parts"]) if a["parts"] is not None else 1, a["N"], a["G"], a["M"], a["K"], a["d"],
                     a["code_bytes"], a["dtype"], a["y"], a["ys"], None, 0, None)


def test_argument_validation_needs_no_gpu():
    L = _lib.load()
    # sum(part_sizes) != N  (aqlm.py:112)
    assert _gemm_rc(L, parts=[32, 16]) == 1 and "sum(part_sizes)" in _lib.last_error()
    # K not a power of two (aqlm.py:86)
    assert _gemm_rc(L, K=200) == 1 and "power of two" in _lib.last_error()
    # K too large for int8 codes (get_int_dtype, aqlm.py:20-29)
    assert _gemm_rc(L, K=4096) == 1
    assert _gemm_rc(L, code_bytes=4) == 2            # int32 codes: unsupported
    assert _gemm_rc(L, dtype=7) == 1
    assert _gemm_rc(L, xs=64) == 1                   # row stride shorter than in_features
    assert _gemm_rc(L, ys=8) == 1
    assert _gemm_rc(L, parts=[8] * 9, N=72) == 1     # more than PQ_MAX_PARTS partitions
    assert _gemm_rc(L, x=None) == 1 and "NULL" in _lib.last_error()
    # empty problems are fine and launch nothing
    assert _gemm_rc(L, T=0) == 0
    assert _gemm_rc(L, N=0, parts=[0], ys=0) == 0


def test_packed_size_and_kernel_choice():
    L = _lib.load()
    pa = _lib.parts_array([4096, 1024, 1024])
    # one KiB per 16-row fragment and 64-group chunk
    assert L.pq_packed_codes_bytes(pa, 3, 6144, 512, 1, 256, 8, 1) == (6144 // 16) * 8 * 1024
    assert L.pq_packed_codes_bytes(_lib.parts_array([40, 24]), 2, 64, 16, 1, 256, 8, 1) == 0    # partitions % 16
    assert L.pq_packed_codes_bytes(_lib.parts_array([64]), 1, 64, 64, 1, 65536, 8, 2) == 0      # K too large for LDS
    assert L.pq_packed_codes_bytes(_lib.parts_array([64]), 1, 64, 6, 1, 256, 2, 1) == 0         # rows of 12 features
    assert L.pq_packed_codes_bytes(_lib.parts_array([64]), 1, 64, 8, 1, 256, 2, 1) > 0          # rows of 16
    from vllm_pq_amd import ops
    name = ops.gemm_kernel_name
    assert name(1, 6144, 512, 1, 256, 8, 1, torch.bfloat16, [4096, 1024, 1024], packed=True) == "mfma_decode_packed"
    assert name(4096, 6144, 512, 1, 256, 8, 1, torch.bfloat16, [4096, 1024, 1024], packed=True) == "mfma_tiled_packed"
    assert name(4, 64, 64, 1, 65536, 8, 2, torch.float16, [64]) == "generic"
    assert name(4, 60, 17, 1, 256, 8, 1, torch.float16, [60]) == "generic"
    with pytest.raises(ValueError):
        name(4, 64, 16, 1, 256, 8, 1, torch.bfloat16, [40, 24], packed=True)


def test_ops_raise_on_cpu_tensors():
    from vllm_pq_amd import ops
    codes = torch.zeros(64, 16, 1, dtype=torch.int8)
    cb = torch.zeros(1, 256, 1, 8, dtype=torch.bfloat16)
    sc = torch.ones(64, 1, 1, 1, dtype=torch.bfloat16)
    x = torch.zeros(2, 128, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.pq_gemm(x, codes, cb, sc, [64], None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.pq_dequant(codes, cb, [64])


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.PQLibraryError, match="no fallback"):
        _lib.load()


def test_kernels_do_not_spill_registers():
    """The MFMA kernels hold inline-asm loads in flight in registers the compiler does not track; a spill could save
    such a register before its load has returned.  `make check-spills` compiles the kernel translation units to
    assembly (cached under csrc/build) and fails on any non-zero vector-register spill count."""
    import shutil
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc"
    if not (os.path.exists(hipcc) or shutil.which("hipcc")):
        pytest.skip("hipcc not available")
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vllm_pq_amd", "csrc")
    r = subprocess.run(["make", "-j4", "check-spills"], cwd=csrc, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "no register spills" in r.stdout
    assert "in-flight lint: 0 violation(s)" in r.stdout


def test_inflight_lint_catches_violations(tmp_path):
    """The lint itself: reading, overwriting or branching with a hand-issued load in flight is reported; the same
    code with the landing marker in place is clean."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("lint_inflight", os.path.join(root, "tools", "lint_inflight.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    head = "kern:\n\t; PQI v5\n\tglobal_load_dword v5, v1, s[2:3]\n\t; PQI v[8:11]\n\tds_read_b128 v[8:11], v2 offset:16\n"
    cases = {
        "clean": head + "\tv_add_u32_e32 v3, v1, v2\n\ts_waitcnt vmcnt(0)\n\t; PQL v5\n\t; PQL v[8:11]\n\tv_mov_b32_e32 v6, v5\n\ts_endpgm\n",
        "read": head + "\tv_mov_b32_e32 v6, v5\n\t; PQL v5\n\t; PQL v[8:11]\n\ts_endpgm\n",
        "write": head + "\tv_mov_b32_e32 v9, v1\n\t; PQL v5\n\t; PQL v[8:11]\n\ts_endpgm\n",
        "branch": head + "\ts_cbranch_scc1 .LBB0_1\n\t; PQL v5\n\t; PQL v[8:11]\n\ts_endpgm\n",
        "never_landed": head + "\ts_endpgm\n",
    }
    got = {}
    for name, text in cases.items():
        f = tmp_path / (name + ".s")
        f.write_text(text)
        got[name] = lint.lint(str(f))
    assert got["clean"] == 0
    assert all(got[k] >= 1 for k in ("read", "write", "branch", "never_landed")), got
    # "; PQR": in flight across branches, checked along the control flow -- a block laid out behind the landing marker
    # that touches the register is found, the same block left alone is not
    head_r = ("kern2:\n\t; PQR v5\n\tglobal_load_dword v5, v1, s[2:3]\n\ts_cbranch_scc1 .LBB1_2\n"
              "\tv_add_u32_e32 v3, v1, v2\n.LBB1_1:\n\ts_waitcnt vmcnt(0)\n\t; PQL v5\n\tv_mov_b32_e32 v6, v5\n\ts_endpgm\n.LBB1_2:\n")
    for name, tail, want in (("reach_clean", "\tv_mov_b32_e32 v7, v1\n\ts_branch .LBB1_1\n", 0),
                             ("reach_touch", "\tv_mov_b32_e32 v7, v5\n\ts_branch .LBB1_1\n", 1)):
        f = tmp_path / (name + ".s")
        f.write_text(head_r + tail)
        assert lint.lint(str(f)) == 0, name          # the scan in text order cannot see the block behind the marker
        assert (lint.lint_reach(str(f)) >= 1) == bool(want), name
