"""pq_gemm_packed_noalloc (the entry point for callers that cannot allocate) against pq_gemm_packed with a caller-owned
workspace, through the C ABI: identical bits, on two calls in a row that share the built-in scratch (the second a
smaller problem), in a fresh process as well as after other tests have used the library."""
import ctypes

import pytest
import torch

from vllm_pq_amd import _lib, ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _problem(seed, T, parts, Kin, bias):
    g = torch.Generator(device="cpu").manual_seed(seed)
    N, G = sum(parts), Kin // 8
    codes = torch.randint(-128, 128, (N, G, 1), generator=g, dtype=torch.int32).to(torch.int8).to(DEV)
    cb = torch.randn(len(parts), 256, 1, 8, generator=g).to(torch.bfloat16).to(DEV)
    scales = (torch.rand(N, 1, 1, 1, generator=g) + 0.5).to(torch.bfloat16).to(DEV)
    x = torch.randn(T, Kin, generator=g).to(torch.bfloat16).to(DEV)
    b = torch.randn(N, generator=g).to(torch.bfloat16).to(DEV) if bias else None
    packed = ops.pq_repack_codes(codes, cb, parts)
    assert packed is not None
    return dict(T=T, parts=parts, N=N, G=G, codes=codes, cb=cb, scales=scales, x=x, bias=b, packed=packed)


def _call(L, p, noalloc, ws=None):
    y = torch.full((p["T"], p["N"]), float("nan"), dtype=torch.bfloat16, device=DEV)
    head = (p["x"].data_ptr(), p["T"], p["x"].stride(0), p["packed"].data_ptr(), p["cb"].data_ptr(), p["scales"].data_ptr(),
            p["bias"].data_ptr() if p["bias"] is not None else None, _lib.parts_array(p["parts"]), len(p["parts"]), p["N"],
            p["G"], 1, 256, 8, 1, _lib.PQ_DTYPE_BF16, y.data_ptr(), p["N"])
    stream = torch.cuda.current_stream(torch.device(DEV)).cuda_stream
    if noalloc:
        rc = L.pq_gemm_packed_noalloc(*head, stream)
    else:
        rc = L.pq_gemm_packed(*head, ws.data_ptr(), ws.numel(), stream)
    assert rc == 0, _lib.last_error()
    return y


def test_noalloc_entry_point_matches_workspace_entry_point():
    L = _lib.load()
    # a long-K layer whose launches split K over blocks through the scratch, then a smaller one of the same kind, then
    # shapes that need no scratch (decode kernel, tiled kernel)
    big = _problem(1, 16, [4096], 14336, bias=True)
    small = _problem(2, 5, [1024, 512], 512 * 17, bias=False)
    others = [_problem(3, 16, [2048, 512], 4096, bias=False), _problem(4, 300, [512], 2048, bias=True)]
    assert ops.workspace_bytes(16, big["N"], big["G"], 1, 256, 8, 1, torch.bfloat16) > 0
    assert ops.workspace_bytes(5, small["N"], small["G"], 1, 256, 8, 1, torch.bfloat16) > 0
    ws = torch.zeros(int(L.pq_gemm_workspace_max_bytes()), dtype=torch.uint8, device=DEV)
    with torch.cuda.device(torch.device(DEV)):
        for p in (big, small, *others, big):
            want = _call(L, p, False, ws)
            before = torch.cuda.memory_allocated()
            got = _call(L, p, True)
            torch.cuda.synchronize()
            assert torch.cuda.memory_allocated() - before <= got.numel() * 2 + 512    # only the test's own output tensor
            assert not torch.isnan(got.float()).any()
            assert torch.equal(got, want), f"T={p['T']} N={p['N']}: {int((got != want).sum())} values differ"


def test_noalloc_entry_point_rejects_what_its_scratch_cannot_hold():
    L = _lib.load()
    # 8192 rows x 14336 inputs at 128 tokens: the preferred split takes 16 MiB of partial sums
    T, N, G = 128, 8192, 14336 // 8
    al = ctypes.c_void_p(256)
    rc = L.pq_gemm_packed_noalloc(al, T, G * 8, al, al, al, None, _lib.parts_array([N]), 1, N, G, 1, 256, 8, 1,
                                  _lib.PQ_DTYPE_BF16, al, N, None)
    assert rc == 3
    msg = _lib.last_error()
    assert "built-in buffer" in msg and "pq_gemm_packed" in msg
