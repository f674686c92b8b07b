#!/usr/bin/env python3
"""What the library GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on the dense bf16 problem the fused PQ kernel
replaces, on this device: y[T][N] = x[T][Kin] @ W[N][Kin]^T with W already dequantised (the reference's route above 6
tokens, aqlm.py:141-145), next to pq_dequant's time for the same layer and the fused kernel's."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_pq_amd import ops

dev, dt = "cuda:0", torch.bfloat16


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


for name, parts, kin in (("gate_up", [14336, 14336], 4096), ("down", [4096], 14336), ("qkv", [4096, 1024, 1024], 4096),
                         ("o", [4096], 4096)):
    N, G = sum(parts), kin // 8
    codes = torch.randint(-128, 128, (N, G, 1), device=dev, dtype=torch.int8)
    cbk = torch.randn(len(parts), 256, 1, 8, device=dev).to(dt)
    sc = (torch.rand(N, 1, 1, 1, device=dev) + 0.5).to(dt)
    packed = ops.pq_repack_codes(codes, cbk, parts)
    w = ops.pq_dequant(codes, cbk, parts)
    t_dq = timeit(lambda: ops.pq_dequant(codes, cbk, parts))
    for T in (512, 1024, 2048, 4096, 8192):
        x = torch.randn(T, kin, device=dev).to(dt)
        t_mm = timeit(lambda: torch.matmul(x, w.t()))
        t_pq = timeit(lambda: ops.pq_gemm_packed(x, packed, codes, cbk, sc, parts, None))
        fl = 2.0 * T * N * kin
        print(f"{name:8s} T={T:5d}  library GEMM {t_mm:8.1f} us = {fl / t_mm * 1e-6:7.1f} TFLOP/s   dequant {t_dq:6.1f} us   "
              f"dequant + GEMM {t_dq + t_mm:8.1f} us   fused PQ kernel {t_pq:8.1f} us = {fl / t_pq * 1e-6:7.1f} TFLOP/s", flush=True)
