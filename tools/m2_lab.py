#!/usr/bin/env python3
"""Times the two-codebook format (K = 256, M = 2, d = 8: AQLM 2x8) beside the one-codebook one on the Llama-3-8B
shapes:  python tools/m2_lab.py [tokens,...] [codebooks,...] [shape,...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_pq_amd import ops

dev, dt = "cuda:0", torch.bfloat16
TS = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "1,16,256,4096").split(",")]
SHAPES = (("qkv", [4096, 1024, 1024], 4096), ("o", [4096], 4096), ("gate_up", [14336, 14336], 4096), ("down", [4096], 14336))
MS = [int(m) for m in (sys.argv[2] if len(sys.argv) > 2 else "1,2").split(",")]
ONLY = sys.argv[3].split(",") if len(sys.argv) > 3 else None
for name, parts, kin in SHAPES:
    if ONLY and name not in ONLY:
        continue
    N, G = sum(parts), kin // 8
    for M in MS:
        copies = max(2, int(400e6 / (N * G * M)))
        ws = []
        for _ in range(copies):
            codes = torch.randint(-128, 128, (N, G, M), device=dev, dtype=torch.int8)
            cbk = torch.randn(len(parts) * M, 256, 1, 8, device=dev).to(dt)
            sc = torch.ones(N, 1, 1, 1, device=dev, dtype=dt)
            ws.append((codes, cbk, sc, ops.pq_repack_codes(codes, cbk, parts)))
        ops.ensure_workspace(torch.device(dev))
        for T in TS:
            x = torch.randn(T, kin, device=dev).to(dt)
            for w in ws[:2]:
                ops.pq_gemm_packed(x, w[3], w[0], w[1], w[2], parts, None)
            torch.cuda.synchronize()
            n = min(copies, 8) if T >= 1024 else copies
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for w in ws[:n]:
                    ops.pq_gemm_packed(x, w[3], w[0], w[1], w[2], parts, None)
            g.replay(); torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / n)
            kern = ops.gemm_kernel_name(T, N, G, M, 256, 8, 1, dt, parts, packed=True)
            fl = 2.0 * T * N * kin
            print(f"{name:8s} M={M} T={T:5d}: {best:8.2f} us  {fl / best / 1e6:7.1f} TFLOP/s  {N * G * M / best / 1e3:7.1f} GB/s codes  {kern}", flush=True)
            del g
        del ws
        torch.cuda.empty_cache()
