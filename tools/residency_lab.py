#!/usr/bin/env python3
"""Experiment: how much shorter is a decode launch whose codes come from the memory-side cache (or the L2) instead of
HBM?  gate_up_proj / o_proj of Llama-3-8B at T tokens, cycling through 1 (L2), 8 (memory-side cache) or 48 (HBM)
copies of the layer's weights, many launches in one HIP graph."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_pq_amd import ops

T = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev, dt = "cuda:0", torch.bfloat16
for name, parts, kin in (("gate_up", [14336, 14336], 4096), ("o_proj", [4096], 4096), ("down", [4096], 14336)):
    N, G = sum(parts), kin // 8
    mb = N * G / 1e6
    for copies in (1, max(2, int(100 / mb)), max(2, int(700 / mb))):
        ws = []
        for _ in range(copies):
            codes = torch.randint(-128, 128, (N, G, 1), device=dev, dtype=torch.int8)
            cbk = torch.randn(len(parts), 256, 1, 8, device=dev).to(dt)
            sc = torch.ones(N, 1, 1, 1, device=dev, dtype=dt)
            ws.append((codes, cbk, sc, ops.pq_repack_codes(codes, cbk, parts)))
        x = torch.randn(T, kin, device=dev).to(dt)
        ops.ensure_workspace(torch.device(dev))
        for w in ws[:2]:
            ops.pq_gemm_packed(x, w[3], w[0], w[1], w[2], parts, None)
        torch.cuda.synchronize()
        n = max(copies * 2, 200)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(n):
                w = ws[i % copies]
                ops.pq_gemm_packed(x, w[3], w[0], w[1], w[2], parts, None)
        g.replay()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / n)
        print(f"{name:8s} T={T:3d} copies={copies:3d} ({copies * mb:7.1f} MB): {best:6.2f} us/launch", flush=True)
        del ws, g
        torch.cuda.empty_cache()
