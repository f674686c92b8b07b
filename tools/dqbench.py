#!/usr/bin/env python3
"""Times pq_dequant (dense-weight reconstruction) against the HBM roofline: bytes = codes read + weight written."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_pq_amd import ops
dev, dt = 'cuda', torch.bfloat16
for name, parts, Kin, K, d in (("gate_up", [14336, 14336], 4096, 256, 8), ("down", [4096], 14336, 256, 8),
                               ("qkv", [4096, 1024, 1024], 4096, 256, 8), ("sq8192.K16.d2", [8192], 8192, 16, 2),
                               ("sq8192.K4096.d4", [8192], 8192, 4096, 4), ("sq8192.K4096.d8", [8192], 8192, 4096, 8)):
    N, G = sum(parts), Kin // d
    idt = torch.int8 if K <= 256 else torch.int16
    sets = []
    for _ in range(3):
        codes = torch.randint(-(K // 2), K // 2, (N, G, 1), device=dev, dtype=torch.int32).to(idt)
        cb = torch.randn(len(parts), K, 1, d, device=dev).to(dt)
        sc = (torch.rand(N, 1, 1, 1, device=dev) + .5).to(dt)
        sets.append((codes, cb, sc))
    for scaled in (False, True):
        f = lambda s: ops.pq_dequant(s[0], s[1], parts, s[2] if scaled else None)
        f(sets[0]); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            keep = [f(s) for s in sets]
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / len(sets))
        del g, keep
        byts = N * G * codes.element_size() + N * Kin * 2
        print(f"{name:18s} scaled={scaled!s:5s} {best:8.1f} us  {byts/best*1e-3:8.1f} GB/s  frac {byts/best*1e-3/8000:.3f}")
