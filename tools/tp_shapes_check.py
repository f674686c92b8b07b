#!/usr/bin/env python3
"""Per-rank shapes of Llama-3-8B at TP = 2, 4, 8 (what `bench.py --gpus N` runs on every rank) through the packed entry
point at the bench's token counts, against an fp32 matmul over the bit-exact dequantised weight."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_pq_amd import ops

dev, dt = "cuda:0", torch.bfloat16
ok = True
for tp in (2, 4, 8):
    shapes = (("qkv", [4096 // tp, 1024 // tp, 1024 // tp], 4096), ("o", [4096], 4096 // tp),
              ("gate_up", [14336 // tp, 14336 // tp], 4096), ("down", [4096], 14336 // tp))
    for name, parts, kin in shapes:
        N, G = sum(parts), kin // 8
        torch.manual_seed(tp * 100 + N)
        codes = torch.randint(-128, 128, (N, G, 1), device=dev, dtype=torch.int8)
        cb = torch.randn(len(parts), 256, 1, 8, device=dev).to(dt)
        sc = (torch.rand(N, 1, 1, 1, device=dev) + 0.5).to(dt)
        packed = ops.pq_repack_codes(codes, cb, parts)
        ops.ensure_workspace(torch.device(dev))
        w = ops.pq_dequant(codes, cb, parts).float()
        for T in (16, 256, 4096):
            x = torch.randn(T, kin, device=dev).to(dt)
            y = ops.pq_gemm_packed(x, packed, codes, cb, sc, parts, None) if packed is not None else ops.pq_gemm(x, codes, cb, sc, parts, None)
            ref = (x.float() @ w.t()) * sc.view(1, -1).float()
            tol = 0.51 * 2.0 ** -7 * ref.abs() + 1e-4 * ref.pow(2).mean().sqrt()
            bad = int(((y.float() - ref).abs() > tol).sum())
            kern = ops.gemm_kernel_name(T, N, G, 1, 256, 8, 1, dt, parts, packed=packed is not None)
            print(f"tp{tp} {name:8s} T={T:5d} N={N:6d} Kin={kin:6d} {kern:22s} {'ok' if bad == 0 else 'BAD %d' % bad}", flush=True)
            ok = ok and bad == 0
sys.exit(0 if ok else 1)
