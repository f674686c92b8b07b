#!/usr/bin/env python3
"""Times the PQ vocabulary table of Llama-3-8B (128256 x 4096, K=256, d=8, bf16): the LM head's logits GEMM and the
embedding-row lookup, against their HBM / MFMA rooflines (HIP-graph replays over rotating copies)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_pq_amd import ops  # noqa: E402

dev, dt = "cuda", torch.bfloat16
V, H, K, d = 128256, 4096, 256, 8
G = H // d
sets = []
for _ in range(6):      # 6 x 66 MB of codes: more than the Infinity Cache
    codes = torch.randint(-128, 128, (V, G, 1), device=dev, dtype=torch.int32).to(torch.int8)
    cb = torch.randn(1, K, 1, d, device=dev).to(dt)
    sc = (torch.rand(V, 1, 1, 1, device=dev) + .5).to(dt)
    sets.append((codes, cb, sc, ops.pq_repack_codes(codes, cb, [V])))
ops.ensure_workspace(torch.device(dev), ops.workspace_bytes(128, V, G, 1, K, d, 1, dt))


def timed(fn):
    fn(sets[0]); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = [fn(s) for s in sets]
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3 / len(sets))
    del g, keep
    return best


for T in (1, 16, 64, 256, 1024, 4096):
    x = torch.randn(T, H, device=dev).to(dt)
    t = timed(lambda s: ops.pq_gemm_packed(x, s[3], s[0], s[1], s[2], [V], None))
    byts = V * G + K * d * 2 + V * 2 + T * H * 2 + T * V * 2
    flops = 2.0 * T * V * H
    bound = max(byts / 8e12, flops / 2.5e15)
    print(f"lm_head   T={T:5d} {t*1e6:9.1f} us  {byts/t/1e9:8.1f} GB/s  {flops/t/1e12:8.1f} TFLOP/s  frac {bound/t:.3f} "
          f"({ops.gemm_kernel_name(T, V, G, 1, K, d, 1, dt, [V], packed=True)})", flush=True)
for T in (16, 256, 4096, 16384):
    ids = torch.randint(0, V, (T,), device=dev)
    t = timed(lambda s: ops.pq_embedding(ids, s[0], s[1], s[2], [V]))
    byts = T * G + T * 2 + T * 8 + T * H * 2
    print(f"embedding T={T:5d} {t*1e6:9.1f} us  {byts/t/1e9:8.1f} GB/s  frac {byts/t/8e12:.3f}", flush=True)
