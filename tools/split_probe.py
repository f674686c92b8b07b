#!/usr/bin/env python3
"""Experiment: a prefill launch whose tiles do not fill whole rounds of the chip (gate_up_proj at 2048 tokens: 896 tiles
of 512 x 128 over 256 blocks = 3.5 rounds, paid as 4), against the same layer cut along its rows into a part that fills
whole rounds with 512 x 128 tiles and a rest on 256 x 128 tiles.  The two parts are separate weight tensors here."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_pq_amd import _lib, ops

dev, dt = "cuda:0", torch.bfloat16


def layer(parts, kin):
    N, G = sum(parts), kin // 8
    codes = torch.randint(-128, 128, (N, G, 1), device=dev, dtype=torch.int8)
    cb = torch.randn(len(parts), 256, 1, 8, device=dev).to(dt)
    sc = (torch.rand(N, 1, 1, 1, device=dev) + 0.5).to(dt)
    return codes, cb, sc, ops.pq_repack_codes(codes, cb, parts), parts


def graph_of(calls, x, copies):
    g = torch.cuda.CUDAGraph()
    for lay, cfg in calls(0):
        pass
    with torch.cuda.graph(g):
        for c in range(copies):
            for lay, cfg in calls(c):
                os.environ["PQ_AMD_TILED_CFG"] = cfg
                os.environ["PQ_AMD_FORCE_KERNEL"] = "tiled" if cfg else ""
                _lib.reload_knobs()
                ops.pq_gemm_packed(x, lay[3], lay[0], lay[1], lay[2], lay[4], None)
    return g


def timeit(g, n):
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


for T, whole_parts, a_parts, b_parts in ((2048, [14336, 14336], [14336, 10240], [4096]),
                                         (3072, [14336, 14336], [14336, 12800], [1536]),
                                         (2560, [14336, 14336], [14336, 11776], [2560])):
    x = torch.randn(T, 4096, device=dev).to(dt)
    copies = 6
    whole = [layer(whole_parts, 4096) for _ in range(copies)]
    pa = [layer(a_parts, 4096) for _ in range(copies)]
    pb = [layer(b_parts, 4096) for _ in range(copies)]
    for lay in (whole[0], pa[0], pb[0]):
        ops.pq_gemm_packed(x, lay[3], lay[0], lay[1], lay[2], lay[4], None)
    torch.cuda.synchronize()
    os.environ["PQ_AMD_TILED_SPLIT"] = "0"
    t_single = timeit(graph_of(lambda c: [(whole[c], "")], x, copies), copies)
    del os.environ["PQ_AMD_TILED_SPLIT"]
    t_whole = timeit(graph_of(lambda c: [(whole[c], "")], x, copies), copies)
    t_split = timeit(graph_of(lambda c: [(pa[c], "6"), (pb[c], "1")], x, copies), copies)
    t_a = timeit(graph_of(lambda c: [(pa[c], "6")], x, copies), copies)
    t_b = timeit(graph_of(lambda c: [(pb[c], "1")], x, copies), copies)
    print(f"T={T}: one launch {t_single:7.1f} us   library's plan {t_whole:7.1f} us   separate tensors {t_split:7.1f} us ({a_parts} on 512x128: {t_a:.1f}, {b_parts} on 256x128: {t_b:.1f})", flush=True)
os.environ["PQ_AMD_TILED_CFG"] = ""
os.environ["PQ_AMD_FORCE_KERNEL"] = ""
