#!/usr/bin/env python3
"""Experiment: what a decode launch gains when its codes were pulled into the L2 of the XCD that reads them before it
starts.  Llama-3-8B layers at T tokens, 48 copies of the weights in rotation (HBM-cold without the touch).  A touch
launch (tools/pq_probe.hip: one dword per 128-byte line, block b -> the 1/256th of the codes block b of the GEMM reads)
runs before each GEMM:  none | right (same copy, same XCD) | wrong (same copy, neighbouring XCD) | early (the NEXT
GEMM's copy: the lines have to survive one launch).  Per-kernel durations come from rocprofv3 --kernel-trace --stats
run around this script.

  python tools/prefetch_probe.py <mode> [T]
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_pq_amd import ops

mode = sys.argv[1] if len(sys.argv) > 1 else "none"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
probe = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpq_probe.so"))
probe.pq_probe_touch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_void_p, ctypes.c_void_p]
dev, dt = "cuda:0", torch.bfloat16
sink = torch.zeros(16, device=dev, dtype=torch.int32)
for name, parts, kin, blocks in (("gate_up", [14336, 14336], 4096, 256), ("o_proj", [4096], 4096, 256),
                                 ("down", [4096], 14336, 256), ("qkv", [4096, 1024, 1024], 4096, 192)):
    N, G = sum(parts), kin // 8
    mb = N * G / 1e6
    copies = max(2, int(700 / mb))
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
    nbytes = ws[0][3].numel() * ws[0][3].element_size()
    per_block = (nbytes // blocks + 127) // 128 * 128

    def touch(w, shift):
        st = torch.cuda.current_stream().cuda_stream
        rc = probe.pq_probe_touch(w[3].data_ptr(), per_block, nbytes, blocks, shift, sink.data_ptr(), st)
        assert rc == 0

    n = copies * 4
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n):
            w = ws[i % copies]
            if mode == "right":
                touch(w, 0)
            elif mode == "wrong":
                touch(w, 1)
            elif mode == "early":
                touch(ws[(i + 1) % copies], 0)
            ops.pq_gemm_packed(x, w[3], w[0], w[1], w[2], parts, None)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"{name:8s} T={T:3d} mode={mode:6s} copies={copies:3d} {nbytes / 1e6:6.1f} MB codes, {per_block} B a block: "
          f"{e0.elapsed_time(e1) * 1e3 / n:6.2f} us per (touch +) launch", flush=True)
    del ws, g
    torch.cuda.empty_cache()
