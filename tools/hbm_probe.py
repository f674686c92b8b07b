#!/usr/bin/env python3
"""What plain streaming reaches on this device, as a yardstick for the element-wise kernels: torch's copy (read n, write n)
and a read-only sum, on buffers larger than the Infinity Cache, next to silu_and_mul and fused_add_rms_norm at 4096 tokens."""
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
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3 / reps)
    return best


n = 512 * 1024 * 1024 // 2
srcs = [torch.randn(n, device=dev).to(dt) for _ in range(3)]
dst = torch.empty(n, device=dev, dtype=dt)
i = [0]


def copy():
    i[0] = (i[0] + 1) % 3
    dst.copy_(srcs[i[0]])


def rsum():
    i[0] = (i[0] + 1) % 3
    return srcs[i[0]].view(torch.int32).sum()


t = timeit(copy)
print(f"copy 512 MiB -> 512 MiB: {t * 1e6:8.1f} us = {2 * n * 2 / t / 1e12:5.2f} TB/s")
t = timeit(rsum)
print(f"sum over 512 MiB (read only): {t * 1e6:8.1f} us = {n * 2 / t / 1e12:5.2f} TB/s")
T, I, H = 4096, 14336, 4096
xs = [torch.randn(T, 2 * I, device=dev).to(dt) for _ in range(3)]
out = torch.empty(T, I, device=dev, dtype=dt)


def silu():
    i[0] = (i[0] + 1) % 3
    ops.silu_and_mul(out, xs[i[0]])


t = timeit(silu)
print(f"silu_and_mul {T} x {I}: {t * 1e6:8.1f} us = {T * I * 6 / t / 1e12:5.2f} TB/s")
hs = [torch.randn(T, H, device=dev).to(dt) for _ in range(8)]
rs = [torch.randn(T, H, device=dev).to(dt) for _ in range(8)]
w = torch.ones(H, device=dev, dtype=dt)


def norm():
    i[0] = (i[0] + 1) % 8
    ops.fused_add_rms_norm(hs[i[0]], rs[i[0]], w, 1e-5)


t = timeit(norm)
print(f"fused_add_rms_norm {T} x {H}: {t * 1e6:8.1f} us = {T * H * 8 / t / 1e12:5.2f} TB/s")
