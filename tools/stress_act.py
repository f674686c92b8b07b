#!/usr/bin/env python3
"""Randomised check of the fused gate_up activation launches against the two steps (bit for bit), many layer sizes and
token counts:  python tools/stress_act.py [cases] [seed]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from pq_testlib import Problem  # noqa: E402
from vllm_pq_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
DEV, bad = "cuda:0", 0
ops.ensure_workspace(torch.device(DEV))
for i in range(n):
    H = 512 * int(rng.integers(1, 9))
    I = 16 * int(rng.integers(1, 200))
    T = int(rng.integers(1, 129))
    bias = bool(rng.integers(0, 2))
    dtype = "bf16" if rng.integers(0, 2) else "fp16"
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    p = Problem(10000 + i, 256, 8, 1, [I, I], H, 1, dtype, bias=bias).to(DEV)
    pk = ops.pq_repack_codes(p.codes_dev, p.codebooks_dev, [I, I])
    g = torch.Generator(device="cpu").manual_seed(20000 + i)
    xw = (torch.randn(T, H, generator=g) * 0.25).to(tdt).to(DEV)
    ssq = (torch.rand(H // 16, (T + 15) // 16 * 16, generator=g) * 16 + 4).to(DEV)
    y = ops.pq_gemm_packed_normed(xw, ssq, 1e-5, pk, p.codes_dev, p.codebooks_dev, p.scales_dev, [I, I], p.bias_dev)
    want = torch.empty(T, I, dtype=tdt, device=DEV)
    ops.silu_and_mul(want, y)
    got = ops.pq_gemm_packed_normed_silu_mul(xw, ssq, 1e-5, pk, p.codes_dev, p.codebooks_dev, p.scales_dev, p.bias_dev)
    diff = int((got.view(torch.int16) != want.view(torch.int16)).sum())
    # the same token alone gives the same row (the result does not depend on the batch)
    t = int(rng.integers(0, T))
    ssq1 = torch.zeros(H // 16, 16, device=DEV)
    ssq1[:, 0] = ssq[:, t]
    one = ops.pq_gemm_packed_normed_silu_mul(xw[t:t + 1].contiguous(), ssq1, 1e-5, pk, p.codes_dev, p.codebooks_dev,
                                             p.scales_dev, p.bias_dev)
    inv = int((one.view(torch.int16) != got[t:t + 1].view(torch.int16)).sum())
    if diff or inv:
        bad += 1
        print(f"case {i}: T={T} H={H} I={I} bias={bias} {dtype}: {diff} values differ from the two steps, {inv} from the single-token launch")
print(f"{n} cases, {bad} failed")
sys.exit(1 if bad else 0)
