#!/usr/bin/env python3
"""Times the element-wise kernels (silu_and_mul, rms_norm, fused_add_rms_norm) at Llama-3-8B sizes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for T in [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "16,256,4096").split(",")]:
    r = bench.time_neighbours(T, torch.device("cuda:0"), torch.bfloat16, 1)
    for k, v in r.items():
        print(f"T={T:5d} {k:20s} {v['launch_us']:8.2f} us  {v['achieved']:8.1f} GB/s  frac {v['frac']:.3f}")


def fused_vs_separate(T, copies=24):
    """Llama-3-8B gate_up_proj + SiLU-and-mul: one fused launch against the GEMM and the stand-alone activation."""
    from vllm_pq_amd import ops
    dev, dt, I, Kin = "cuda:0", torch.bfloat16, 14336, 4096
    sets = []
    for _ in range(copies):
        codes = torch.randint(-128, 128, (2 * I, Kin // 8, 1), device=dev, dtype=torch.int32).to(torch.int8)
        cb = torch.randn(2, 256, 1, 8, device=dev).to(dt)
        sc = (torch.rand(2 * I, 1, 1, 1, device=dev) + 0.5).to(dt)
        sets.append((codes, cb, sc, ops.pq_repack_codes(codes, cb, [I, I])))
    x = torch.randn(T, Kin, device=dev).to(dt)
    out = torch.empty(T, I, device=dev, dtype=dt)

    def separate(s):
        y = ops.pq_gemm_packed(x, s[3], s[0], s[1], s[2], [I, I], None)
        ops.silu_and_mul(out, y)

    def fused(s):
        return ops.pq_gemm_packed_silu_mul(x, s[3], s[0], s[1], s[2], None)

    res = {}
    for name, fn in (("separate", separate), ("fused", fused)):
        fn(sets[0])
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for s in sets:
                fn(s)
        g.replay()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / len(sets))
        res[name] = best
    print(f"gate_up + silu_and_mul T={T:4d}: separate {res['separate']:7.2f} us   fused entry {res['fused']:7.2f} us")


if len(sys.argv) > 2 and sys.argv[2] == "fused":
    for T in (1, 4, 16, 24, 32, 64):
        fused_vs_separate(T)
