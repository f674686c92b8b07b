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


def layer_chain(T, layers=32):
    """All linears and element-wise steps of `layers` Llama-3-8B decoder layers back to back (attention itself
    left out: the value heads stand in), one HIP-graph replay per step: fused MLP entry against the two-step one."""
    import bench as B
    from vllm_pq_amd.host.layers import RMSNorm
    from vllm_pq_amd import ops
    dev, dt = torch.device("cuda:0"), torch.bfloat16
    stack = B.build_stack(1, dev, dt, layers=layers)
    norms = [(RMSNorm(4096, 1e-5).to(dev), RMSNorm(4096, 1e-5).to(dev)) for _ in range(layers)]
    for a, b in norms:
        a.weight.data = a.weight.data.to(dt); b.weight.data = b.weight.data.to(dt)
    x0 = torch.randn(T, 4096, device=dev).to(dt)

    def step(fused):
        h, res = x0.clone(), x0.clone()
        for blk, (n1, n2) in zip(stack, norms):
            h, res = n1(h, res)
            qkv, _ = blk["qkv_proj"](h)
            h, _ = blk["o_proj"](qkv[:, :4096])
            h, res = n2(h, res)
            gu = blk["gate_up_proj"]
            if fused:
                a = gu.quant_method.apply_silu_and_mul(gu, h)
            else:
                y, _ = gu(h)
                a = torch.empty(T, 14336, device=dev, dtype=dt)
                ops.silu_and_mul(a, y)
            h, _ = blk["down_proj"](a)
        return h

    out = {}
    for fused in (False, True):
        step(fused)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step(fused)
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        out[fused] = best
    print(f"layer chain x{layers} T={T:4d}: two-step MLP {out[False]:7.3f} ms   fused MLP entry {out[True]:7.3f} ms")


if len(sys.argv) > 2 and sys.argv[2] == "chain":
    for T in (1, 16, 64, 4096):
        layer_chain(T)
