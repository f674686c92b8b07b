#!/usr/bin/env python3
"""How much of a prefill launch's time is the clock the chip holds: gate_up_proj (28672 x 4096) at 4096 tokens
through the library with different operand statistics and weight working sets, many launches back to back.

  python tools/powerprobe.py

Operands: "randn" (normal values: exponents and mantissas toggle), "narrow" (random sign and mantissa, one exponent:
values in +-[0.5, 1), what tools/tiled_lab uses), "zeros".  Working set: 1 copy of the weights (codes stay in the
Infinity Cache) or 41 (600 MB: streamed from HBM).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_pq_amd import ops  # noqa: E402


def operand(kind, shape, dt, dev):
    if kind == "randn":
        return torch.randn(shape, device=dev).to(dt)
    if kind == "zeros":
        return torch.zeros(shape, device=dev, dtype=dt)
    bits = torch.randint(0, 65536, shape, device=dev, dtype=torch.int32)
    bits = (bits & 0x807f) | 0x3f00
    return bits.to(torch.int16).view(torch.bfloat16) if dt == torch.bfloat16 else bits.to(torch.int16).view(dt)


def main():
    dev, dt = "cuda:0", torch.bfloat16
    parts, kin, T = [14336, 14336], 4096, 4096
    N, G = sum(parts), kin // 8
    for kind in ("randn", "narrow", "zeros"):
        for copies in (1, 41):
            ws = []
            for _ in range(copies):
                codes = torch.randint(-128, 128, (N, G, 1), device=dev, dtype=torch.int8)
                cbk = operand(kind, (2, 256, 1, 8), dt, dev)
                sc = torch.ones(N, 1, 1, 1, device=dev, dtype=dt)
                ws.append((codes, cbk, sc, ops.pq_repack_codes(codes, cbk, parts)))
            x = operand(kind, (T, kin), dt, dev)
            y = None
            for w in ws[:2]:
                y = ops.pq_gemm_packed(x, w[3], w[0], w[1], w[2], parts, None)
            torch.cuda.synchronize()
            n = 164
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(n):
                    w = ws[i % copies]
                    ops.pq_gemm_packed(x, w[3], w[0], w[1], w[2], parts, None)
            g.replay()
            torch.cuda.synchronize()
            ts = []
            for _ in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / n)
            fl = 2.0 * T * N * kin
            print(f"{kind:7s} copies={copies:2d}: " + " ".join(f"{t:7.1f}" for t in ts) +
                  f" us/launch  -> {fl / min(ts) / 1e6:7.1f} TFLOP/s best", flush=True)
            del ws, x, y, g
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
