#!/usr/bin/env python3
"""Per-kernel micro-benchmark: times pq_gemm / pq_gemm_packed on one linear shape with rotating weight
copies (so codes stream from HBM, not from the Infinity Cache), replayed from a HIP graph.

  python tools/kbench.py --shape gate_up --tokens 1,16 [--natural] [--copies 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_pq_amd import ops  # noqa: E402

SHAPES = {  # name: (partition sizes, Kin)
    "qkv": ([4096, 1024, 1024], 4096), "o": ([4096], 4096), "gate_up": ([14336, 14336], 4096),
    "down": ([4096], 14336), "lm_head": ([128256], 4096), "sq4k": ([4096], 4096), "sq8k": ([8192], 8192),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="gate_up")
    ap.add_argument("--tokens", default="1,16")
    ap.add_argument("--K", type=int, default=256)
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--natural", action="store_true", help="reference code layout instead of the packed one")
    ap.add_argument("--workspace", action="store_true", help="give the library its K-split workspace")
    ap.add_argument("--copies", type=int, default=0, help="weight copies to rotate over (0: enough for 600 MB)")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    dev = "cuda:0"
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[args.dtype]
    for shape in args.shape.split(","):
        parts, kin = SHAPES[shape]
        N, G = sum(parts), kin // args.d
        cb_bytes = 1 if args.K <= 256 else 2
        idt = torch.int8 if cb_bytes == 1 else torch.int16
        copies = args.copies or max(2, int(600e6 / (N * G * cb_bytes)) + 1)
        ws = []
        for _ in range(copies):
            codes = torch.randint(-128, 128, (N, G, 1), device=dev, dtype=torch.int8).to(idt)
            cbk = torch.randn(len(parts), args.K, 1, args.d, device=dev).to(dt)
            sc = (torch.rand(N, 1, 1, 1, device=dev) + 0.5).to(dt)
            pk = None if args.natural else ops.pq_repack_codes(codes, cbk, parts)
            ws.append((codes, cbk, sc, pk))
        for T in [int(t) for t in args.tokens.split(",")]:
            x = torch.randn(T, kin, device=dev).to(dt)
            ops.drop_workspace()
            if args.workspace:
                ops.ensure_workspace(torch.device(dev), ops.workspace_bytes(T, N, G, 1, args.K, args.d, cb_bytes, dt))

            def call(w):
                if w[3] is not None:
                    return ops.pq_gemm_packed(x, w[3], w[0], w[1], w[2], parts, None)
                return ops.pq_gemm(x, w[0], w[1], w[2], parts, None)

            for w in ws[:2]:
                call(w)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for w in ws:
                    call(w)
            g.replay()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(args.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e-3 / copies)
            byts = N * G * cb_bytes + len(parts) * args.K * args.d * 2 + N * 2 + T * kin * 2 + T * N * 2
            flops = 2.0 * T * N * kin
            name = ops.gemm_kernel_name(T, N, G, 1, args.K, args.d, cb_bytes, dt, parts, packed=ws[0][3] is not None)
            print(f"{shape:8s} T={T:5d} {name:20s} {best*1e6:9.2f} us  {byts/best/1e9:8.1f} GB/s  "
                  f"{flops/best/1e12:8.2f} TFLOP/s  (copies={copies})", flush=True)


if __name__ == "__main__":
    main()
