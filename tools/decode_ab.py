#!/usr/bin/env python3
"""A/B of the two kernels that can take a T <= 16 launch (the K-split stream kernel and the decode kernel) on the
per-rank model shapes: same inputs, codes rotating over copies larger than the Infinity Cache, launches replayed
from a HIP graph, best of `--reps` replays / launches.  Prints one line per (shape, T) and writes a CSV.

  python tools/decode_ab.py --out gpurun_out/decode_ab.csv
"""
import argparse
import csv
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_pq_amd import _lib, ops  # noqa: E402
from tools.sweep import MODEL_SHAPES  # noqa: E402


def timed(x, sets, parts, reps):
    def call(s):
        return ops.pq_gemm_packed(x, s[3], s[0], s[1], s[2], parts, None)
    for s in sets[:2]:
        call(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for s in sets:
            call(s)
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / len(sets))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/decode_ab.csv")
    ap.add_argument("--reps", type=int, default=7)
    args = ap.parse_args()
    dev, dt, K, d = "cuda", torch.bfloat16, 256, 8
    rows = []
    for (name, parts, kin) in MODEL_SHAPES:
        N, G = sum(parts), kin // d
        copies = max(2, min(64, int(400e6 / (N * G)) + 1))
        sets = []
        for _ in range(copies):
            codes = torch.randint(-128, 128, (N, G, 1), device=dev, dtype=torch.int32).to(torch.int8)
            cbk = torch.randn(len(parts), K, 1, d, device=dev).to(dt)
            sc = (torch.rand(N, 1, 1, 1, device=dev) + 0.5).to(dt)
            sets.append((codes, cbk, sc, ops.pq_repack_codes(codes, cbk, parts)))
        ops.ensure_workspace(torch.device(dev, torch.cuda.current_device()), ops.workspace_bytes(128, N, G, 1, K, d, 1, dt))
        for T in (1, 2, 4, 8, 16):
            x = torch.randn(T, kin, device=dev).to(dt)
            r = dict(shape=name, N=N, Kin=kin, T=T)
            for k in ("stream", "decode"):
                os.environ["PQ_AMD_FORCE_KERNEL"] = k
                _lib.reload_knobs()
                kern = ops.gemm_kernel_name(T, N, G, 1, K, d, 1, dt, parts, packed=True)
                r[k + "_us"] = timed(x, sets, parts, args.reps) if k in kern else float("nan")
            os.environ.pop("PQ_AMD_FORCE_KERNEL")
            _lib.reload_knobs()
            r["default"] = ops.gemm_kernel_name(T, N, G, 1, K, d, 1, dt, parts, packed=True)
            rows.append(r)
            print(f"{name:16s} T={T:2d}  stream {r['stream_us']:6.2f} us   decode {r['decode_us']:6.2f} us   default {r['default']}",
                  flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w", newline="") as f:
        wr = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        wr.writeheader()
        for r in rows:
            wr.writerow({k: (f"{v:.3f}" if isinstance(v, float) else v) for k, v in r.items()})


if __name__ == "__main__":
    main()
