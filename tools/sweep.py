#!/usr/bin/env python3
"""Roofline sweep of the PQ linear through the C ABI: the model shapes (Llama-3-8B TP=1/TP=4, Llama-3-70B
TP=8 per-rank shapes) and the square sweep Kin = N in {4096, 8192} x K in {16, 256, 4096} x d in {2, 4, 8},
each at T in {1 .. 8192}.  Weights rotate over enough copies to exceed the Infinity Cache, launches are
replayed from a HIP graph, time = best of `--reps` replays / launches.  Writes a CSV: one row per point with
the algorithmic bytes and FLOPs, the duration, and the fraction of the binding roofline
max(bytes / 8 TB/s, flops / 2.5 PFLOP/s) / time.

  python tools/sweep.py --out gpurun_out/sweep.csv [--quick]
"""
import argparse
import csv
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_pq_amd import ops  # noqa: E402

HBM, MFMA = 8e12, 2.5e15
FOLD = True
MODEL_SHAPES = [
    ("8B.qkv", [4096, 1024, 1024], 4096), ("8B.o", [4096], 4096),
    ("8B.gate_up", [14336, 14336], 4096), ("8B.down", [4096], 14336),
    ("8B/tp4.qkv", [1024, 256, 256], 4096), ("8B/tp4.o", [4096], 1024),
    ("8B/tp4.gate_up", [3584, 3584], 4096), ("8B/tp4.down", [4096], 3584),
    ("70B/tp8.qkv", [1024, 128, 128], 8192), ("70B/tp8.o", [8192], 1024),
    ("70B/tp8.gate_up", [3584, 3584], 8192), ("70B/tp8.down", [8192], 3584),
]
TS = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192]


def point(w, name, parts, kin, K, d, T, dt, reps, budget_bytes):
    dev = "cuda"
    N, G = sum(parts), kin // d
    cb = 1 if K <= 256 else 2
    idt = torch.int8 if cb == 1 else torch.int16
    key = (tuple(parts), kin, K, d)
    if w.get("key") != key:
        w.clear()
        w["key"] = key
        copies = max(2, min(64, int(budget_bytes / (N * G * cb)) + 1))
        w["sets"] = []
        for _ in range(copies):
            lo, hi = (-(K // 2), K // 2) if K < 65536 else (-32768, 32768)
            codes = torch.randint(lo, hi, (N, G, 1), device=dev, dtype=torch.int32).to(idt)
            cbk = torch.randn(len(parts), K, 1, d, device=dev).to(dt)
            sc = (torch.rand(N, 1, 1, 1, device=dev) + 0.5).to(dt)
            # as the linear method does at load time: 16-entry codebooks folded in pairs (ops.fold_pairs) -- what is
            # launched is the format with 256 entries and twice the sub-vector length; the row keeps the original
            # format's name and algorithmic bytes
            folded = ops.fold_pairs(codes, cbk) if FOLD else None
            if folded is not None:
                codes, cbk = folded
            w["sets"].append((codes, cbk, sc, ops.pq_repack_codes(codes, cbk, parts)))
    sets = w["sets"]
    x = torch.randn(T, kin, device=dev).to(dt)
    # the scratch a loaded layer has (process_weights_after_loading sizes it the same way)
    ops.ensure_workspace(torch.device(dev, torch.cuda.current_device()), ops.workspace_bytes(128, N, G, 1, K, d, cb, dt))

    def call(s):
        if s[3] is not None:
            return ops.pq_gemm_packed(x, s[3], s[0], s[1], s[2], parts, None)
        return ops.pq_gemm(x, s[0], s[1], s[2], parts, None)

    # long launches need few copies per replay
    flops = 2.0 * T * N * kin
    n_launch = len(sets) if flops < 2e11 else max(2, min(len(sets), 4))
    for s in sets[:2]:
        call(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for s in sets[:n_launch]:
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
        best = min(best, e0.elapsed_time(e1) * 1e-3 / n_launch)
    byts = N * G * cb + len(parts) * K * d * 2 + N * 2 + T * kin * 2 + T * N * 2
    kc, kb = sets[0][0], sets[0][1]
    kern = ops.gemm_kernel_name(T, N, kc.shape[1], 1, kb.shape[1], kb.shape[3], cb, dt, parts, packed=sets[0][3] is not None)
    if kb.shape[1] != K:
        kern += "(folded)"
    bound = "mfma" if flops / MFMA >= byts / HBM else "hbm"
    frac = max(flops / MFMA, byts / HBM) / best
    return dict(shape=name, N=N, Kin=kin, K=K, d=d, T=T, kernel=kern, us=best * 1e6, bytes=byts, flops=flops,
                gbps=byts / best / 1e9, tflops=flops / best / 1e12, bound=bound, frac=frac)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/sweep.csv")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--quick", action="store_true", help="model shapes only")
    ap.add_argument("--cache-mb", type=int, default=400, help="bytes of codes to rotate over")
    ap.add_argument("--only", default="", help="comma-separated substrings: only the shapes whose name holds one")
    ap.add_argument("--ts", default="", help="comma-separated token counts instead of the full grid")
    ap.add_argument("--no-fold", action="store_true", help="run 16-entry formats as they are, without folding pairs of groups")
    args = ap.parse_args()
    global FOLD
    FOLD = not args.no_fold
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[args.dtype]
    grid = [(n, p, kin, 256, 8) for (n, p, kin) in MODEL_SHAPES]
    if not args.quick:
        for size in (4096, 8192):
            for K in (16, 256, 4096):
                for d in (2, 4, 8):
                    grid.append((f"sq{size}.K{K}.d{d}", [size], size, K, d))
    if args.only:
        grid = [g for g in grid if any(k in g[0] for k in args.only.split(","))]
    ts = [int(t) for t in args.ts.split(",")] if args.ts else TS
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    rows, w, t0 = [], {}, time.time()
    with open(args.out, "w", newline="") as f:
        wr = None
        for (name, parts, kin, K, d) in grid:
            for T in ts:
                r = point(w, name, parts, kin, K, d, T, dt, args.reps, args.cache_mb * 1e6)
                if wr is None:
                    wr = csv.DictWriter(f, fieldnames=list(r.keys()))
                    wr.writeheader()
                wr.writerow({k: (f"{v:.4g}" if isinstance(v, float) else v) for k, v in r.items()})
                f.flush()
                rows.append(r)
            print(f"[{time.time() - t0:6.1f}s] {name:18s} " + " ".join(f"{r['frac']:.2f}" for r in rows[-len(ts):]),
                  flush=True)
    print("T grid:", ts)


if __name__ == "__main__":
    main()
