#!/usr/bin/env python3
"""Exhaustive kernel-shape search against the library's own choice: for every (layer shape, batch size) time
every tile shape / K split of the streaming kernel and every tile shape of the tiled kernel (forced through the
PQ_AMD_* environment knobs) and report how far the default dispatch is from the best one found ("regret").

  python tools/autotune.py [--out gpurun_out/autotune.csv] [--tokens 1,16,32,...] [--full]
"""
import argparse
import csv
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_pq_amd import _lib, ops  # noqa: E402
from sweep import MODEL_SHAPES  # noqa: E402

STREAM = ["1x1", "2x1", "4x1", "7x1", "8x1", "4x2", "7x2", "2x4", "4x4"]
KNOBS = ("PQ_AMD_FORCE_KERNEL", "PQ_AMD_STREAM_CFG", "PQ_AMD_STREAM_KSPLIT", "PQ_AMD_TILED_CFG", "PQ_AMD_WIDE_CFG")


def set_env(**kw):
    for k in KNOBS:
        os.environ.pop(k, None)
    for k, v in kw.items():
        os.environ[k] = str(v)
    _lib.reload_knobs()


def time_call(fn, sets, reps=3):
    try:
        fn(sets[0])
        torch.cuda.synchronize()
    except (ValueError, RuntimeError):
        return None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for s in sets:
            fn(s)
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
    ap.add_argument("--out", default="gpurun_out/autotune.csv")
    ap.add_argument("--tokens", default="1,4,16,32,64,128,256,512")
    ap.add_argument("--full", action="store_true", help="all per-rank model shapes (default: Llama-3-8B TP=1)")
    ap.add_argument("--square", action="store_true",
                    help="the codebook sweep instead: Kin = N in {4096, 8192} x K in {16, 256, 4096} x d in {2, 4, 8}")
    ap.add_argument("--custom", default="", help="one more shape: 'rows,rows,..:in_features'")
    args = ap.parse_args()
    dev, dt = "cuda", torch.bfloat16
    shapes = [(n, p, k, 256, 8) for (n, p, k) in (MODEL_SHAPES if args.full else MODEL_SHAPES[:4])]
    if args.square:
        shapes = [(f"sq{n}.K{K}.d{d}", [n], n, K, d) for n in (4096, 8192) for K in (16, 256, 4096) for d in (2, 4, 8)]
    if args.custom:
        ps, kin = args.custom.split(":")
        shapes = [("custom " + args.custom, [int(v) for v in ps.split(",")], int(kin), 256, 8)]
    rows = []
    for name, parts, kin, K, d in shapes:
        N, G = sum(parts), kin // d
        cbytes = 1 if K <= 256 else 2
        copies = max(2, min(48, int(400e6 / (N * G * cbytes)) + 1))
        sets = []
        for _ in range(copies):
            codes = torch.randint(-(K // 2), K // 2, (N, G, 1), device=dev, dtype=torch.int32).to(
                torch.int8 if cbytes == 1 else torch.int16)
            cb = torch.randn(len(parts), K, 1, d, device=dev).to(dt)
            sc = (torch.rand(N, 1, 1, 1, device=dev) + 0.5).to(dt)
            sets.append((codes, cb, sc, ops.pq_repack_codes(codes, cb, parts)))
        ops.ensure_workspace(torch.device(dev, 0), ops.workspace_bytes(128, N, G, 1, K, d, cbytes, dt))
        for T in [int(t) for t in args.tokens.split(",")]:
            x = torch.randn(T, kin, device=dev).to(dt)
            use = sets if T <= 512 else sets[:4]
            fn = lambda s: ops.pq_gemm_packed(x, s[3], s[0], s[1], s[2], parts, None)  # noqa: E731
            set_env()
            default = time_call(fn, use)
            cands = {}
            for cfg in STREAM:
                if int(cfg.split("x")[1]) * 16 > max(T, 16) * 2 and cfg.split("x")[1] != "1":
                    continue       # a token tile more than twice the batch
                for ks in (1, 2, 4):
                    set_env(PQ_AMD_FORCE_KERNEL="stream", PQ_AMD_STREAM_CFG=cfg, PQ_AMD_STREAM_KSPLIT=ks)
                    t = time_call(fn, use)
                    if t is not None:
                        cands[f"stream {cfg} ks{ks}"] = t
            if T <= 128 and K <= 256 and d == 8:
                set_env(PQ_AMD_FORCE_KERNEL="decode")
                if ops.gemm_kernel_name(T, N, G, 1, K, d, cbytes, dt, parts, packed=True) == "mfma_decode_packed":
                    t = time_call(fn, use)
                    if t is not None:
                        cands["decode"] = t
            if T > 16 and K <= 256 and d == 8 and G % 64 == 0 and G // 64 <= 8:
                # the mid-batch kernel: its own choice of block shape, and every shape whose grid is within two passes of the chip
                set_env(PQ_AMD_FORCE_KERNEL="wide")
                if ops.gemm_kernel_name(T, N, G, 1, K, d, cbytes, dt, parts, packed=True) == "mfma_wide_packed":
                    t = time_call(fn, use)
                    if t is not None:
                        cands["wide"] = t
                    nch = G // 64
                    wr = 8 >> (3 if nch >= 5 else 2 if nch >= 3 else 1 if nch >= 2 else 0)
                    for nf in (1, 2, 3, 4):
                        for rounds in range(1, 64 // (wr * nf) + 1):
                            rows_blk = wr * rounds * nf * 16
                            blocks = sum(-(-p_ // rows_blk) for p_ in parts) * (-(-T // 32))
                            if blocks > 512 or (blocks < 128 and rounds > 1):
                                continue
                            set_env(PQ_AMD_FORCE_KERNEL="wide", PQ_AMD_WIDE_CFG=f"{nf}x{rounds}")
                            t = time_call(fn, use)
                            if t is not None:
                                cands[f"wide {nf}x{rounds}"] = t
            if T > 16:
                for i in range(7):
                    set_env(PQ_AMD_FORCE_KERNEL="tiled", PQ_AMD_TILED_CFG=i)
                    t = time_call(fn, use)
                    if t is not None:
                        cands[f"tiled cfg{i}"] = t
            set_env()
            default = min(default, time_call(fn, use))      # before and after the candidates: clocks and caches settle
            best = min(cands, key=cands.get)
            wides = {k: v for k, v in cands.items() if k.startswith("wide")}
            others = {k: v for k, v in cands.items() if not k.startswith("wide")}
            bw = min(wides, key=wides.get) if wides else ""
            bo = min(others, key=others.get) if others else ""
            rows.append(dict(shape=name, T=T, default_us=round(default, 2), best_us=round(cands[best], 2), best=best,
                             regret=round(default / cands[best] - 1, 3), wide_us=round(wides.get("wide", 0.0), 2),
                             best_wide=bw, best_wide_us=round(wides[bw], 2) if bw else 0.0, best_other=bo,
                             best_other_us=round(others[bo], 2) if bo else 0.0))
            print(f"{name:16s} T={T:5d} default {default:8.2f} us   best {cands[best]:8.2f} us ({best:16s}) "
                  f"regret {100 * (default / cands[best] - 1):5.1f} %   wide {wides.get('wide', 0):7.2f}  best wide {bw} "
                  f"{wides.get(bw, 0):7.2f}  best other {bo} {others.get(bo, 0):7.2f}", flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w", newline="") as f:
        wr = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        wr.writeheader()
        wr.writerows(rows)


if __name__ == "__main__":
    main()
