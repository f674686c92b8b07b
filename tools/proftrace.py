#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel-trace CSV.

  python tools/proftrace.py <trace dir> [--launch-log launches.json]

Without a launch log: per (kernel, grid, workgroup) count / avg / med / min duration in us.  The persistent tiled
kernel launches 256 blocks whatever the layer, so that key mixes layers.  With the launch log `bench.py --launch-log`
wrote during the same run (every PQ GEMM launch of the process in launch order: layer, T, N, Kin, kernel) the GEMM
dispatches of the trace, in dispatch order, are matched one to one with it and summarised per problem: the averages
of that table are per layer and comparable with bench.py's own HIP-event figure.
"""
import argparse
import collections
import csv
import glob
import json

GEMM_MARKS = ("gemm_stream_kernel", "gemm_tiled_kernel", "gemm_decode_kernel", "gemm_wide_kernel", "gemm_generic_kernel")


def short(name):
    return name[name.find("pq::"):][:70] if "pq::" in name else name[:50]


def stats(v):
    v2 = sorted(v)
    return f"n={len(v):5d} avg={sum(v) / len(v):9.2f} med={v2[len(v2) // 2]:9.2f} min={v2[0]:9.2f} us"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace_dir")
    ap.add_argument("--launch-log", default=None)
    args = ap.parse_args()
    rows = []
    for f in glob.glob(args.trace_dir + "/**/*kernel_trace.csv", recursive=True):
        rows.extend(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3

    if args.launch_log:
        with open(args.launch_log) as f:
            log = json.load(f)
        gemms = [r for r in rows if any(m in r["Kernel_Name"] for m in GEMM_MARKS)]
        print(f"# per problem: {len(gemms)} GEMM dispatches in the trace, {len(log)} launches in the log")
        if len(gemms) != len(log):
            print("# COUNTS DIFFER: the log and the trace are not of the same run; no per-problem table")
        else:
            d = collections.defaultdict(list)
            for r, (layer, T, N, Kin, kern) in zip(gemms, log):
                d[(layer, T, N, Kin, kern)].append(dur(r))
            for k, v in sorted(d.items(), key=lambda kv: (-kv[0][1], -sum(kv[1]))):
                print(f"{k[0]:14s} T={k[1]:5d} N={k[2]:6d} Kin={k[3]:6d} {k[4]:22s} {stats(v)}")
        print()
    print("# per (kernel, grid, workgroup)")
    d = collections.defaultdict(list)
    for r in rows:
        d[(short(r["Kernel_Name"]), r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"])].append(dur(r))
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k[0]:72s} grid=({k[1]},{k[2]}) wg={k[3]} {stats(v)}")


if __name__ == "__main__":
    main()
