#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel-trace CSV: per (kernel, grid) count / avg / min duration in us."""
import collections
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
d = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        name = name[name.find("pq::"):][:70] if "pq::" in name else name[:50]
        d[(name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"])].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{k[0]:72s} grid=({k[1]},{k[2]}) wg={k[3]} n={len(v):5d} avg={sum(v)/len(v):9.2f} med={v2[len(v2)//2]:9.2f} min={v2[0]:9.2f} us")
