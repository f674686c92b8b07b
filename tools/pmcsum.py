#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection CSV: per kernel (name, grid) the mean, median and maximum of each
counter.  The persistent tiled kernel launches one block per CU whatever the layer, so one (name, grid) key mixes
layers; in the bench run most of its launches are the dominant layer's, whose value is the median."""
import collections
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "pq::" in name:
            name = name[name.find("pq::"):][:60]
        elif "_ZN2pq" in name:                      # a kernel whose name came out mangled
            name = "pq::" + name[name.find("_ZN2pq") + 6:][:56]
        else:
            name = name[:40]
        key = (name, r.get("Grid_Size", r.get("Grid_Size_X", "")))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, ctrs in agg.items():
    if "pq::" not in key[0]:
        continue
    print(key)
    for c, v in sorted(ctrs.items()):
        v = sorted(v)
        print(f"   {c:32s} n={len(v):4d} mean={sum(v)/len(v):16.1f} median={v[len(v)//2]:16.1f} max={v[-1]:16.1f}")
