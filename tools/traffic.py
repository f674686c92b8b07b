#!/usr/bin/env python3
"""HBM bytes per launch of the two dominant kernels from the rocprofv3 counter passes of tools/profile_round.sh:
FETCH_SIZE (KiB; doubled: gfx950 counts a 16-byte-per-lane stream at half) + WRITE_SIZE (KiB), medians over the
launches of the kernel with the dominant layer's grid.  Writes the JSON bench.py reads for `roofline.traffic`.

  python tools/traffic.py <pmc_fetch dir> <pmc_write dir> <round tag>  > profiles/traffic_<tag>.json
"""
import collections
import csv
import glob
import json
import os
import socket
import subprocess
import sys


def medians(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[(r["Kernel_Name"], r.get("Grid_Size", r.get("Grid_Size_X", "")))].append(float(r["Counter_Value"]))
    return {k: sorted(v)[len(v) // 2] for k, v in agg.items()}


def pick(meds, mark, grid):
    hits = [v for (name, g), v in meds.items() if mark in name and g == grid]
    return hits[0] if len(hits) == 1 else None


def main():
    fetch, write, tag = medians(sys.argv[1], "FETCH_SIZE"), medians(sys.argv[2], "WRITE_SIZE"), sys.argv[3]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:   # the commit the counters belong to (no .git on the GPU box: the caller passes it in PQ_COMMIT)
        commit = os.environ.get("PQ_COMMIT") or subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"],
                                                                capture_output=True, text=True, timeout=10).stdout.strip()
    except Exception:
        commit = ""
    gpu = ""
    try:
        import torch
        if torch.cuda.is_available():
            pr = torch.cuda.get_device_properties(0)
            gpu = f"{pr.name} ({pr.gcnArchName}, {pr.multi_processor_count} CUs)"
    except Exception:
        pass
    out = {"commit": commit or "unknown", "box": {"host": socket.gethostname(), "gpu": gpu},
           "note": f"HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 per the gfx950 correction for "
                   f"16-B-per-lane streams, + WRITE_SIZE), round {tag}, medians over the launches of the kernel "
                   f"(the dominant layer's); see profiles/README.md"}
    # gate_up_proj at 4096 tokens: the 512 x 128 tiled kernel (256 persistent blocks); at 16 tokens: the decode
    # kernel's plain instantiation with 256 blocks
    for key, mark, grid in (("gate_up_proj@T4096:mfma_tiled_packed", "gemm_tiled_kernel<bool _Accum, 8, 1, 2, 8, 4, 2", "131072"),
                            ("gate_up_proj@T16:mfma_decode_packed", "gemm_decode_kernelIDF16bLb0ELi0ELi0E", "131072")):
        f, w = pick(fetch, mark, grid), pick(write, mark, grid)
        if f is not None and w is not None:
            out[key] = {"fetch_size_kib_raw": f, "write_size_kib": w, "hbm_bytes": int(round((2 * f + w) * 1024))}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
