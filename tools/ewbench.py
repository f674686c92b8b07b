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
