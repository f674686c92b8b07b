#!/usr/bin/env python3
"""Layer chain (32 Llama-3-8B decoder layers, linears + norms + activation, one HIP-graph replay) with the add + norm
and the activation carried by the GEMMs against separate launches, per token count:  python tools/chain_ab.py 16,32,64,128"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B

dev, dt = torch.device("cuda:0"), torch.bfloat16
stack = B.build_stack(1, dev, dt)
for T in [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "16,32,64,128").split(",")]:
    r = B.layer_chain(stack, T, dev, dt)
    print(f"T={T:4d}: fused {r['ms']:.3f} ms   separate {r.get('ms_separate_norms', float('nan')):.3f} ms", flush=True)
