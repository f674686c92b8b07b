#!/usr/bin/env python3
"""Host time of one PQ linear call issued eagerly (what a prefill step, which vLLM does not capture into a graph,
pays per layer): wall clock over many back-to-back calls on a small problem, so that the GPU is never the limit."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from vllm_pq_amd import ops

dev, dt = torch.device("cuda:0"), torch.bfloat16
stack = B.build_stack(1, dev, dt, layers=1)
blk = stack[0]
x = torch.randn(1, 4096, device=dev).to(dt)
layer = blk["o_proj"]
qm = layer.quant_method


def timed(fn, n=3000):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6


parts = list(layer.codebooks.output_partition_sizes)
print("torch.empty                       %.1f us issue" % timed(lambda: torch.empty((1, 4096), dtype=dt, device=dev))[0])
print("layer(x) (module forward)         %.1f us issue, %.1f us with drain" % timed(lambda: layer(x)))
print("quant_method.apply                %.1f us issue, %.1f us with drain" % timed(lambda: qm.apply(layer, x)))
print("ops.pq_gemm_packed                %.1f us issue, %.1f us with drain" % timed(
    lambda: ops.pq_gemm_packed(x, layer.pq_packed_codes, layer.codes, layer.codebooks, layer.scales, parts, None)))
print("ops._pq_gemm_impl (no dispatcher) %.1f us issue, %.1f us with drain" % timed(
    lambda: ops._pq_gemm_impl(x, layer.codes, layer.codebooks, layer.scales, parts, None, layer.pq_packed_codes)))
